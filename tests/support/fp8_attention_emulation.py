"""Which fp8 form of the DiT self-attention would survive peaky softmax rows? (last sentence.) Priced on the fp32 oracle BEFORE any
kernel is written, like fp8_linear_emulation.py: F.scaled_dot_product_attention is wrapped for the DiT's self-attention calls (16 heads x 72, no mask,
as many keys as queries) and its operands are fake-quantised the way a kernel would see them - products and sums in fp32, i.e. what an MFMA with exact
accumulation computes:

  full fp8      what flash_attn_fp8_kernel does: Q (one scale per query row), K (per 64-key tile and head) in e4m3 for Q K^T; the probabilities in e4m3
                with one scale per (query, 32-key block) taken from the block's maximum, V in e4m3 per 64-key tile for P V; the denominator is the
                sum of the SAME quantised probabilities (the kernel's ones row)
  QK bf16       Q K^T on bf16 operands (the logits keep 8 bits), P and V as above
  PV bf16       Q K^T on e4m3 operands, P and V in bf16 (which half of 'full fp8' costs what)
  V hi + lo     Q K^T on bf16, P in bf16, V as a pair of e4m3 values (hi = e4m3(V), lo = e4m3(V - hi)): two fp8 MFMAs per V tile
  bf16          every operand rounded to bf16 (the shipped bf16 kernel's operand precision): the floor

on the seeded weights and on the stress weights (tests/support/stress_weights.py with stress_512.npz's gains), 512 x 512, uint8 result against the
un-quantised oracle's. Test infrastructure: runs the oracle, not the product.

    python tests/support/fp8_attention_emulation.py > profiles/r06_fp8_attention_forms_emulation.txt        (CPU only; about 10 s per pass on 8 threads)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
F = torch.nn.functional


def fq(x, dims):
    """Fake-quantise to OCP e4m3 with one scale over `dims` (amax -> 448)."""
    s = x.abs().amax(dim=dims, keepdim=True).clamp_min(1e-30) / 448.0
    return (x / s).to(torch.float8_e4m3fn).to(torch.float32) * s


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def attention(q, k, v, qk, pv):
    """q, k, v [B, H, T, d] fp32; qk in {fp8, bf16}, pv in {fp8, bf16, hilo}. Exact softmax in fp32 around the quantised operands."""
    B, H, T, d = q.shape
    scale = d ** -0.5
    if qk == "fp8":
        qq = fq(q, (-1,))
        kk = fq(k.reshape(B, H, T // 64, 64, d), (-2, -1)).reshape(B, H, T, d)
    else:
        qq, kk = bf(q), bf(k)
    s = (qq @ kk.transpose(-1, -2)) * scale
    p = torch.exp(s - s.amax(-1, keepdim=True))
    if pv == "fp8":
        pq = fq(p.reshape(B, H, T, T // 32, 32), (-1,)).reshape(B, H, T, T)
        vq = fq(v.reshape(B, H, T // 64, 64, d), (-2, -1)).reshape(B, H, T, d)
    elif pv == "hilo":
        pq = bf(p)
        hi = fq(v.reshape(B, H, T // 64, 64, d), (-2, -1)).reshape(B, H, T, d)
        lo = fq((v - hi).reshape(B, H, T // 64, 64, d), (-2, -1)).reshape(B, H, T, d)
        vq = hi + lo
    else:
        pq, vq = bf(p), bf(v)
    return (pq @ vq) / pq.sum(-1, keepdim=True)


FORMS = [("bf16 operands (floor)", "bf16", "bf16"), ("full fp8 (the shipped fp8 kernel's operands)", "fp8", "fp8"), ("QK bf16, PV fp8", "bf16", "fp8"),
         ("QK fp8, PV bf16", "fp8", "bf16"), ("QK bf16, P bf16, V hi + lo e4m3", "bf16", "hilo")]


def main():
    import bench
    from oracle import dit as odit, glue as oglue, swinir as oswin, vae as ovae
    from tests.golden.make_stress_fixture import base_state_dicts
    from tests.support.stress_weights import stress_state_dicts
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    base = base_state_dicts()
    z = np.load(os.path.join(ROOT, "tests", "golden", "stress_512.npz"))
    gains = {"dit": [float(v) for v in z["logit_gain_dit"]], "vae_encoder": float(z["logit_gain_vae"][0]), "vae_decoder": float(z["logit_gain_vae"][1])}
    sets = {"seeded": (base, bench.synthetic_lq(1, 512, 512, 15)[0].numpy()),
            "stress": (stress_state_dicts(base, float(z["frac"]), float(z["gain"]), gains), bench.synthetic_lq(1, 512, 512, int(z["lq_seed"]))[0].numpy())}
    y, mask = bench.synthetic_prompt()
    real = F.scaled_dot_product_attention
    form = [None]

    def sdpa(q, k, v, attn_mask=None, scale=None, **kw):
        if form[0] is not None and attn_mask is None and q.shape[1] == 16 and q.shape[-2] == k.shape[-2] and q.shape[-1] == 72:
            return attention(q, k, v, form[0][0], form[0][1])
        return real(q, k, v, attn_mask=attn_mask, scale=scale, **kw)

    F.scaled_dot_product_attention = sdpa
    try:
        for name, (sds, img) in sets.items():
            def run():
                out, _ = oglue.process([img], lambda x: oswin.swinir_forward(sds["swin"], x), lambda x: ovae.vae_encode_mean(sds["vae"], x),
                                       lambda lat, t, yy, mm: odit.dit_forward(sds["dit"], lat, t, yy, mm), lambda zz: ovae.vae_decode(sds["vae"], zz),
                                       oglue.alphas_cumprod_diffusers(), y, mask)
                return out[0].astype(np.float64)
            form[0] = None
            ref = run()
            print(f"== {name} weights, 512 x 512 (1024 tokens per head), DiT self-attention operands of all 28 blocks; uint8 result against the un-quantised fp32 oracle")
            for label, qk, pv in FORMS:
                form[0] = (qk, pv)
                got = run()
                mse = float(((got - ref) ** 2).mean()) / 255.0 ** 2
                print(f"{label:50s} {10 * np.log10(1 / max(mse, 1e-12)):7.2f} dB   noise {mse * 1e6:8.2f}e-6", flush=True)
    finally:
        F.scaled_dot_product_attention = real


if __name__ == "__main__":
    main()
