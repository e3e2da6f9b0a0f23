"""What would e4m3 operands in the DiT linears cost? (VERDICT r04 item 1c asks for fp8 operands in gemm_pp_kernel's launches.) Priced BEFORE
writing the kernel, on the fp32 oracle: oracle.dit._lin is wrapped so that chosen groups of linears see fake-quantised operands - the
activation rounded to OCP e4m3 with one scale per token row (amax -> 448), the weight with one scale per output channel, products and sums
in fp32: exactly what an MX-scaled fp8 MFMA with exact accumulation would compute. Full architectures, bench.py's seeded weights, one
512 x 512 image; the uint8 result of the whole path against the un-quantised oracle's.

    python tests/support/fp8_linear_emulation.py          (CPU only; about 20 s per pass on 8 threads)

Test infrastructure: runs the oracle, not the product."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

GROUPS = {"attn1.to_q|attn1.to_k|attn1.to_v": "self-attention q/k/v projection (LN-modulated input)", "attn1.to_out.0": "self-attention output projection",
          "attn2.to_q": "cross-attention q projection (raw residual stream)", "attn2.to_out.0": "cross-attention output projection",
          "ff.net.0.proj": "MLP fc1 (LN-modulated input)", "ff.net.2": "MLP fc2 (GELU output)"}


def fq(x, dim):
    """Fake-quantise to e4m3 with one scale along `dim` (amax -> 448)."""
    s = x.abs().amax(dim=dim, keepdim=True).clamp_min(1e-12) / 448.0
    return (x / s).to(torch.float8_e4m3fn).to(torch.float32) * s


def main():
    import bench
    from instarevive_amd import weights as W
    from oracle import dit as odit, glue as oglue, swinir as oswin, vae as ovae
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    sds = dict(swin=bench.random_state_dict(W.swinir_shapes(dict(embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8, mlp_ratio=2)), 1),
               vae=bench.random_state_dict(W.vae_shapes(dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2)), 2),
               dit=bench.random_state_dict(W.dit_shapes(dict(num_layers=28, num_attention_heads=16, attention_head_dim=72, caption_channels=4096)), 3))
    y, mask = bench.synthetic_prompt()
    img = bench.synthetic_lq(1, 512, 512, 15)[0].numpy()
    plain_lin = odit._lin
    active = [()]

    def lin(sd, p, x):
        if p.startswith("transformer_blocks.") and any(p.endswith(k) for g in active[0] for k in g.split("|")):
            return torch.nn.functional.linear(fq(x, -1), fq(sd[p + ".weight"], 1), sd[p + ".bias"])
        return plain_lin(sd, p, x)

    odit._lin = lin

    def run(groups):
        active[0] = groups
        out, _ = oglue.process([img], lambda x: oswin.swinir_forward(sds["swin"], x), lambda x: ovae.vae_encode_mean(sds["vae"], x),
                               lambda lat, t, yy, mm: odit.dit_forward(sds["dit"], lat, t, yy, mm), lambda z: ovae.vae_decode(sds["vae"], z),
                               oglue.alphas_cumprod_diffusers(), y, mask)
        return out[0].astype(np.float64)

    ref = run(())
    print("e4m3 operands (per-token activation scale, per-channel weight scale, fp32 accumulation) in the DiT linears of all 28 blocks;")
    print("uint8 result against the un-quantised fp32 oracle: PSNR, and the noise power in 1e-6 of full scale (two uint8 roundings: 2.6)")
    for g in list(GROUPS) + ["ALL"]:
        out = run(tuple(GROUPS) if g == "ALL" else (g,))
        mse = ((out - ref) ** 2).mean() / 255.0 ** 2
        print(f"{(GROUPS.get(g) or 'all six groups'):58s} {10 * np.log10(1 / (mse + 1e-12)):7.2f} dB   {mse * 1e6:8.2f}", flush=True)


if __name__ == "__main__":
    main()
