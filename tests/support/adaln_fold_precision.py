"""the precision cost of folding the adaLN LayerNorm + modulate into the GEMM that follows it, measured on the fp32 oracle's own
intermediates (CPU; test infrastructure like the other emulations in this directory - the product never imports it).

    python tests/support/adaln_fold_precision.py > profiles/r06_adaln_fold_precision.txt

With the timestep fixed, h = LN(x) (1 + scale) + shift, out = h W^T + b can be written as
    out = rstd_row * (x W'^T - mu_row * colsum(W')) + (shift W^T + b),   W' = W (1 + scale)   (a constant matrix per layer),
which removes the 57 LayerNorm launches per image (1.36 ms) if the GEMM may read the RAW residual stream as bf16. This script runs the oracle's DiT
(full 28-block architecture, 64 x 64 latent = 1024 tokens) on the seeded weights and on the stress weights (tests/support/stress_weights.py), taps
every adaLN site (the LN in front of q|k|v and the LN in front of fc1: oracle/dit.py block(), PixArtMS.py:67-79), and compares against the fp32 product
  cur  = bf16(h) bf16(W)^T                      - what the HIP path computes today (layernorm_v4_kernel writes bf16(h), gemm_pp_kernel multiplies), and
  fold = rstd (bf16(x) bf16(W')^T - mu colsum)  - the folded form with fp32 row statistics and an fp32 epilogue.
Reported per site: relative L2 error of both and their ratio, plus |mu| / sigma and max|x| / sigma of the rows (what the ratio follows)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def measure(sd, tag):
    from oracle import dit as odit
    F = torch.nn.functional
    C = 1152
    real_ln, real_lin = F.layer_norm, odit._lin
    state, rows = {"x": None}, []

    def ln(x, shape, weight=None, bias=None, eps=1e-5):
        if tuple(shape) == (C,) and weight is None and x.dim() == 3:
            state["x"] = x
        return real_ln(x, shape, weight, bias, eps)

    def lin(sd_, p, h):
        if state["x"] is not None and (p.endswith("attn1.to_q") or p.endswith("ff.net.0.proj")) and h.dim() == 3 and h.shape[-1] == C:
            x = state["x"][0].double()
            hh = h[0].double()
            n = real_ln(x, (C,), eps=1e-6)
            # h = n * (1 + scale) + shift per channel: exact from the tokens (two unknowns per channel)
            nm, hm = n.mean(0), hh.mean(0)
            one_sc = ((n - nm) * (hh - hm)).sum(0) / ((n - nm) ** 2).sum(0)
            sh = hm - one_sc * nm
            names = ("attn1.to_q", "attn1.to_k", "attn1.to_v") if p.endswith("attn1.to_q") else ("ff.net.0.proj",)
            base = p[: -len("attn1.to_q")] if p.endswith("attn1.to_q") else p[: -len("ff.net.0.proj")]
            W = torch.cat([sd_[base + nme + ".weight"] for nme in names]).double()
            ref = hh @ W.T
            cur = (bf(hh.float()).double() @ bf(W.float()).double().T)
            Wp = bf((W * one_sc[None, :]).float()).double()
            mu = x.mean(1, keepdim=True)
            rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-6)
            fold = rstd * (bf(x.float()).double() @ Wp.T - mu * Wp.sum(1)[None, :]) + (sh[None, :] @ W.T)
            e_cur = float((cur - ref).norm() / ref.norm())
            e_fold = float((fold - ref).norm() / ref.norm())
            sig = x.std(1, unbiased=False)
            rows.append((p, e_cur, e_fold, float((mu.abs()[:, 0] / sig).median()), float((x.abs().max(1).values / sig).median())))
            if not p.endswith("attn1.to_q"):
                state["x"] = None
        return real_lin(sd_, p, h)

    F.layer_norm, odit._lin = ln, lin
    try:
        g = torch.Generator().manual_seed(5)
        lat = torch.randn(1, 4, 64, 64, generator=g)
        y = torch.randn(1, 120, 4096, generator=g) * 0.2
        mask = torch.ones(1, 120)
        with torch.no_grad():
            odit.dit_forward(sd, lat, torch.tensor([400.0]), y, mask)
    finally:
        F.layer_norm, odit._lin = real_ln, real_lin
    print(f"== {tag}: 28 blocks x 2 adaLN sites, 1024 tokens; relative L2 error against the fp32 product")
    print(f"{'site':46s} {'bf16(h) W (today)':>18s} {'folded':>10s} {'ratio':>7s} {'|mu|/sigma':>11s} {'max|x|/sigma':>13s}")
    for p, ec, ef, ms, xs in rows:
        if int(p.split('.')[1]) in (0, 1, 7, 14, 21, 27):
            print(f"{p:46s} {ec:18.2e} {ef:10.2e} {ef / ec:7.2f} {ms:11.3f} {xs:13.1f}")
    import statistics
    rq = [ef / ec for p, ec, ef, _, _ in rows if p.endswith("to_q")]
    rf = [ef / ec for p, ec, ef, _, _ in rows if not p.endswith("to_q")]
    print(f"all 28 blocks: folded / today error ratio, q|k|v sites median {statistics.median(rq):.2f} (max {max(rq):.2f}); fc1 sites median {statistics.median(rf):.2f} (max {max(rf):.2f})")
    return rows


def main():
    torch.set_num_threads(8)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_stress_fixture as msf
    from tests.support.stress_weights import stress_state_dicts
    sds = msf.base_state_dicts()
    print(__doc__.split("\n\n")[0].replace("\n", " "))
    measure(sds["dit"], "seeded weights (bench.py's)")
    st = stress_state_dicts(sds, frac=0.01, gain=30.0, logit_gain=4.0)
    measure(st["dit"], "stress weights (1 % of the residual-stream channels x30, logits x4)")


if __name__ == "__main__":
    main()
