#!/usr/bin/env python3
"""At which logit spread do the DiT's fixed-reference attention kernels (flash_attn_pp2_kernel, flash_attn_x72_kernel) first take the rescaling fallback,
and what does it cost? The stress weights of tests/golden/stress_512.npz (1 % channels x30, per-block logit gains calibrated to a
median per-row spread of 34 at 512 x 512) with the DiT gains multiplied by f; per f and size: ms per ir_pipeline call (median of 3) and the attention
launches that raised the overflow flag (ir_attn_fallback_count; 28 DiT self-attention + 2 VAE launches per call). The median spread the SAME gains give
at the size's token count comes from the oracle's fixtures (stress_512.npz at 512, stress_headline.npz at 2048) and scales with f.

    python tools/spread_sweep.py [--sizes 512 2048] [--factors 1 1.5 2 2.5 3 4]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[512, 2048])
    ap.add_argument("--factors", type=float, nargs="+", default=[1.0, 1.5, 2.0, 2.5, 3.0, 4.0])
    a = ap.parse_args()
    from instarevive_amd import _lib as L
    from tests.support.stress_weights import stress_state_dicts
    dev = torch.device("cuda", 0)
    swin, vae, dit, sched, sds = bench.build_models(dev, lambda m: None)
    ctx = dit.ctx
    y, mask = bench.synthetic_prompt()
    yd, md = y.to(dev), mask.to(dev)
    z = np.load(os.path.join(ROOT, "tests", "golden", "stress_512.npz"))
    g0 = {"dit": [float(v) for v in z["logit_gain_dit"]], "vae_encoder": float(z["logit_gain_vae"][0]), "vae_decoder": float(z["logit_gain_vae"][1])}
    spread = {512: float(np.median(z["spread_median"][:28]))}
    hp = os.path.join(ROOT, "tests", "golden", "stress_headline.npz")
    if os.path.exists(hp):
        zh = np.load(hp)
        if "spread_median_2048" in zh:
            spread[2048] = float(np.nanmedian(zh["spread_median_2048"][:28]))
    acp, sf = float(sched.alphas_cumprod[400]), float(vae.config.scaling_factor)
    for S in a.sizes:
        din = (bench.upscale_bicubic(bench.synthetic_lq(1, S // 4, S // 4, 22), 4.0) if S > 512 else bench.synthetic_lq(1, 512, 512, int(z["lq_seed"]))).to(dev)
        dout = torch.empty_like(din)
        ws = ctx.workspace(ctx.ws_bytes(L.STAGE_PIPELINE, 1, S, S, 0, 512, 448))
        print(f"== {S} x {S} ({(S // 16) ** 2} DiT tokens); median per-row logit spread of the DiT blocks at f = 1: {spread.get(S, float('nan')):.1f} (oracle)")
        for f in a.factors:
            gains = dict(g0, dit=[g * f for g in g0["dit"]])
            st = stress_state_dicts(sds, float(z["frac"]), float(z["gain"]), gains)
            vae.load_state_dict(st["vae"])
            dit.load_state_dict(st["dit"])
            dit.set_prompt(yd, md)
            dit.ensure_pos(S // 16, S // 16)
            times, fb = [], None
            ctx.check(ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), 1), "count")
            for i in range(4):
                if i == 1:
                    fb = ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), 0)
                    ctx.check(ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), -1), "count")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.check(ctx.lib.ir_pipeline(ctx.h, ctx.stream(), L.ptr(din), L.ptr(dout), None, 1, S, S, 0, 512, 448, 400.0, acp, sf, L.ptr(ws), ws.numel()), "ir_pipeline")
                e1.record()
                torch.cuda.synchronize()
                if i:
                    times.append(e0.elapsed_time(e1))
            fin = bool(torch.isfinite(dout.float()).all())
            print(f"f = {f:4.1f} (median spread ~{spread.get(S, float('nan')) * f:6.1f}): {np.median(times):8.2f} ms, {fb:2d} of 30 attention launches took the fallback, "
                  f"output std {float(dout.float().std()):.1f}{'' if fin else ' NON-FINITE'}", flush=True)
        vae.load_state_dict(sds["vae"])
        dit.load_state_dict(sds["dit"])


if __name__ == "__main__":
    main()
