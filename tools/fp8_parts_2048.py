#!/usr/bin/env python3
"""Choose the DEFAULT fp8 operand set of cfg-5 by the guard, not by speed: at the headline size (2048 x 2048, the
input of tests/golden/headline_crops.npz) every part of ir_set_fp8_mask is switched ON alone; its time saving (ms per ir_pipeline call,
HIP events, median of 5) and its added noise power (crops against the fp32 oracle's, minus the bf16 path's) are measured, and the parts
are taken greedily by noise per millisecond until the whole path would drop below 46.3 dB + margin (0.1 dB within a 30 dB reference:
tests/support/psnr_guard.py). Prints the table, the chosen mask and the measured time / PSNR of that mask, of "all" and of "attention".

    python tools/fp8_parts_2048.py [--gate 46.3] [--margin 0.15]

The oracle is not run here: the fixture is the checker."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tools.fp8_attribution import PARTS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gate", type=float, default=46.3)
    ap.add_argument("--margin", type=float, default=0.15)
    ap.add_argument("--size", type=int, default=2048)
    a = ap.parse_args()
    from instarevive_amd import _lib as L
    from tests.golden.make_headline_crops import CROP, inputs_for
    dev = torch.device("cuda", 0)
    swin, vae, dit, sched, sds = bench.build_models(dev, lambda m: print(m, flush=True))
    ctx = dit.ctx
    y, mask = bench.synthetic_prompt()
    dit.set_prompt(y.to(dev), mask.to(dev))
    z = np.load(os.path.join(ROOT, "tests", "golden", "headline_crops.npz"))
    S = a.size
    img = inputs_for(S)
    pos, want = z[f"pos_{S}"], z[f"crops_{S}"].astype(np.float64)
    din = torch.from_numpy(img)[None].to(dev)
    dout = torch.empty_like(din)
    dit.ensure_pos(S // 16, S // 16)
    vae.enable_fp8(True)
    ctx.check(ctx.lib.ir_set_fp8(ctx.h, 0), "ir_set_fp8")
    acp, sf = float(sched.alphas_cumprod[400]), float(vae.config.scaling_factor)
    ws = ctx.workspace(ctx.ws_bytes(L.STAGE_PIPELINE, 1, S, S, L.FLAG_FP8, 512, 448))

    def run(mask_bits, reps=5):
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, mask_bits), "ir_set_fp8_mask")
        flags = L.FLAG_FP8 if mask_bits else 0
        times = []
        for i in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.check(ctx.lib.ir_pipeline(ctx.h, ctx.stream(), L.ptr(din), L.ptr(dout), None, 1, S, S, flags, 512, 448, 400.0, acp, sf, L.ptr(ws), ws.numel()), "ir_pipeline")
            e1.record()
            torch.cuda.synchronize()
            if i:
                times.append(e0.elapsed_time(e1))
        out = dout[0].cpu().numpy()
        got = np.stack([out[yy:yy + CROP, xx:xx + CROP] for yy, xx in pos]).astype(np.float64)
        mse = ((got - want) ** 2).mean() / 255.0 ** 2
        return float(np.median(times)), mse

    try:
        t_bf, n_bf = run(0)
        db = lambda n: 10 * np.log10(1.0 / n)
        print(f"bf16: {t_bf:.2f} ms, {db(n_bf):.2f} dB vs the oracle's crops (noise {n_bf * 1e6:.2f}e-6)")
        budget = 10 ** (-(a.gate + a.margin) / 10) - n_bf
        print(f"gate {a.gate} dB + margin {a.margin} dB -> noise budget for all fp8 parts together {budget * 1e6:.2f}e-6")
        rows = []
        print(f"{'part (ON alone)':52s} {'ms':>8s} {'saved':>7s} {'dB':>7s} {'noise added e-6':>16s} {'noise / ms':>11s}")
        for b, name in PARTS:
            t, n = run(1 << b)
            rows.append((b, name, t_bf - t, n - n_bf))
            print(f"{name:52s} {t:8.2f} {t_bf - t:7.2f} {db(n):7.2f} {(n - n_bf) * 1e6:16.2f} {(n - n_bf) * 1e6 / max(t_bf - t, 1e-3):11.2f}", flush=True)
        chosen, used = 0, 0.0
        for b, name, saved, added in sorted(rows, key=lambda r: max(r[3], 0.0) / max(r[2], 1e-3)):
            if saved <= 0.05:
                continue
            if used + max(added, 0.0) <= budget:
                chosen |= 1 << b
                used += max(added, 0.0)
        print(f"greedy choice: mask {chosen:#x} = {[n for b, n in PARTS if chosen >> b & 1]}; predicted noise {used * 1e6:.2f}e-6 of {budget * 1e6:.2f}e-6")
        all_bits = sum(1 << b for b, _ in PARTS)
        for label, m in (("chosen", chosen), ("all", all_bits), ("attention", 0b111), ("all but the encoder's convs", all_bits & ~(0x1f << 4))):
            t, n = run(m)
            print(f"{label:30s} mask {m:#8x}: {t:7.2f} ms ({t_bf - t:5.2f} saved), {db(n):.2f} dB vs oracle crops -> within 0.1 dB up to a {db(n) - 16.33:.1f} dB reference", flush=True)
    finally:
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, 0x5006), "ir_set_fp8_mask")   # back to IR_FP8_MASK_DEFAULT
        vae.enable_fp8(False)


if __name__ == "__main__":
    main()
