#!/usr/bin/env python3
"""Which attention raises the fixed-reference overflow flag when the softmax rows get peaky, and what does its fallback cost?
bench.py's seeded full-size models at 2048 x 2048; the q / k projections of ONE attention family at a time (VAE encoder mid block, VAE decoder mid
block, the 28 DiT blocks) scaled so that its logits grow by the given factor; per configuration: ms per ir_pipeline call (median of 3) and the
number of attention launches that took the rescaling fallback (ir_attn_fallback_count).

    python tools/peaky_probe.py [--gains 4 16 64]"""
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
    ap.add_argument("--gains", type=float, nargs="+", default=[4.0, 16.0, 64.0])
    ap.add_argument("--fp8_vae_attention", action="store_true", help="run the VAE mid-block attention parts on fp8 operands (ir_set_fp8_mask 0b110): what does ITS "
                    "fallback cost (since round 5: the bf16 one-wave-per-SIMD kernel, not the 4-wave one), and does the result equal the bf16 path's")
    a = ap.parse_args()
    from instarevive_amd import _lib as L
    from tests.support.stress_weights import stress_state_dicts
    dev = torch.device("cuda", 0)
    swin, vae, dit, sched, sds = bench.build_models(dev, lambda m: None)
    ctx = dit.ctx
    y, mask = bench.synthetic_prompt()
    yd, md = y.to(dev), mask.to(dev)
    S = 2048
    din = bench.upscale_bicubic(bench.synthetic_lq(1, 512, 512, 1000), 4.0).to(dev)
    dout = torch.empty_like(din)
    acp, sf = float(sched.alphas_cumprod[400]), float(vae.config.scaling_factor)
    ws = ctx.workspace(ctx.ws_bytes(L.STAGE_PIPELINE, 1, S, S, L.FLAG_FP8 if a.fp8_vae_attention else 0, 512, 448))
    fl = [0]
    if a.fp8_vae_attention:
        vae.enable_fp8(True)
        ctx.check(ctx.lib.ir_set_fp8(ctx.h, 0), "ir_set_fp8")
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, 0b110), "ir_set_fp8_mask")

    def psnr(x, y):
        mse = float(((x.double() - y.double()) ** 2).mean())
        return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)

    def run():
        dit.set_prompt(yd, md)
        dit.ensure_pos(S // 16, S // 16)
        times = []
        ctx.check(ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), 1), "count")
        for i in range(4):
            if i == 1:
                fb = ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), 0)
                ctx.check(ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), -1), "count")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.check(ctx.lib.ir_pipeline(ctx.h, ctx.stream(), L.ptr(din), L.ptr(dout), None, 1, S, S, fl[0], 512, 448, 400.0, acp, sf, L.ptr(ws), ws.numel()), "ir_pipeline")
            e1.record()
            torch.cuda.synchronize()
            if i:
                times.append(e0.elapsed_time(e1))
        return float(np.median(times)), fb

    t0, f0 = run()
    print(f"seeded weights (flat softmax): {t0:.2f} ms, {f0} fallbacks")
    if a.fp8_vae_attention:
        fl[0] = L.FLAG_FP8
        t8, f8 = run()
        print(f"the same with the VAE attention parts on fp8 operands: {t8:.2f} ms ({t8 - t0:+.2f}), {f8} fallbacks")
    one = {"dit": [1.0] * 28, "vae_encoder": 1.0, "vae_decoder": 1.0}
    for fam in ("vae_encoder", "vae_decoder", "dit"):
        for g in a.gains:
            gains = dict(one)
            gains[fam] = [g] * 28 if fam == "dit" else g
            st = stress_state_dicts(sds, frac=0.0, gain=1.0, logit_gain=gains)
            (vae if fam != "dit" else dit).load_state_dict(st["vae" if fam != "dit" else "dit"])
            if a.fp8_vae_attention:   # (load_state_dict re-packed the fp8 weight forms: enable_fp8 stays on)
                ctx.check(ctx.lib.ir_set_fp8(ctx.h, 0), "ir_set_fp8")
                fl[0] = 0
                tb, _ = run()
                ref = dout.clone()
                fl[0] = L.FLAG_FP8
            t, f = run()
            extra = f"; fp8 VAE attention: {t - tb:+.2f} ms against the bf16 pass of the same weights, {psnr(dout, ref):.2f} dB against it" if a.fp8_vae_attention else ""
            print(f"{fam:12s} logits x{g:<4g}: {t:7.2f} ms ({t - t0:+6.2f}), {f} attention launch(es) took the fallback{extra}", flush=True)
        (vae if fam != "dit" else dit).load_state_dict(sds["vae" if fam != "dit" else "dit"])


if __name__ == "__main__":
    main()
