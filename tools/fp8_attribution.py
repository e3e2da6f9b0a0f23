#!/usr/bin/env python3
"""Which parts of the fp8 operand set (BASELINE.json configs[4]) carry its error? Full-size architectures, bench.py's seeded weights, one
512 x 512 image (and optionally a 1024 x 1024 one through the headline fixture): the uint8 result of the HIP path against the fp32 oracle's
with every part's bit of ir_set_fp8_mask switched ON alone and OFF alone.

    python tools/fp8_attribution.py            # prints a table; DESIGN.md section 4 quotes it

PSNR against the oracle -> the reference quality up to which the path stays within north_star's 0.1 dB: P_err - 16.33 dB
(tests/support/psnr_guard.py). The oracle is the checker here (test infrastructure), as in bench.py's cpu_baseline leg."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from instarevive_amd.pipeline import process  # noqa: E402

PARTS = [(0, "DiT self-attention"), (1, "VAE encoder mid attention"), (2, "VAE decoder mid attention"),
         (4, "encoder level 0 convs (full resolution, 128 ch)"), (5, "encoder level 1 convs"), (6, "encoder level 2 convs"), (7, "encoder level 3 convs"),
         (8, "encoder mid-block convs"), (12, "decoder level 0 convs (full resolution, 128 ch)"), (13, "decoder level 1 convs"),
         (14, "decoder level 2 convs"), (15, "decoder level 3 convs"), (16, "decoder mid-block convs")]


def psnr(a, b):
    mse = float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean()) / 255.0 ** 2
    return 10.0 * np.log10(1.0 / (mse + 1e-8))


def main():
    from oracle import dit as odit, glue as oglue, swinir as oswin, vae as ovae
    dev = torch.device("cuda", 0)
    swin, vae, dit, sched, sds = bench.build_models(dev, lambda m: None)
    y, mask = bench.synthetic_prompt()
    yd, md = y.to(dev), mask.to(dev)
    img = bench.synthetic_lq(1, 512, 512, 15)[0].numpy()
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    ref, _ = oglue.process([img], lambda x: oswin.swinir_forward(sds["swin"], x), lambda x: ovae.vae_encode_mean(sds["vae"], x),
                           lambda lat, t, yy, mm: odit.dit_forward(sds["dit"], lat, t, yy, mm), lambda z: ovae.vae_decode(sds["vae"], z),
                           oglue.alphas_cumprod_diffusers(), y, mask)
    kw = dict(preprocess_model=swin, vae=vae, y=yd, y_mask=md, noise_scheduler=sched)
    ctx = dit.ctx
    bf = process(dit, [img], 1, "wavelet", False, False, 512, 448, **kw)[0][0]
    p_bf = psnr(bf, ref[0])
    print(f"bf16 path: {p_bf:.2f} dB vs fp32 oracle (within 0.1 dB up to a reference quality of {p_bf - 16.33:.1f} dB)")
    vae.enable_fp8(True)

    def run(mask_bits):
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, mask_bits), "ir_set_fp8_mask")
        out = process(dit, [img], 1, "wavelet", False, False, 512, 448, fp8=True, **kw)[0][0]
        return psnr(out, ref[0]), psnr(out, bf)

    all_bits = sum(1 << b for b, _ in PARTS)
    try:
        p_all, q_all = run(all_bits)
        print(f"fp8, all parts:  {p_all:.2f} dB vs oracle, {q_all:.2f} dB vs the bf16 path (within 0.1 dB up to {p_all - 16.33:.1f} dB)")
        print(f"{'part':52s} {'ON alone: vs oracle / vs bf16':>32s} {'OFF alone: vs oracle':>22s}")
        rows = []
        for b, name in PARTS:
            on, on_bf = run(1 << b)
            off, _ = run(all_bits & ~(1 << b))
            rows.append((b, name, on, on_bf, off))
            print(f"{name:52s} {on:14.2f} / {on_bf:6.2f} dB {off:18.2f} dB", flush=True)
        # operand sets of interest: attention only; convs only; everything but the full-resolution conv levels
        attn = (1 << 0) | (1 << 1) | (1 << 2)
        for label, m in (("attention only (DiT + both VAE mid blocks)", attn), ("VAE convs only", all_bits & ~attn),
                         ("all but encoder level 0", all_bits & ~(1 << 4)), ("all but encoder level 0 + decoder level 0", all_bits & ~((1 << 4) | (1 << 12))),
                         ("all but the whole encoder's convs", all_bits & ~(0x1f << 4))):
            p, q = run(m)
            print(f"{label:52s} {p:14.2f} / {q:6.2f} dB   (within 0.1 dB up to {p - 16.33:.1f} dB)", flush=True)
    finally:
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, 0x5006), "ir_set_fp8_mask")   # back to IR_FP8_MASK_DEFAULT
        vae.enable_fp8(False)


if __name__ == "__main__":
    main()
