#!/usr/bin/env python3
"""Fit and check the rule of instarevive_amd/fp8_select.py (--fp8 auto) against the fp32 oracle - test infrastructure, the product never runs this.

For the seeded weights and for the stress weights (tests/support/stress_weights.py with the gains of tests/golden/stress_512.npz), at 512 x 512:
  * the float-domain noise of the bf16 path against the oracle's decoded floats (what BF16_SHARE reserves of the 0.1 dB allowance);
  * every part's cost as the product measures it (fp8 part alone vs the bf16 pass, floats) next to what it costs against the oracle;
  * the operand set the rule chooses for a range of BF16_SHARE values, and the uint8 PSNR of the bf16 path and of that set against the oracle.

    python tools/fp8_auto_calib.py > gpurun_out/r06_fp8_auto_calib.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from instarevive_amd import _lib as L  # noqa: E402
from instarevive_amd import fp8_select as F  # noqa: E402
from instarevive_amd.pipeline import process  # noqa: E402


def psnr_u8(a, b):
    mse = float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean()) / 255.0 ** 2
    return 10.0 * np.log10(1.0 / (mse + 1e-12))


def oracle_float(sds, img, y, mask):
    from oracle import dit as odit, glue as oglue, swinir as oswin, vae as ovae
    x = torch.from_numpy(img).permute(2, 0, 1)[None].float() / 255.0
    control = oswin.swinir_forward(sds["swin"], x)
    lat = ovae.vae_encode_mean(sds["vae"], control * 2 - 1) * 0.18215
    x0 = oglue.generate_sample_1step(lambda l, t, yy, mm: odit.dit_forward(sds["dit"], l, t, yy, mm), oglue.alphas_cumprod_diffusers(), lat, 400, y, mask)
    dec = ovae.vae_decode(sds["vae"], x0 / 0.18215) / 2 + 0.5
    u8 = (dec.clamp(0, 1).permute(0, 2, 3, 1) * 255).numpy().clip(0, 255).astype(np.uint8)[0]   # no colour fix: the quantity the rule is about
    return dec, u8


def main():
    dev = torch.device("cuda", 0)
    swin, vae, dit, sched, sds = bench.build_models(dev, lambda m: None)
    y, mask = bench.synthetic_prompt()
    yd, md = y.to(dev), mask.to(dev)
    ctx = dit.ctx
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    from tests.support.stress_weights import stress_state_dicts
    z = np.load(os.path.join(ROOT, "tests", "golden", "stress_512.npz"))
    gains = {"dit": [float(v) for v in z["logit_gain_dit"]], "vae_encoder": float(z["logit_gain_vae"][0]), "vae_decoder": float(z["logit_gain_vae"][1])}
    sets = {"seeded": sds, "stress": stress_state_dicts(sds, float(z["frac"]), float(z["gain"]), gains)}
    img = F.calibration_image()
    x = torch.from_numpy(img).to(dev).permute(2, 0, 1)[None].float() / 255.0
    for name, sd in sets.items():
        vae.load_state_dict(sd["vae"])
        dit.load_state_dict(sd["dit"])
        dit.invalidate_prompt()
        ref_f, ref_u8 = oracle_float(sd, img, y, mask)
        vae.enable_fp8(False)
        _, base = F._decoded(swin, vae, dit, yd, md, x)
        to_u8 = lambda t: (t.clamp(0, 1).permute(0, 2, 3, 1) * 255).cpu().numpy().clip(0, 255).astype(np.uint8)[0]
        n_bf = float(((base.cpu().double() - ref_f.double()) ** 2).mean())
        print(f"== {name} weights, calibration image {img.shape[0]} x {img.shape[1]}")
        print(f"bf16 path vs the fp32 oracle: float-domain noise {n_bf * 1e6:.2f}e-6 = {n_bf / F.ALLOWANCE:.2f} of the 0.1 dB-at-30 dB allowance ({F.ALLOWANCE * 1e6:.2f}e-6); "
              f"uint8 {psnr_u8(to_u8(base), ref_u8):.2f} dB")
        costs = F.measure_parts(swin, vae, dit, yd, md)
        vae.enable_fp8(True)
        print(f"{'part ON alone':40s} {'vs bf16 (e-6, product)':>24s} {'vs oracle added (e-6)':>22s} {'uint8 vs oracle dB':>20s}")
        for bit, pname, _ in F.PARTS:
            ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, 1 << bit), "mask")
            _, got = F._decoded(swin, vae, dit, yd, md, x)
            n = float(((got.cpu().double() - ref_f.double()) ** 2).mean())
            print(f"{pname:40s} {costs[bit] * 1e6:24.2f} {(n - n_bf) * 1e6:22.2f} {psnr_u8(to_u8(got), ref_u8):20.2f}")
        for share in (0.5, 0.6, 0.7, 0.75, 0.8, 0.9):
            m, used = F.choose(costs, F.ALLOWANCE * (1 - share))
            ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, m if m else 1 << 31), "mask")
            _, got = F._decoded(swin, vae, dit, yd, md, x)
            n = float(((got.cpu().double() - ref_f.double()) ** 2).mean())
            saved = sum(s for b, _, s in F.PARTS if m >> b & 1)
            print(f"BF16_SHARE {share:.2f}: mask {m:#8x} ({saved:5.2f} ms saved at 2048 x 2048), predicted noise {used * 1e6:.2f}e-6, measured vs oracle added {(n - n_bf) * 1e6:.2f}e-6; "
                  f"uint8 {psnr_u8(to_u8(got), ref_u8):.2f} dB (bf16 {psnr_u8(to_u8(base), ref_u8):.2f})")
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, L.FP8_MASK_DEFAULT), "mask")
        vae.enable_fp8(False)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
