#!/usr/bin/env python3
"""Timing of the ControlLDM one-step path at the widths of configs/cldm.yaml: one ir_cldm_pipeline call per step
(SwinIR preprocess -> condition encoder -> ControlNet + SD-2.1 UNet at t = 999 -> first-stage decoder), seeded random weights, inputs
resident in HBM. Prints one JSON line in the shape of bench.py's (metric images/sec, per-kernel HIP-event rows); --verify compares the
last output with the fp32 CPU oracle (oracle/cldm.py + swinir / vae) on the same inputs.

    python tools/bench_cldm.py [--size 512] [--batch 1] [--steps 5] [--warmup 2] [--verify]

This is NOT the driver's bench (bench.py measures BASELINE.json's metric on the DiT path); it exists so that the N4 row has a measured
number and a kernel table next to its parity tests.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (random_state_dict, the SwinIR / VAE shape tables)

PEAK_BF16_TFLOPS, PEAK_HBM_GBS = 2500.0, 8000.0


def build(device, log):
    from instarevive_amd import weights as W
    from instarevive_amd.cldm import Reflow_ControlLDM
    t0 = time.time()
    cfg = dict(model_channels=320, channel_mult=(1, 2, 4, 4), num_res_blocks=2, attention_resolutions=(4, 2, 1), num_head_channels=64, context_dim=1024,
               in_channels=4, hint_channels=4, out_channels=4)
    swin_cfg = dict(embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8, mlp_ratio=2)
    sds = dict(unet=bench.random_state_dict(W.unet_shapes(cfg), 11), cnet=bench.random_state_dict(W.unet_shapes(cfg, True), 12),
               vae=bench.random_state_dict(W.vae_shapes(dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2)), 2),
               swin=bench.random_state_dict(W.swinir_shapes(swin_cfg), 1))
    unet_p = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                  channel_mult=[1, 2, 4, 4], num_head_channels=64, use_spatial_transformer=True, use_linear_in_transformer=True, transformer_depth=1,
                  context_dim=1024, use_checkpoint=True, legacy=False)
    ctrl_p = dict(unet_p, hint_channels=4)
    ctrl_p.pop("out_channels")
    m = Reflow_ControlLDM(control_stage_config=dict(params=ctrl_p), unet_config=dict(params=unet_p),
                          first_stage_config=dict(params=dict(ddconfig=dict(ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2))),
                          preprocess_config=dict(params=dict(img_size=64, patch_size=1, in_chans=3, embed_dim=180, depths=[6] * 8, num_heads=[6] * 8,
                                                             window_size=8, mlp_ratio=2, sf=8, img_range=1.0, upsampler="nearest+conv",
                                                             resi_connection="1conv", unshuffle=True, unshuffle_scale=8)))
    m.model.diffusion_model.load_state_dict(sds["unet"])
    m.control_model.load_state_dict(sds["cnet"])
    m.first_stage_model.load_state_dict(sds["vae"])
    m.preprocess_model.load_state_dict(sds["swin"], strict=False)
    m.to(device)
    log(f"models built and uploaded in {time.time() - t0:.1f}s "
        f"(UNet {sum(v.numel() for v in sds['unet'].values()) / 1e6:.0f} M, ControlNet {sum(v.numel() for v in sds['cnet'].values()) / 1e6:.0f} M parameters)")
    return m, sds, cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512, help="image edge (multiple of 64); the latent is size / 8")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--verify", action="store_true", help="compare the last output with the fp32 CPU oracle")
    ap.add_argument("--graph", action="store_true", help="replay the step as one hipGraph (IR_FLAG_GRAPH); no per-kernel rows")
    args = ap.parse_args()
    # clock / power trace (tools/power_sampler.py: a child that only reads sysfs), started before this process's first GPU call
    import subprocess
    import tempfile
    sampler_file = os.path.join(tempfile.gettempdir(), f"ir_power_cldm_{os.getpid()}.txt")
    try:
        sampler = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "power_sampler.py"), "--out", sampler_file, "--card", "0"], stdin=subprocess.PIPE,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except OSError:
        sampler = None
    if not torch.cuda.is_available():
        raise SystemExit("bench_cldm.py needs an MI355X GPU; the product path has no CPU fallback")
    device = torch.device("cuda", 0)

    def log(msg):
        print(f"[bench_cldm] {msg}", file=sys.stderr, flush=True)

    from instarevive_amd import _lib as L
    from instarevive_amd.cldm import _set_context
    m, sds, cfg = build(device, log)
    ctx = m.ctx
    n, h, w = args.batch, args.size, args.size
    g = torch.Generator().manual_seed(77)
    lq = torch.rand(n, 3, h, w, generator=g).to(device)
    zT = torch.randn(n, 4, h // 8, w // 8, generator=g).to(device)
    context = torch.randn(1, 77, 1024, generator=g) * 0.5
    _set_context(ctx, context)
    samples, control = torch.empty_like(lq), torch.empty_like(lq)
    ws = ctx.workspace(ctx.ws_bytes(L.STAGE_CLDM_PIPELINE, n, h, w))
    log(f"workspace {ws.numel() / 2 ** 20:.0f} MiB")

    def step():
        ctx.check(ctx.lib.ir_cldm_pipeline(ctx.h, ctx.stream(), L.ptr(lq), L.ptr(zT), L.ptr(samples), L.ptr(control), n, h, w, L.FLAG_GRAPH if args.graph else 0, 999.0, 0.18215,
                                           L.ptr(ws), ws.numel()), "ir_cldm_pipeline")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if not args.graph:
        ctx.profile_begin()
    wall0 = time.time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    power = None
    if sampler is not None:
        from tools.power_sampler import summarise
        time.sleep(0.05)
        power = summarise(sampler_file, wall0, wall0 + dt * args.steps)
        try:
            sampler.stdin.close()
            sampler.wait(timeout=2)
        except Exception:
            pass
        if power is not None:
            log(f"timed loop: gfx clock {power.get('clock_mhz')} MHz (min {power.get('clock_mhz_min')}, max {power.get('clock_mhz_max')}), socket power "
                f"{power.get('power_w')} W, {power.get('samples')} samples")
    kprof = ctx.profile_end_kernels() if not args.graph else {}
    total_ms = sum(v["ms"] for v in kprof.values())
    flops = sum(v["flops"] for v in kprof.values()) / args.steps
    per_kernel = {}
    for name, v in sorted(kprof.items(), key=lambda kv: -kv[1]["ms"]):
        short = name.split("/", 1)[1]
        row = dict(ms_per_step=round(v["ms"] / args.steps, 3), launches_per_step=v["launches"] // args.steps)
        if v["flops"] > 0:
            ach = v["flops"] / (v["ms"] / 1e3) / 1e12
            row.update(bound="mfma", tflop_per_step=round(v["flops"] / args.steps / 1e12, 4), achieved=round(ach, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                       frac=round(ach / PEAK_BF16_TFLOPS, 4))
        elif v["bytes"] > 0:
            ach = v["bytes"] / (v["ms"] / 1e3) / 1e9
            row.update(bound="hbm", gb_per_step=round(v["bytes"] / args.steps / 1e9, 3), achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                       frac=round(ach / PEAK_HBM_GBS, 4))
        per_kernel[short] = row
        log(f"  {short[:58]:58s} {row['ms_per_step']:8.3f} ms/step {row['launches_per_step']:4d} launches  "
            + (f"{row['achieved']:8.1f} {row['unit']} = {row['frac']:.3f}" if "frac" in row else ""))
    log(f"kernels {total_ms / args.steps:.2f} ms/step of {dt * 1e3:.2f} ms/step wall; {flops / 1e12:.2f} TFLOP algorithmic per step")
    verify = None
    if args.verify:
        from oracle import cldm as ocldm
        from oracle import swinir as oswin
        from oracle import vae as ovae
        t1 = time.time()
        swin_cfg = dict(embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8, mlp_ratio=2)
        ctl = oswin.swinir_forward(sds["swin"], lq.cpu(), swin_cfg)
        c_latent = ovae.vae_encode_mean(sds["vae"], ctl * 2 - 1) * 0.18215
        sd = {**{"model.diffusion_model." + k: v for k, v in sds["unet"].items()}, **{"control_model." + k: v for k, v in sds["cnet"].items()}}
        z = ocldm.reflow_sample(sd, zT.cpu(), c_latent, context.expand(n, -1, -1), cfg)
        ref = (ovae.vae_decode(sds["vae"], z / 0.18215) + 1) / 2
        got = samples.cpu()
        mse = float(((got.double() - ref.double()) ** 2).mean())
        psnr = 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)
        rel = float((got - ref).norm() / ref.norm())
        verify = dict(verified=bool(psnr >= 35.0 and float(got.std()) > 1e-3), psnr_vs_fp32_oracle_db=round(psnr, 2), rel_l2_vs_fp32_oracle=round(rel, 5),
                      oracle_seconds=round(time.time() - t1, 1), oracle_images_per_sec=round(n / (time.time() - t1), 4), oracle_threads=torch.get_num_threads())
        log(f"verify: {psnr:.2f} dB, rel-L2 {rel:.4f} against the fp32 oracle ({time.time() - t1:.1f} s on {torch.get_num_threads()} threads)")
    dom = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_step"]) if per_kernel else None
    line = {"metric": "ControlLDM one-step restoration images/sec", "value": round(n / dt, 3), "unit": "images/sec", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{h}x{w} LQ -> SwinIR -> condition encoder -> ControlNet + SD-2.1 UNet (t=999, 77x1024 context) -> decoder, batch {n}",
                       "weights": "seeded random, configs/cldm.yaml widths"},
            "algorithmic_tflop_per_step": round(flops / 1e12, 3), "path_tflops": round(flops / dt / 1e12, 1),
            "roofline": dict(per_kernel[dom], kernel=dom, per_kernel=per_kernel) if dom else None, "graph": bool(args.graph),
            "clock_mhz": power.get("clock_mhz") if power else None, "power_w": power.get("power_w") if power else None}
    if verify:
        line.update(verify)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
