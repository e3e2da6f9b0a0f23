#!/usr/bin/env python3
"""Headline benchmark: 512 -> 2048 one-step super-resolution images/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step is ONE pass of the hot path — process() of test_scripts/inference.py:55-166: SwinIR -> VAE encode -> one
PixArt-DiT step at t=400 (x0 from eps) -> VAE decode -> uint8 — over one batch of synthetic LQ images per rank, through
the fused C-ABI entry ir_pipeline with input and output resident in HBM. Workload = BASELINE.json configs[1]:
512x512 LQ, sr_scale 4 => 2048x2048 network input, untiled, batch 1 per GPU, full-size architectures (SwinIR 15.8 M,
VAE 83.7 M, DiT 611 M parameters) with seeded random weights (no checkpoints exist offline; throughput is weight-
independent). Images are independent, so ranks shard the batch with no data-path collective (weak scaling).

Also printed on the same JSON line:
  roofline     : the dominant kernel class (by GPU time) measured with HIP events on the launch stream during the timed
                 steps (ir_profile_begin/end), algorithmic FLOPs / time vs the dense bf16 MFMA peak (2.5 PFLOP/s);
  cpu_baseline : the oracle (CPU fp32 restatement of the reference, kind "port") timed on this box's host cores on a
                 bounded sample (one 512x512 network pass), extrapolated to the 2048x2048 workload by algorithmic FLOPs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md, chip-level table)
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 (MX-scaled f8f6f4 MFMA), same table
PEAK_HBM_GBS = 8000.0
# What a BARE bf16 MFMA loop on random operands delivers on this pool's chips once the card sits at its power-limited clock (tools/mfma_power_probe.hip,
# profiles/r06_mfma_power_probe.txt: 1.70-1.72 PFLOP/s whatever the MFMA shape - 32x32x16 at 1.71 GHz, 16x16x32 at 2.12 GHz and a lower issue rate). NOT the
# roofline's peak (that stays the guide's 2.5 PFLOP/s): a second yardstick in the line, `frac_of_power_limited_mfma`, for MFMA-bound bf16 kernels.
POWER_LIMITED_BF16_TFLOPS = 1700.0


def pmc_file_for_this_tree():
    """The newest profiles/r*_pmc_kernels.json collected on THESE kernel sources (its csrc_sha16 = instarevive_amd.build.source_hash()), else
    (None, why): roofline.traffic must come from the evidence run of the same code, a stale file is refused."""
    import glob
    from instarevive_amd.build import source_hash
    cur = source_hash()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_kernels.json")), reverse=True)
    seen = []
    for f in files:
        try:
            sha = json.load(open(f)).get("csrc_sha16")
        except (OSError, ValueError):
            continue
        if sha == cur:
            return f, None
        seen.append(f"{os.path.basename(f)} ({sha or 'no hash'})")
    return None, f"no PMC file for kernel sources {cur}: {', '.join(seen[:2]) or 'none found'} - stale, refused; run tools/r06_evidence.sh"


# ---------------------------------------------------------------- algorithmic FLOP model (BASELINE.md section 2, 2*MAC, matmul/conv only)
def flops_model(h, w, n_tok=300, copies=0):
    px = h * w
    swin = 0.1815e12 / (512 * 512) * px                       # linear in pixels
    t_vae = px // 64
    enc = (1.1167e12 - 4 * 4096 ** 2 * 512) / (512 * 512) * px + 4.0 * t_vae ** 2 * 512
    dec = (2.5145e12 - 4 * 4096 ** 2 * 512) / (512 * 512) * px + 4.0 * t_vae ** 2 * 512
    t = px // 256
    C, L = 1152, 28
    dit = (14 * C * C * 2 * L + 4 * n_tok * C * L) * t + (n_tok * 2 * C * C * 2 * L + n_tok * (4096 * C + C * C) * 2) + 4.0 * t * t * C * L \
        + t * (16 * C + 32 * C) * 2
    if copies:  # ControlNet-Half: `copies` more blocks (with their prompt K/V), after_proj each, before_proj, patch embedding of c
        dit += (14 * C * C * 2 + 4 * n_tok * C) * copies * t + n_tok * 2 * C * C * 2 * copies + 4.0 * t * t * C * copies \
            + (copies + 1) * 2 * C * C * t + t * 16 * C * 2
    return dict(swinir=swin, vae_encode=enc, dit=dit, vae_decode=dec, total=swin + enc + dit + dec)


def upconv_phase_saving(h, w):
    """FLOPs the three decoder Upsample convs do NOT execute in their sub-pixel phase form (conv_halo_s1_kernel<0,4>: four 2x2 convs on the
    low-resolution tensor, 16 of the 9-tap form's 36 tap products per source pixel). roofline.achieved stays on the reference's algorithmic
    9-tap FLOPs (the contract's definition); the kernel row reports the executed rate beside it."""
    px = h * w
    algorithmic = 2 * 9 * (px // 16 * 512 * 512 + px // 4 * 512 * 512 + px * 256 * 256)
    return algorithmic * 5 / 9


def flops_model_tiled(h, w, tile=512, stride=448, n_tok=300, copies=0):
    """--tiled: SwinIR and the VAE encoder run on the whole image, the DiT step and the decoder once per tile (inference.py:119-153)."""
    def starts(size):
        v = list(range(0, size - tile + 1, stride))
        return len(v) + (1 if (size - tile) % stride else 0)
    nt = starts(h) * starts(w)
    full, per = flops_model(h, w, n_tok, copies), flops_model(tile, tile, n_tok, copies)
    out = dict(swinir=full["swinir"], vae_encode=full["vae_encode"], dit=nt * per["dit"], vae_decode=nt * per["vae_decode"], tiles=nt)
    out["total"] = out["swinir"] + out["vae_encode"] + out["dit"] + out["vae_decode"]
    return out


def conv_flops_model(h, w):
    """Algorithmic FLOPs of the 3x3-convolution class alone (what the `conv3x3` kernel class executes, without padding)."""
    px = h * w
    # SwinIR: conv_first 192->180, 8 RSTB convs + conv_after_body 180->180 (at 1/64 res), before_up 180->64, up1..3, hr, last
    g = px / 64
    swin = 2 * 9 * (g * (192 * 180 + 9 * 180 * 180 + 180 * 64) + 64 * 64 * (4 * g + 16 * g + 64 * g + 64 * g) + 64 * 3 * 64 * g)
    # VAE convs (incl. conv_in / conv_out); resnet shortcuts and attention projections are 1x1 -> `linear` class
    def res(cin, cout, p):
        return 2 * 9 * p * (cin * cout + cout * cout)
    enc = 2 * 9 * px * 3 * 128 + res(128, 128, px) * 2 + 2 * 9 * (px / 4) * 128 * 128 + res(128, 256, px / 4) + res(256, 256, px / 4) + \
        2 * 9 * (px / 16) * 256 * 256 + res(256, 512, px / 16) + res(512, 512, px / 16) + 2 * 9 * (px / 64) * 512 * 512 + \
        res(512, 512, px / 64) * 4 + 2 * 9 * (px / 64) * 512 * 8
    dec = 2 * 9 * (px / 64) * 4 * 512 + res(512, 512, px / 64) * 5 + 2 * 9 * (px / 16) * 512 * 512 + res(512, 512, px / 16) * 3 + \
        2 * 9 * (px / 4) * 512 * 512 + res(512, 256, px / 4) + res(256, 256, px / 4) * 2 + 2 * 9 * px * 256 * 256 + res(256, 128, px) + \
        res(128, 128, px) * 2 + 2 * 9 * px * 128 * 3
    return swin + enc + dec


# ---------------------------------------------------------------- synthetic weights in the reference checkpoints' key layout
def random_state_dict(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        leaf = name.rsplit(".", 1)[-1]
        if "table" in name:
            t = (torch.rand(shape, generator=g) - 0.5) * 0.4
        elif leaf == "bias":
            t = (torch.rand(shape, generator=g) - 0.5) * 0.1
        elif len(shape) == 1:
            t = 1.0 + (torch.rand(shape, generator=g) - 0.5) * 0.2
        else:
            fan_in = int(np.prod(shape[1:]))
            t = (torch.rand(shape, generator=g) - 0.5) * 2 * (3.0 / fan_in) ** 0.5
        sd[name] = t
    return sd


def build_models(device, log, control=0):
    from instarevive_amd.models import AutoencoderKL, ControlTransformerHalf, DDPMScheduler, SwinIR, Transformer2DModel
    from instarevive_amd import weights as W
    t0 = time.time()
    swin_cfg = dict(embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8, mlp_ratio=2)
    sds = dict(swin=random_state_dict(W.swinir_shapes(swin_cfg), 1),
               vae=random_state_dict(W.vae_shapes(dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2)), 2),
               dit=random_state_dict(W.dit_shapes(dict(num_layers=28, num_attention_heads=16, attention_head_dim=72, caption_channels=4096)), 3))
    swin = SwinIR(img_size=64, patch_size=1, in_chans=3, embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8, mlp_ratio=2, sf=8,
                  img_range=1.0, upsampler="nearest+conv", resi_connection="1conv", unshuffle=True, unshuffle_scale=8)
    swin.load_state_dict(sds["swin"], strict=False)
    vae = AutoencoderKL()
    vae.load_state_dict(sds["vae"])
    dit = Transformer2DModel()
    dit.load_state_dict(sds["dit"])
    for m in (swin, vae, dit):
        m.to(device)
    if control:  # ControlNet-Half branch with seeded non-zero projections (zero-initialised ones would still run the same kernels)
        ctl = ControlTransformerHalf(dit, copy_blocks_num=control)
        g = torch.Generator().manual_seed(4)
        ctl.load_state_dict(dict({"base_model." + k: v for k, v in sds["dit"].items()},
                                 **{k: (v if "copied_block" in k else (torch.rand(v.shape, generator=g) - 0.5) * 0.05) for k, v in ctl._sd.items()}))
    log(f"models built and uploaded in {time.time() - t0:.1f}s")
    return swin, vae, dit, DDPMScheduler(), sds


def synthetic_prompt(seed=1234, n_tok=300, valid=25):
    g = torch.Generator().manual_seed(seed)
    y = torch.randn(1, n_tok, 4096, generator=g) * 0.1
    mask = torch.zeros(1, 1, n_tok)
    mask[..., :valid] = 1
    return y, mask


def synthetic_lq(n, h, w, seed):
    """uint8 HWC images: uniform noise low-passed by a 3x3 box (image-like spectrum), as SURVEY.md section 8(d) prescribes."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 3, h, w, generator=g)
    x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="replicate"), 3, stride=1)
    return (x.permute(0, 2, 3, 1) * 255).to(torch.uint8).contiguous()


def upscale_bicubic(imgs_u8, scale):
    from PIL import Image
    out = []
    for im in imgs_u8.numpy():
        pil = Image.fromarray(im)
        out.append(np.array(pil.resize((int(np.ceil(pil.size[0] * scale)), int(np.ceil(pil.size[1] * scale))), Image.BICUBIC)))
    return torch.from_numpy(np.stack(out))


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sds, y, mask, h_full, w_full, log, hip_fn=None, big_pass=True):
    """Oracle (kind 'port') on the host cores: one 512x512 network pass and - the bounded sample the figure is taken from - one 1024x1024
    pass (22.7 TFLOP, 20-30 s on the GPU box's 16-thread share), extrapolated to the headline size by algorithmic FLOPs; the 512 pass
    validates that extrapolation one size down (`flop_scaling_check`). hip_fn(img) -> uint8 image: the timed HIP path on the 512x512
    sample, so that the line also carries its PSNR against the oracle (the oracle is the CHECKER here, never the thing measured as GPU
    work) and the reference quality up to which that error stays within north_star's 0.1 dB (tests/support/psnr_guard.py)."""
    from oracle import dit as odit, glue as oglue, swinir as oswin, vae as ovae
    # a GPU box gives this job a CPU share (16 cores per GPU); os.cpu_count() reports the whole host
    cores = int(os.environ.get("IR_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
    torch.set_num_threads(cores)

    def one(hs):
        img = synthetic_lq(1, hs, hs, 77).numpy()
        t0 = time.time()
        preds, _ = oglue.process([img[0]], lambda x: oswin.swinir_forward(sds["swin"], x), lambda x: ovae.vae_encode_mean(sds["vae"], x),
                                 lambda lat, t, yy, mm: odit.dit_forward(sds["dit"], lat, t, yy, mm), lambda z: ovae.vae_decode(sds["vae"], z),
                                 oglue.alphas_cumprod_diffusers(), y, mask)
        return time.time() - t0, img[0], preds[0]

    dt5, img5, ref5 = one(512)
    f5, f_full = flops_model(512, 512)["total"], flops_model(h_full, w_full)["total"]
    log(f"cpu baseline: 512x512 pass {dt5:.1f}s on {cores} threads ({f5 / dt5 / 1e12:.3f} TFLOP/s)")
    out = dict(unit="images/sec", cores=cores, kind="port", cpu_model=cpu_model_name())
    if big_pass:
        dt10, _, _ = one(1024)
        f10 = flops_model(1024, 1024)["total"]
        log(f"cpu baseline: 1024x1024 pass {dt10:.1f}s ({f10 / dt10 / 1e12:.3f} TFLOP/s); predicted from the 512 pass by FLOPs {dt5 * f10 / f5:.1f}s")
        out.update(value=1.0 / (dt10 * f_full / f10),
                   sample=f"one 1024x1024 network pass of the fp32 oracle ({f10 / 1e12:.2f} of {f_full / 1e12:.2f} TFLOP), {dt10:.2f} s, extrapolated to "
                          f"{h_full}x{w_full} by algorithmic FLOPs",
                   flop_scaling_check=dict(pass_512_s=round(dt5, 2), pass_1024_s=round(dt10, 2), predicted_1024_from_512_s=round(dt5 * f10 / f5, 2),
                                           measured_over_predicted=round(dt10 / (dt5 * f10 / f5), 3)))
    else:
        out.update(value=1.0 / (dt5 * f_full / f5),
                   sample=f"one 512x512 network pass of the fp32 oracle ({f5 / 1e12:.2f} of {f_full / 1e12:.2f} TFLOP), {dt5:.2f} s, "
                          f"extrapolated to {h_full}x{w_full} by algorithmic FLOPs")
    if hip_fn is not None:
        got = hip_fn(img5)
        mse = float(((got.astype(np.float64) - ref5.astype(np.float64)) ** 2).mean()) / 255.0 ** 2
        p_err = 10.0 * np.log10(1.0 / (mse + 1e-8))
        cross = p_err + 10.0 * np.log10(10 ** 0.01 - 1.0)   # independent error: dPSNR = 10 log10(1 + 10^((P_ref - P_err) / 10)) = 0.1 dB
        out["parity_512"] = dict(psnr_vs_oracle_db=round(p_err, 2), within_0p1_db_up_to_reference_psnr_db=round(cross, 1),
                                 note="uint8 result of the timed HIP path on the 512x512 sample against the fp32 oracle's; a path with this error "
                                      "moves PSNR(., GT) by <= 0.1 dB as long as the reference itself scores at most the second figure")
        log(f"parity on the 512x512 sample: {p_err:.2f} dB vs oracle; within 0.1 dB of the reference's PSNR up to a reference quality of {cross:.1f} dB")
    return out


def cli_files_leg(k, sds, value, log):
    """`inference.py --sr_scale 4` as a fresh child process over k synthetic 512 x 512 PNG files with the bench's own full-size weights written
    in the reference's file formats (tools/cli_artifacts.py). Never raises: a failed leg is reported as {"error": ...}."""
    import shutil
    import subprocess
    import tempfile
    from tools import cli_artifacts as A
    d = tempfile.mkdtemp(prefix="ir_cli_")
    try:
        t0 = time.time()
        flags = A.write_full_artifacts(d, sds)
        A.write_lq_pngs(os.path.join(d, "in"), k)
        log(f"cli leg: artefacts + {k} PNGs written in {time.time() - t0:.1f}s under {d}")
        cmd = [sys.executable, os.path.join(ROOT, "inference.py"), "--input", os.path.join(d, "in"), "--output", os.path.join(d, "out"), "--sr_scale", "4"] + flags
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
        rates = A.parse_cli_rate(r.stdout)
        written = len([f for f in os.listdir(os.path.join(d, "out"))]) if os.path.isdir(os.path.join(d, "out")) else 0
        if r.returncode != 0 or not rates or written != k:
            return dict(error=f"rc {r.returncode}, {written} of {k} files written", tail=(r.stdout[-300:] + r.stderr[-600:]))
        c = rates[0]
        log(f"cli leg: {c['files']} files in {c['seconds']:.2f}s = {c['files_per_s']:.2f} files/s; after the first result {c['steady_files_per_s']:.2f} files/s "
            f"({c['workers']} host threads) against {value:.2f} images/s device-resident")
        return dict(files=k, files_per_s=c["files_per_s"], steady_files_per_s=c["steady_files_per_s"], result_rate=c.get("result_rate"), host_threads=c["workers"],
                    ratio_to_value=round(c["files_per_s"] / value, 3), steady_ratio_to_value=round(c["steady_files_per_s"] / value, 3),
                    command="inference.py --sr_scale 4 (child process; 512x512 PNG in, 2048x2048 PNG out, default --workers)",
                    note="files_per_s: first read submitted -> last PNG closed, including the child's library / workspace warm-up on its first image; "
                         "steady_*: from the first finished result on (the last files' PNG encoding - about 1 s of drain after the GPU has finished - is "
                         "inside both: a longer folder amortises it); result_rate: finished results per second between the first and the last one, i.e. "
                         "what the GPU side of the stream delivers. Model loading is excluded from all")
    except Exception as e:  # noqa: BLE001 - the headline line must still be printed
        return dict(error=repr(e)[:300])
    finally:
        shutil.rmtree(d, ignore_errors=True)


def self_launch(n_ranks):
    """Run this script as n_ranks child processes under torch.distributed.run (one rank per GPU) and return the exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:   # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--lq", type=int, default=512, help="LQ edge (pixels)")
    ap.add_argument("--sr_scale", type=float, default=4.0)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step")
    ap.add_argument("--tiled", action="store_true")
    ap.add_argument("--net_hw", type=str, default="", help="HxW network input (overrides --lq/--sr_scale), e.g. 2176x3840 for the padded 4K case")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--fp8_parts", choices=["default", "qualified", "all", "attention", "no_encoder_convs", "convs"], default="default",
                    help="with --fp8: which parts take e4m3 operands. default = chosen ON THE LOADED WEIGHTS by instarevive_amd/fp8_select.py (one 512 x 512 calibration "
                         "image; north_star's 0.1 dB budget): on the bench's seeded weights the qualified set (the attention parts + the decoder's level-0 / level-2 "
                         "convs, >= 46.3 dB against the fp32 oracle at 2048 x 2048); the line also carries what the same rule chooses on the stress weights and "
                         "what that costs (fp8_auto). qualified = that set without calibration; all = every part BASELINE.json configs[4] names plus the VAE "
                         "mid-block attention: faster, but OUTSIDE the 0.1 dB tolerance above a 25.8 dB reference")
    ap.add_argument("--cpu_small", action="store_true", help="CPU baseline from the 512x512 oracle pass only (skips the 1024x1024 pass, about 25 s)")
    ap.add_argument("--control", type=int, default=0, metavar="COPIES", help="diagnostic: run the DiT step with the ControlNet-Half branch "
                    "(COPIES copied blocks, 13 in the reference configs; c = the LQ latent). Not the headline workload: no such weights are released")
    ap.add_argument("--graph", action="store_true", help="diagnostic: replay the step as one hipGraph (IR_FLAG_GRAPH); implies --no_profile, "
                    "so the JSON line carries no roofline")
    ap.add_argument("--profile_all", action="store_true", help="one pass: every launch of the timed loop bracketed by events (the form before round 4; costs about 1.4 %)")
    ap.add_argument("--no_profile", action="store_true", help="experiment: time the loop without the per-launch HIP events (no roofline)")
    ap.add_argument("--no_verify", action="store_true", help="skip the fast-vs-plain-kernel check of the last timed output")
    ap.add_argument("--no_host_rate", action="store_true", help="skip the host-buffer (PCIe-inclusive) rates")
    ap.add_argument("--logit_gain", type=float, nargs="*", default=[], metavar="G", help="diagnostic leg after the timed steps: every self-attention logit (DiT blocks, "
                    "VAE mid blocks) times G - q and k projections scaled by sqrt(G), tests/support/stress_weights.py - and the same steps timed again, with the "
                    "number of attention launches that raised the fixed-reference overflow flag and took the rescaling fallback (the seeded weights give "
                    "near-uniform softmax rows: logit spread about 8)")
    ap.add_argument("--cli_files", type=int, default=96, metavar="K", help="after the timed steps (N = 1, headline workload only): write K synthetic 512 x 512 PNGs and "
                    "the full-size artefacts, run `inference.py --sr_scale 4` on them as a fresh child process and report its files/s beside `value` (0 = skip)")
    ap.add_argument("--fp8", action="store_true", help="BASELINE configs[4]: fp8 (e4m3) MFMA operands in the parts ir_fp8_features() reports (the JSON line names them)")
    args = ap.parse_args()

    from instarevive_amd import parallel
    rank, world, local = parallel.env_rank_world()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` starts its own N ranks: fresh child processes (one per GPU, torch.distributed.run, rendezvous on
        # 127.0.0.1), spawned BEFORE this process touches the GPU; rank 0's JSON line and the exit code are relayed.
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    # clock / power trace of this rank's card over the whole run: a child that only reads sysfs, started BEFORE this process's first GPU call
    sampler, sampler_file = None, None
    if rank == 0 and not os.environ.get("IR_NO_POWER_TRACE"):
        import subprocess
        import tempfile
        sampler_file = os.path.join(tempfile.gettempdir(), f"ir_power_{os.getpid()}.txt")
        try:
            sampler = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "power_sampler.py"), "--out", sampler_file, "--card", str(local)],
                                       stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except OSError:
            sampler = None
        if sampler is not None:
            import atexit

            def stop_sampler():
                try:
                    sampler.stdin.close()
                    sampler.wait(timeout=5)
                    if os.environ.get("IR_KEEP_POWER_TRACE") and os.path.exists(sampler_file):   # the raw trace, for profiles/
                        import shutil
                        shutil.copy(sampler_file, os.environ["IR_KEEP_POWER_TRACE"])
                    os.unlink(sampler_file)
                except (OSError, subprocess.TimeoutExpired):
                    pass
            atexit.register(stop_sampler)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU; the product path has no CPU fallback")
    # IR_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share devices, collectives
    # go through the host); the driver's runs use the default, one process per GPU over RCCL
    backend = os.environ.get("IR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    # IR_FORCE_COLLECTIVES=1 (tests): a ONE-rank group goes through the same RCCL calls as the multi-GPU runs - the in-step gather of device
    # tensors, the MAX / MIN all-reduces, the barriers - so that the branch the driver's 2- / 4- / 8-GPU runs take has executed on a one-GPU box
    if world > 1 or parallel.force_collectives():
        import torch.distributed as dist
        parallel.init_distributed(backend)   # one process per GPU; "nccl" is RCCL on ROCm

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    from instarevive_amd import _lib as L
    swin, vae, dit, sched, sds = build_models(device, log, args.control)
    ctx = dit.ctx
    y, mask = synthetic_prompt()
    y_dev, mask_dev = y.to(device), mask.to(device)
    dit.set_prompt(y_dev, mask_dev)
    peak_tflops = PEAK_FP8_TFLOPS if args.fp8 else PEAK_BF16_TFLOPS

    if args.net_hw:
        hh, ww = (int(v) for v in args.net_hw.lower().split("x"))
        net_in = synthetic_lq(args.batch, hh, ww, 1000 + rank)
    else:
        lq = synthetic_lq(args.batch, args.lq, args.lq, 1000 + rank)
        net_in = upscale_bicubic(lq, args.sr_scale) if args.sr_scale != 1 else lq    # inference.py:265-269 (host, outside the timed region)
    n, h, w = net_in.shape[:3]
    assert h % 64 == 0 and w % 64 == 0
    flags = ((L.FLAG_TILED | L.FLAG_FIX_WAVELET) if args.tiled else 0) | (L.FLAG_CONTROL_LQ if args.control else 0) | (L.FLAG_FP8 if args.fp8 else 0)
    if args.fp8:
        feats = ctx.lib.ir_fp8_features()   # what THIS build moves to fp8 operands: the workload string says exactly that
        # --fp8_parts narrows the operand set (ir_set_fp8_mask; tools/fp8_attribution.py: the attention products cost 0.1 dB against the oracle,
        # the e4m3 conv activations 5.5 dB): "attention" = the three attention parts only, "no_encoder_convs" = everything but the encoder's convs
        conv_bits = L.FP8_CONV_BITS
        masks = {"qualified": L.FP8_MASK_QUALIFIED, "all": L.FP8_MASK_ALL, "attention": L.FP8_MASK_ATTENTION, "no_encoder_convs": 0xffffffff & ~(0x1f << 4), "convs": conv_bits}
        if args.fp8_parts == "default":   # what inference.py --fp8 default does: the operand set these weights allow
            from instarevive_amd import fp8_select
            fmask = fp8_select.auto_mask(swin, vae, dit, y_dev, mask_dev, log=log, use_cache=False)
            if dist is not None:   # every rank runs the same set (the calibration is deterministic; this pins it)
                t = torch.tensor([fmask], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
                dist.broadcast(t, 0)
                fmask = int(t[0])
        else:
            fmask = masks[args.fp8_parts]
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, fmask), "ir_set_fp8_mask")
        lvl = lambda base, name: [f"{name} level {l}" for l in range(4) if fmask >> (base + l) & 1] + ([f"{name} mid block"] if fmask >> (base + 4) & 1 else [])
        conv_parts = lvl(4, "encoder") + lvl(12, "decoder")
        conv_words = ("the VAE ResnetBlock 3x3 convs" if fmask & conv_bits == conv_bits else
                      ("the VAE ResnetBlock 3x3 convs of " + ", ".join(conv_parts)) if conv_parts else None)
        fp8_words = ", fp8 MFMA operands (MX-scaled e4m3) in " + " and ".join(
            w for bit, w, on in ((1, conv_words, bool(fmask & conv_bits)), (2, "the DiT self-attention products", bool(fmask & 1)),
                                 (4, "the VAE mid-block attention products", bool(fmask & 6))) if (feats & bit) and on)
        fp8_words += {"default": f" [operand set {fmask:#x} chosen on the loaded weights by instarevive_amd/fp8_select.py (calibration image, 0.1 dB budget): parity_2048 holds it to >= 46.3 dB vs the fp32 oracle on these seeded weights; fp8_auto.stress_weights: what the same rule chooses, costs and keeps on heavy-tailed / peaky weights]",
                      "qualified": " [the set qualified on flat-softmax weights, not calibrated]",
                      "all": " [ALL parts: OUT OF TOLERANCE above a 25.8 dB reference (42.1 dB vs the oracle); opt-in]"}.get(args.fp8_parts, f" [--fp8_parts {args.fp8_parts}]")
        vae.enable_fp8(True)                                # packs + uploads the fp8 weight forms of the VAE resnet convs
        ctx.check(ctx.lib.ir_set_fp8(ctx.h, 0), "ir_set_fp8")  # ... the mode itself is switched per call by IR_FLAG_FP8
    if args.graph:
        flags |= L.FLAG_GRAPH
        args.no_profile = True
    tile_size, tile_stride = 512, 448
    if args.tiled:
        dit.ensure_pos(tile_size // 16, tile_size // 16)
    else:
        dit.ensure_pos(h // 16, w // 16)
    din = net_in.to(device)
    dout = torch.empty_like(din)
    ws = ctx.workspace(ctx.ws_bytes(L.STAGE_PIPELINE, n, h, w, flags, tile_size, tile_stride))
    log(f"workload {n}x{h}x{w} per GPU, workspace {ws.numel() / 2**30:.1f} GiB, flags {flags}")
    acp, sf = float(sched.alphas_cumprod[400]), float(vae.config.scaling_factor)

    gathered = [None]
    # the per-rank image counts are static: exchanged once here, so the timed step carries exactly one RCCL gather and no host sync
    plan = parallel.GatherPlan(dout, dst=0) if dist is not None else None

    def step():
        ctx.check(ctx.lib.ir_pipeline(ctx.h, ctx.stream(), L.ptr(din), L.ptr(dout), None, n, h, w, flags, tile_size, tile_stride, 400.0, acp, sf,
                                      L.ptr(ws), ws.numel()), "ir_pipeline")
        if plan is not None:  # BASELINE configs[3]: the finished uint8 images of every rank are gathered on rank 0 over xGMI, inside the step
            gathered[0] = plan.gather(dout)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # An event between two launches is a barrier packet: ~700 of them per image cost 1.4 % (133.2 against 131.4 ms, A/B on one box). The timed
    # loop therefore brackets the launches of the DOMINANT kernel only - the live duration the roofline object needs - and the per-kernel
    # table comes from a second, untimed pass of the same K steps with every launch bracketed (--profile_all: one pass, as before round 4).
    dominant = None
    if not args.no_profile and not args.profile_all:
        ctx.profile_begin()
        for _ in range(max(args.warmup, 1)):
            step()
        torch.cuda.synchronize()
        ctx.profile_end()
        k0 = ctx.profile_end_kernels()
        dominant = max(k0, key=lambda k: k0[k]["ms"])
        if dist is not None:   # every rank brackets the same kernel
            names = sorted(k0)
            t = torch.tensor([names.index(dominant)], device=device if backend == "nccl" else "cpu")
            dist.broadcast(t, 0)
            dominant = names[int(t[0])]
    else:
        for _ in range(args.warmup):
            step()
    barrier()
    if not args.no_profile:
        ctx.profile_begin(only=dominant)
    wall0 = time.time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    power = None
    if sampler is not None:   # what the card's clock and socket power were DURING the timed loop (hwmon samples every 20 ms)
        from tools.power_sampler import summarise
        time.sleep(0.05)
        try:
            pr = torch.cuda.get_device_properties(local)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except AttributeError:
            bdf = None
        power = summarise(sampler_file, wall0, wall0 + dt, bdf)
        if power is not None:
            log(f"timed loop: gfx clock {power.get('clock_mhz')} MHz (min {power.get('clock_mhz_min')}, max {power.get('clock_mhz_max')}), socket power {power.get('power_w')} W "
                f"(max {power.get('power_w_max')}), {power['samples']} samples")
    if args.no_profile:   # diagnostic forms (--graph / --no_profile): the same JSON line without the per-launch measurements
        if rank == 0:
            fm = flops_model_tiled(h, w, tile_size, tile_stride, copies=args.control) if args.tiled else flops_model(h, w, copies=args.control)
            ms = dt / args.steps * 1e3
            log(f"unprofiled: {ms:.2f} ms/step")
            src = f"{h}x{w} synthetic network input" if args.net_hw else f"{args.lq}x{args.lq} LQ, sr_scale {args.sr_scale:g} -> {h}x{w} network input"
            print(json.dumps({
                "metric": "512->2048 one-step SR images/sec", "value": round(world * n * args.steps / dt, 4), "unit": "images/sec", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "fp8" if args.fp8 else "bf16", "data": "synthetic",
                "config": {"workload": f"{src}, {('tiled 512/448 + wavelet, %d tiles' % fm['tiles']) if args.tiled else 'untiled'}, batch {n} per GPU, "
                                       "full SwinIR->VAE-enc->DiT(t=400)->VAE-dec path"
                                       + (", whole step replayed as ONE hipGraph (IR_FLAG_GRAPH) - the step is GPU-bound (kernel time = wall time in the profiled "
                                          "form), so the graph buys nothing over plain launches: 237.9 against 238.7 ms per padded 4K frame, profiles/r05_bench_4k_tiled*.log"
                                          if args.graph else ", plain launches, no per-launch events"),
                           "global_batch": n * world, "parallelism": f"dp{world}", "weights": "seeded random, full-size architectures"},
                "algorithmic_tflop_per_image": round(fm["total"] / 1e12, 2), "path_tflops": round(fm["total"] * n * world / (ms / 1e3) / 1e12, 1),
                "roofline": None, "cpu_baseline": None}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    prof = ctx.profile_end()
    kprof = ctx.profile_end_kernels()
    table_ms_per_step = None
    if dominant is not None:   # the per-kernel table: the same K steps again with every launch bracketed, outside the timed region
        dom_live = kprof[dominant]
        ctx.profile_begin()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        table_ms_per_step = (time.perf_counter() - t1) / args.steps * 1e3
        prof = ctx.profile_end()
        kprof = ctx.profile_end_kernels()
        kprof[dominant] = dict(dom_live, ms_table_pass=kprof[dominant]["ms"])   # the dominant kernel's row stays the live measurement
        log(f"timed loop {dt / args.steps * 1e3:.2f} ms/step with events around {dominant.split('/', 1)[1]} only; table pass (every launch bracketed) {table_ms_per_step:.2f} ms/step")
    if dist is not None:
        t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    ms_per_step = dt / args.steps * 1e3
    value = world * n * args.steps / dt

    # ---- outside the timed region: is what was timed correct? The last output of the timed loop against a second pass through the
    # older 4-wave kernels (an independent implementation of every large contraction; ir_set_plain_kernels)
    verify = None
    if not args.no_verify:
        fast = dout.clone()
        ref_flags = flags & ~L.FLAG_FP8   # fp8 run: the reference pass is the bf16 path (plain kernels), which also prices the fp8 error
        ctx.check(ctx.lib.ir_set_plain_kernels(ctx.h, 1), "ir_set_plain_kernels")
        try:
            ctx.check(ctx.lib.ir_pipeline(ctx.h, ctx.stream(), L.ptr(din), L.ptr(dout), None, n, h, w, ref_flags, tile_size, tile_stride, 400.0, acp, sf,
                                          L.ptr(ws), ws.numel()), "ir_pipeline")
            torch.cuda.synchronize()
        finally:
            ctx.check(ctx.lib.ir_set_plain_kernels(ctx.h, 0), "ir_set_plain_kernels")
        mse = float(((fast.double() - dout.double()) ** 2).mean())
        psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        std = float(fast.double().std())
        if args.fp8:   # e4m3 operands carry 3 mantissa bits
            # against the bf16 pass through the PLAIN kernels (itself 48.6 dB from the fast bf16 kernels): the guard-chosen default set measured
            # 46.8 dB at 2048 x 2048, every part 42.5 dB (profiles/r02_bench_fp8.log); the gates are those minus a margin. The criterion
            # itself (>= 46.3 dB against the fp32 ORACLE) is the parity_2048 field
            gate8 = 45.0 if args.fp8_parts in ("default", "qualified", "attention") else 38.0
            verify = dict(verified=bool(psnr >= gate8 and std > 1.0), psnr_fp8_vs_bf16_plain_kernels_db=round(psnr, 2), output_std=round(std, 2), gate_db=gate8)
        else:
            verify = dict(verified=bool(psnr >= 45.0 and std > 1.0), psnr_fast_vs_plain_kernels_db=round(psnr, 2), output_std=round(std, 2))
        if dist is not None:
            ok = torch.tensor([1.0 if verify["verified"] else 0.0], device=device if backend == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            verify["verified"] = bool(ok[0] > 0)
            if rank == 0 and gathered[0] is not None:
                verify["gathered_images"] = int(gathered[0].shape[0])
                verify["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else backend
                verify["world_size"] = dist.get_world_size()
        log(f"verify: fast vs plain kernels {psnr:.2f} dB, output std {std:.1f}")

    # ---- outside the timed region: the TIMED configuration (bf16, or the fp8 operand set) at the headline size against the fp32 oracle, through
    # the committed fixture tests/golden/headline_crops.npz (24 crops of 128 x 128 of ONE oracle pass at 2048 x 2048 on bench.py's seeded weights)
    parity_2048 = None
    crops_file = os.path.join(ROOT, "tests", "golden", "headline_crops.npz")
    if rank == 0 and (h, w, n) == (2048, 2048, 1) and not args.tiled and not args.control and os.path.exists(crops_file):
        from tests.golden.make_headline_crops import CROP, inputs_for
        zc = np.load(crops_file)
        cin = torch.from_numpy(inputs_for(2048))[None].to(device)
        cout = torch.empty_like(cin)
        ctx.check(ctx.lib.ir_pipeline(ctx.h, ctx.stream(), L.ptr(cin), L.ptr(cout), None, 1, h, w, flags, tile_size, tile_stride, 400.0, acp, sf, L.ptr(ws), ws.numel()), "ir_pipeline")
        torch.cuda.synchronize()
        got = cout[0].cpu().numpy()
        got = np.stack([got[yy:yy + CROP, xx:xx + CROP] for yy, xx in zc["pos_2048"]]).astype(np.float64)
        mse_c = float(((got - zc["crops_2048"].astype(np.float64)) ** 2).mean()) / 255.0 ** 2
        p_c = 10.0 * np.log10(1.0 / (mse_c + 1e-12))
        parity_2048 = dict(psnr_vs_oracle_db=round(p_c, 2), within_0p1_db_up_to_reference_psnr_db=round(p_c + 10.0 * np.log10(10 ** 0.01 - 1.0), 1),
                           meets_0p1_db_at_30_db_reference=bool(p_c >= 46.3),
                           note="the timed configuration on the input of tests/golden/headline_crops.npz (512 x 512 LQ, sr_scale 4 -> 2048 x 2048), 24 crops of "
                                "128 x 128 of its uint8 result against the fp32 oracle's; >= 46.3 dB keeps PSNR(., GT) within 0.1 dB of the reference's up to a 30 dB reference")
        log(f"parity at 2048 x 2048 (oracle crops): {p_c:.2f} dB -> within 0.1 dB up to a {parity_2048['within_0p1_db_up_to_reference_psnr_db']} dB reference")
        del cin, cout

    # ---- the drop-in boundary hands over HOST arrays (inference.py:91-93,157-166): the same workload through process() (pinned staging,
    # synchronous) and through process_stream() (what the CLI runs: upload / download of neighbouring batches overlapped with compute)
    host = None
    if world == 1 and not args.no_host_rate and not args.control and not args.fp8:
        from instarevive_amd.pipeline import process, process_stream
        imgs = list(net_in.numpy())
        kw = dict(preprocess_model=swin, vae=vae, y=y_dev, y_mask=mask_dev, noise_scheduler=sched)
        process(dit, imgs, 1, "wavelet", False, args.tiled, tile_size, tile_stride, **kw)      # staging buffers
        t1 = time.perf_counter()
        for _ in range(args.steps):
            process(dit, imgs, 1, "wavelet", False, args.tiled, tile_size, tile_stride, **kw)
        dt_sync = (time.perf_counter() - t1) / args.steps
        k = args.steps + 2
        for _ in process_stream(dit, (imgs for _ in range(2)), "wavelet", False, args.tiled, tile_size, tile_stride, **kw):   # staging buffers
            pass
        t1 = time.perf_counter()
        for _ in process_stream(dit, (imgs for _ in range(k)), "wavelet", False, args.tiled, tile_size, tile_stride, **kw):
            pass
        dt_stream = (time.perf_counter() - t1) / k
        host = dict(value_host=round(n / dt_stream, 4), value_host_sync=round(n / dt_sync, 4), ms_per_step_stream=round(dt_stream * 1e3, 2),
                    ms_per_step_sync=round(dt_sync * 1e3, 2),
                    note="uint8 HWC host arrays in, prediction + stage-1 image out; stream = process_stream(), sync = one process() per step")
        log(f"host-buffer rate: stream {dt_stream * 1e3:.2f} ms/step, sync {dt_sync * 1e3:.2f} ms/step (device-resident {ms_per_step:.2f})")

    # ---- peaky softmax rows: the same steps with every self-attention logit scaled; what the overflow fallback costs when it fires
    peaky = None
    if args.logit_gain and world == 1 and not args.control:
        from tests.support.stress_weights import stress_state_dicts
        peaky = []
        try:
            for gain in args.logit_gain:
                st = stress_state_dicts(sds, frac=0.0, gain=1.0, logit_gain=gain)
                vae.load_state_dict(st["vae"])
                dit.load_state_dict(st["dit"])
                if args.fp8:
                    vae.enable_fp8(True)
                    ctx.check(ctx.lib.ir_set_fp8(ctx.h, 0), "ir_set_fp8")
                dit.set_prompt(y_dev, mask_dev)
                dit.ensure_pos(tile_size // 16 if args.tiled else h // 16, tile_size // 16 if args.tiled else w // 16)
                ctx.check(ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), 1), "ir_attn_fallback_count")
                step()
                torch.cuda.synchronize()
                per_step = ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), 0)
                ctx.check(ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), -1), "ir_attn_fallback_count")
                step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                ms_g = (time.perf_counter() - t1) / args.steps * 1e3
                peaky.append(dict(logit_gain=gain, ms_per_step=round(ms_g, 2), flat_ms_per_step=round(ms_per_step, 2), attention_launches_per_step=30 * n if not args.tiled else None,
                                  fallback_launches_per_step=per_step, output_std=round(float(dout.double().std()), 2)))
                log(f"logit gain {gain:g}: {ms_g:.2f} ms/step (flat {ms_per_step:.2f}), {per_step} attention launches per step took the rescaling fallback")
        finally:
            vae.load_state_dict(sds["vae"])
            dit.load_state_dict(sds["dit"])
            if args.fp8:
                vae.enable_fp8(True)
                ctx.check(ctx.lib.ir_set_fp8(ctx.h, 0), "ir_set_fp8")
            dit.set_prompt(y_dev, mask_dev)
            dit.ensure_pos(tile_size // 16 if args.tiled else h // 16, tile_size // 16 if args.tiled else w // 16)

    # ---- cfg-5 on weights that look like released ones: the SAME selection rule on the stress weights (heavy-tailed channels, peaky
    # attention: tests/support/stress_weights.py with the gains of tests/golden/stress_512.npz) - the operand set it chooses there, what that set and the
    # bf16 path take per step at the timed size, and both against the fp32 oracle's 512 x 512 result of the fixture
    fp8_auto = None
    if args.fp8 and args.fp8_parts == "default" and world == 1 and not (args.control or args.tiled):
        from instarevive_amd import fp8_select
        from instarevive_amd.pipeline import process as _process
        from tests.support.stress_weights import stress_state_dicts
        zs = np.load(os.path.join(ROOT, "tests", "golden", "stress_512.npz"))
        gains = {"dit": [float(v) for v in zs["logit_gain_dit"]], "vae_encoder": float(zs["logit_gain_vae"][0]), "vae_decoder": float(zs["logit_gain_vae"][1])}
        st = stress_state_dicts(sds, float(zs["frac"]), float(zs["gain"]), gains)
        simg = synthetic_lq(1, 512, 512, int(zs["lq_seed"]))[0].numpy()
        psnr_u8 = lambda a, b: float(10 * np.log10(255.0 ** 2 / max(float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean()), 1e-12)))

        def timed(fl):
            def one():
                ctx.check(ctx.lib.ir_pipeline(ctx.h, ctx.stream(), L.ptr(din), L.ptr(dout), None, n, h, w, fl, tile_size, tile_stride, 400.0, acp, sf, L.ptr(ws), ws.numel()), "ir_pipeline")
            one()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / args.steps * 1e3
        try:
            vae.enable_fp8(False)
            vae.load_state_dict(st["vae"])
            dit.load_state_dict(st["dit"])
            dit.invalidate_prompt()
            smask = fp8_select.auto_mask(swin, vae, dit, y_dev, mask_dev, log=log, use_cache=False)
            kw5 = dict(preprocess_model=swin, vae=vae, y=y_dev, y_mask=mask_dev, noise_scheduler=sched)
            sbf = _process(dit, [simg], 1, "wavelet", False, False, 512, 448, **kw5)[0][0]
            vae.enable_fp8(True)
            ctx.check(ctx.lib.ir_set_fp8(ctx.h, 0), "ir_set_fp8")
            ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, smask), "ir_set_fp8_mask")
            s8 = _process(dit, [simg], 1, "wavelet", False, False, 512, 448, fp8=True, **kw5)[0][0] if smask else sbf
            dit.set_prompt(y_dev, mask_dev)
            dit.ensure_pos(h // 16, w // 16)
            ms_bf = timed(flags & ~L.FLAG_FP8)
            ms_s8 = timed(flags) if smask else ms_bf
            fp8_auto = dict(seeded_weights=dict(mask=f"{fmask:#x}", ms_per_step=round(ms_per_step, 2)),
                            stress_weights=dict(mask=f"{smask:#x}", ms_per_step=round(ms_s8, 2), bf16_ms_per_step=round(ms_bf, 2),
                                                psnr_vs_oracle_512_db=round(psnr_u8(s8, zs["pred"]), 2), bf16_psnr_vs_oracle_512_db=round(psnr_u8(sbf, zs["pred"]), 2),
                                                gate="chosen set >= bf16 - 1.3 dB against the fp32 oracle (tests/test_headline_gpu.py)"),
                            rule="instarevive_amd/fp8_select.py: a part keeps its qualified cost only while it deviates from the bf16 pass as it did when it was qualified; budget 5.28e-6")
            log(f"fp8 auto on the stress weights: operand set {smask:#x}, {ms_s8:.2f} ms/step (bf16 {ms_bf:.2f}), {fp8_auto['stress_weights']['psnr_vs_oracle_512_db']:.2f} dB vs the oracle at 512 x 512 "
                f"(bf16 {fp8_auto['stress_weights']['bf16_psnr_vs_oracle_512_db']:.2f})")
        finally:
            vae.enable_fp8(False)
            vae.load_state_dict(sds["vae"])
            dit.load_state_dict(sds["dit"])
            vae.enable_fp8(True)
            ctx.check(ctx.lib.ir_set_fp8(ctx.h, 0), "ir_set_fp8")
            ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, fmask), "ir_set_fp8_mask")
            dit.set_prompt(y_dev, mask_dev)
            dit.ensure_pos(h // 16, w // 16)

    # ---- the shipped command line on FILES: K PNGs in, K PNGs out, through inference.py as a child process with its
    # reader / writer threads; the rate is the child's own clock from the first read to the last closed PNG (model loading excluded)
    cli = None
    if world == 1 and args.cli_files > 0 and not (args.control or args.fp8 or args.tiled or args.net_hw) and (h, w, n) == (2048, 2048, 1):
        # (the leg loads a second copy of the models in a child process and writes K PNGs + the artefacts to $TMPDIR - about 25-30 s and
        # 3.5 GB of scratch disk, both released afterwards; the parent's workspace goes first, and a failure is said on stderr, not only inside the line)
        ws = None
        ctx._ws = None
        torch.cuda.empty_cache()
        cli = cli_files_leg(args.cli_files, sds, value, log)
        if cli.get("error"):
            print(f"[bench] CLI LEG FAILED: {cli['error']} {cli.get('tail', '')[-400:]}", file=sys.stderr, flush=True)

    if rank == 0:
        fm = flops_model_tiled(h, w, tile_size, tile_stride, copies=args.control) if args.tiled else flops_model(h, w, copies=args.control)
        total_ms = sum(v["ms"] for v in prof.values())
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
            if v["launches"]:
                log(f"  {k:13s} {v['ms'] / args.steps:9.2f} ms/step  {v['launches'] // args.steps:5d} launches/step")
        log(f"  kernels {total_ms / args.steps:.1f} ms/step of {(table_ms_per_step or ms_per_step):.1f} ms/step wall{' (table pass)' if table_ms_per_step else ''}; timed loop {ms_per_step:.2f} ms/step; whole path {fm['total'] * n / (ms_per_step / 1e3) / 1e12:.1f} TFLOP/s algorithmic")
        # ---- one row per kernel: algorithmic FLOPs (un-padded dims) or bytes of its launches / the summed HIP-event duration of its launches
        per_kernel = {}
        for name, v in sorted(kprof.items(), key=lambda kv: -kv[1]["ms"]):
            short = name.split("/", 1)[1]
            fp8_kernel = "fp8" in short
            row = dict(ms_per_step=round(v["ms"] / args.steps, 3), launches_per_step=v["launches"] // args.steps)
            pk = PEAK_FP8_TFLOPS if fp8_kernel else PEAK_BF16_TFLOPS
            t_mfma, t_hbm = v["flops"] / (pk * 1e12), v["bytes"] / (PEAK_HBM_GBS * 1e9)   # seconds at either peak: the larger one bounds the kernel
            if v["flops"] > 0 and t_mfma >= t_hbm:
                ach = v["flops"] / (v["ms"] / 1e3) / 1e12
                row.update(bound="mfma", tflop_per_step=round(v["flops"] / args.steps / 1e12, 4), achieved=round(ach, 1), peak=pk, unit="TFLOP/s", frac=round(ach / pk, 4))
            elif v["bytes"] > 0:
                ach = v["bytes"] / (v["ms"] / 1e3) / 1e9
                row.update(bound="hbm", gb_per_step=round(v["bytes"] / args.steps / 1e9, 3), achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4))
                if v["flops"] > 0:
                    row["tflop_per_step"] = round(v["flops"] / args.steps / 1e12, 4)
            if "conv_halo_s1_kernel" in short and not fp8_kernel and not os.environ.get("IR_NO_UP2X2") and row.get("bound") == "mfma":
                skipped = n * (fm["tiles"] * upconv_phase_saving(tile_size, tile_size) if args.tiled else upconv_phase_saving(h, w))
                ex = (v["flops"] / args.steps - skipped) / (v["ms"] / args.steps / 1e3) / 1e12
                row.update(executed_tflop_per_step=round((v["flops"] / args.steps - skipped) / 1e12, 4), executed_achieved=round(ex, 1), executed_frac=round(ex / pk, 4),
                           note="achieved/frac price the reference's algorithmic 9-tap FLOPs; the three Upsample convs run as four 2x2 phase convs (4/9 of their MACs), executed_* price what the MFMA pipe really did"
                                + ("" if os.environ.get("IR_NO_S1_NORM") else "; the 128-output-channel launches also carry the GroupNorm apply + SiLU of their input (about 4.5 ms of "
                                   "stand-alone passes at 2048 x 2048 moved into this row: +3.5 ms here, IR_NO_S1_NORM=1 gives the two-launch form and 0.62)"))
            if short == "conv_halo_kernel" and not os.environ.get("IR_NO_UP2X2") and row.get("bound") == "mfma":
                # SwinIR's three 64-channel upsampler convs (outputs at 1/16, 1/4 and 1/1 of the pixels) run in the same phase form
                skipped = n * 2 * 9 * 64 * 64 * (h * w // 16 + h * w // 4 + h * w) * 5 / 9
                ex = (v["flops"] / args.steps - skipped) / (v["ms"] / args.steps / 1e3) / 1e12
                row.update(executed_tflop_per_step=round((v["flops"] / args.steps - skipped) / 1e12, 4), executed_achieved=round(ex, 1), executed_frac=round(ex / pk, 4),
                           note="achieved/frac price the algorithmic 9-tap FLOPs; SwinIR's three upsampler convs run as four 2x2 phase convs")
            if row.get("bound") == "mfma":
                # every MFMA-bound row carries BOTH prices: achieved / frac on the reference's algorithmic FLOPs (the contract's definition) and
                # executed_* on what the matrix pipe ran (equal unless the kernel skips MACs the reference does: the phase-form Upsample convs).
                row.setdefault("executed_tflop_per_step", row["tflop_per_step"])
                row.setdefault("executed_achieved", row["achieved"])
                row.setdefault("executed_frac", row["frac"])
                if row["frac"] > 1.0:
                    # only possible when the launches of this row are (mostly) phase-form convs priced on 9-tap FLOPs (the --fp8 lines: the other convs
                    # moved to the fp8 kernel): the algorithmic figure is not a roofline fraction then - the row's frac is the executed one
                    row.update(algorithmic_frac=row["frac"], frac=row["executed_frac"], frac_basis="executed (the algorithmic 9-tap FLOPs of this row's phase-form "
                               "launches exceed what the MFMA pipe ran; algorithmic_frac keeps that figure)")
            elif "frac" in row:   # HBM-bound rows move exactly their algorithmic bytes: the two prices coincide
                row.setdefault("executed_achieved", row["achieved"])
                row.setdefault("executed_frac", row["frac"])
            per_kernel[short] = row
            log(f"    {short[:58]:58s} {row['ms_per_step']:8.2f} ms/step {row['launches_per_step']:4d} launches  "
                + (f"{row['achieved']:8.1f} {row['unit']} = {row['frac']:.3f} of {row['bound']} peak" if "frac" in row else "")
                + (f" (executed {row['executed_frac']:.3f})" if row.get("executed_frac", row.get("frac")) != row.get("frac") else ""))
        # the dominant KERNEL (not family) by GPU time carries the roofline line; every other kernel is in per_kernel
        dom_name = dominant if dominant is not None else max(kprof, key=lambda k: kprof[k]["ms"])
        d, dshort = kprof[dom_name], dom_name.split("/", 1)[1]
        drow = per_kernel[dshort]
        traffic, traffic_src = None, None
        if not args.tiled and (h, w, n) == (2048, 2048, 1) and not args.fp8 and not args.control:
            pmc, why = pmc_file_for_this_tree()
            traffic_src = why
            if pmc is not None:
                per = json.load(open(pmc)).get("per_kernel", {})
                keys = [k for k in per if k.split("<")[0] in dshort]   # every template instantiation the timing row aggregates
                launches = sum(per[k]["launches_per_step"] for k in keys)
                if keys and launches > 0:
                    traffic = sum(per[k]["read_bytes_per_step"] + per[k]["write_bytes_per_step"] for k in keys) / launches
                    traffic_src = (f"static: profiles/{os.path.basename(pmc)} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command on the SAME "
                                   f"kernel sources - csrc_sha16 matches -, not this run; {len(keys)} instantiation(s), {launches:.0f} launches per step)")
        roof = dict(bound=drow.get("bound", "mfma"), kernel=dshort, achieved=drow.get("achieved"), peak=drow.get("peak"), unit=drow.get("unit"),
                    frac=drow.get("frac"), traffic=traffic, traffic_source=traffic_src, launches_per_step=drow["launches_per_step"],
                    avg_launch_ms=round(d["ms"] / max(d["launches"], 1), 4), share_of_gpu_time=round(d["ms"] / total_ms, 3),
                    algorithmic_tflop_per_step=drow.get("tflop_per_step"),
                    algorithmic_bytes_per_launch=round(d["bytes"] / max(d["launches"], 1)))
        if roof["bound"] == "mfma" and not args.fp8 and roof.get("achieved"):
            roof["frac_of_power_limited_mfma"] = round(roof["achieved"] / POWER_LIMITED_BF16_TFLOPS, 4)
            roof["power_limited_mfma_note"] = ("achieved / 1700 TFLOP/s = what a bare bf16 MFMA loop on random operands sustains on this pool's chips at the power-limited clock "
                                                "(profiles/r06_mfma_power_probe.txt); `frac` stays on the guide's 2500 TFLOP/s peak")
        if dominant is not None:
            roof["timing"] = ("live: HIP events around this kernel's launches only, in the timed loop; per_kernel's other rows, per_class_ms and roofline_family come from a "
                              "second pass of the same steps with every launch bracketed")
            roof["table_pass_ms_per_step"] = round(table_ms_per_step, 2)
            roof["avg_launch_ms_table_pass"] = round(d["ms_table_pass"] / max(d["launches"], 1), 4)
        else:
            roof["timing"] = "live: HIP events around every launch of the timed loop"
        roof["per_kernel"] = per_kernel
        # a figure that keeps its meaning from round to round next to the dominant-kernel one: the conv + GEMM family (every kernel of the
        # conv3x3 and linear classes, as BENCH_r02's `roofline` was defined) - algorithmic FLOPs / summed event time
        fam = {k: v for k, v in kprof.items() if k.split("/", 1)[0] in ("conv3x3", "linear") and v["launches"]}
        fam_ms, fam_fl = sum(v["ms"] for v in fam.values()), sum(v["flops"] for v in fam.values())
        if fam_ms > 0:
            fam_ach = fam_fl / (fam_ms / 1e3) / 1e12
            roof_family = dict(bound="mfma", kernels=sorted(k.split("/", 1)[1] for k in fam), achieved=round(fam_ach, 1), peak=peak_tflops, unit="TFLOP/s",
                               frac=round(fam_ach / peak_tflops, 4), ms_per_step=round(fam_ms / args.steps, 2),
                               algorithmic_tflop_per_step=round(fam_fl / args.steps / 1e12, 3))
        else:
            roof_family = None
        roof["per_class_ms"] = {k: round(v["ms"] / args.steps, 2) for k, v in prof.items() if v["launches"]}
        path_ach = fm["total"] * n / (ms_per_step / 1e3) / 1e12
        roof["whole_path"] = dict(algorithmic_tflop_per_step=round(fm["total"] * n / 1e12, 2), achieved=round(path_ach, 1), peak=PEAK_BF16_TFLOPS,
                                  frac=round(path_ach / PEAK_BF16_TFLOPS, 4))
        cpu = None
        if world == 1 and not args.no_cpu_baseline and not args.control:  # the CPU baseline times the headline workload only
            def hip_512(img):   # the timed configuration's kernels (bf16, or the fp8 operand set) on the CPU leg's 512 x 512 sample
                from instarevive_amd.pipeline import process   # (--fp8: the fp8 weight forms were uploaded when the run was set up)
                return process(dit, [img], 1, "wavelet", False, False, 512, 448, preprocess_model=swin, vae=vae, y=y_dev, y_mask=mask_dev,
                               noise_scheduler=sched, fp8=bool(args.fp8))[0][0]
            cpu = cpu_baseline(sds, y, mask, h, w, log, hip_fn=hip_512, big_pass=not args.cpu_small)
        src = f"{h}x{w} synthetic network input" if args.net_hw else f"{args.lq}x{args.lq} LQ, sr_scale {args.sr_scale:g} -> {h}x{w} network input"
        line = {
            "metric": "512->2048 one-step SR images/sec", "value": round(value, 4), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp8" if args.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": f"{src}, {('tiled 512/448 + wavelet, %d tiles' % fm['tiles']) if args.tiled else 'untiled'}, batch {n} per GPU, "
                                   "full SwinIR->VAE-enc->DiT(t=400)->VAE-dec path"
                                   + (f" + ControlNet-Half ({args.control} copied blocks, c = LQ latent)" if args.control else "")
                                   + (fp8_words if args.fp8 else "")
                                   + (f", one {'RCCL' if backend == 'nccl' else backend} gather of the uint8 results on rank 0 per step" if world > 1 else ""),
                       "global_batch": n * world, "parallelism": f"dp{world}", "weights": "seeded random, full-size architectures"},
            "algorithmic_tflop_per_image": round(fm["total"] / 1e12, 2),
            "path_tflops": round(fm["total"] * n * world / (ms_per_step / 1e3) / 1e12, 1),
            "roofline": roof, "roofline_family": roof_family, "cpu_baseline": cpu}
        if verify is not None:
            line.update(verify)
        if host is not None:
            line.update(host)
        if parity_2048 is not None:
            line["parity_2048"] = parity_2048
        if peaky is not None:
            line["peaky_attention"] = peaky
        if fp8_auto is not None:
            line["fp8_auto"] = fp8_auto
        if cli is not None:
            line["cli"] = cli
        line["clock_mhz"], line["power_w"] = (power or {}).get("clock_mhz"), (power or {}).get("power_w")
        line["power_trace"] = power
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
