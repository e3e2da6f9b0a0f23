"""Pin BASELINE.json configs[1] (and the 1024x1024 half-way size) AT ITS OWN SIZE against the fp32 oracle.

    python tests/golden/make_headline_crops.py [--sizes 1024 2048]

Runs oracle.glue.process ONCE per size in the build container (CPU, fp32, full-size architectures with bench.py's seeded random
weights and bench.py's synthetic LQ image + bicubic upscale - exactly the tensors `python bench.py` feeds the HIP path) and writes
tests/golden/headline_crops.npz:

    x0_<S>      the x^_0 latent the one DiT step returns (float16 copy of the fp32 oracle result; 4 x S/8 x S/8)
    crops_<S>   24 (2048) / 12 (1024) crops of 128 x 128 pixels of the oracle's final uint8 image, at seeded positions (corners first)
    stage1_<S>  the four corner crops of the oracle's stage-1 (SwinIR) uint8 image
    pos_<S>     the (y, x) origins of the crops
    sum_<S>     int64 sum of all bytes of the full uint8 result (a cheap whole-image checksum)
    secs_<S>    host seconds the oracle needed (8 threads of the build container)

The oracle itself is pinned by the reference-generated fixtures of make_golden.py (tests/test_oracle_golden.py); this file extends the
reach of that pin to the headline size, where the HIP path runs kernels (gemm_pp_kernel, 16384-token attention, 65536-token VAE attention)
that no smaller whole-path comparison selects. The oracle needs about 1 minute at 1024 and 6-10 minutes at 2048 (about 25 GB of host memory).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

CROP = 128
N_CROPS = {1024: 12, 2048: 24}   # crops of the final image kept per size (the first ones of a 32-long seeded sequence; the file stays < 3 MB)
N_STAGE1 = 4                      # stage-1 crops kept (the four corners)
LQ_SEED = {1024: 21, 2048: 22}   # bench.synthetic_lq seeds; the LQ edge is S / 4 (sr_scale 4)


def inputs_for(size):
    """The network-size uint8 image of `size`: bench.py's synthetic LQ (size / 4) upscaled bicubically by 4 (inference.py's sr_scale)."""
    import bench
    lq = bench.synthetic_lq(1, size // 4, size // 4, LQ_SEED[size])
    return bench.upscale_bicubic(lq, 4.0)[0].numpy()


def crop_positions(size):
    rng = np.random.Generator(np.random.PCG64(1000 + size))
    pos = rng.integers(0, size - CROP + 1, size=(32, 2))
    pos[0] = (0, 0)                       # corners: the padding paths of every conv
    pos[1] = (size - CROP, size - CROP)
    pos[2] = (0, size - CROP)
    pos[3] = (size - CROP, 0)
    return pos[:N_CROPS[size]].astype(np.int32)


def take(img, pos):
    return np.stack([img[y:y + CROP, x:x + CROP] for y, x in pos])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[1024, 2048])
    ap.add_argument("--out", default=os.path.join(HERE, "headline_crops.npz"))
    a = ap.parse_args()
    import bench
    from instarevive_amd import weights as W
    from oracle import dit as odit, glue as oglue, swinir as oswin, vae as ovae
    torch.set_num_threads(int(os.environ.get("IR_CPU_THREADS", os.cpu_count())))
    swin_cfg = dict(embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8, mlp_ratio=2)
    sds = dict(swin=bench.random_state_dict(W.swinir_shapes(swin_cfg), 1),
               vae=bench.random_state_dict(W.vae_shapes(dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2)), 2),
               dit=bench.random_state_dict(W.dit_shapes(dict(num_layers=28, num_attention_heads=16, attention_head_dim=72, caption_channels=4096)), 3))
    y, mask = bench.synthetic_prompt()
    out = dict(np.load(a.out)) if os.path.exists(a.out) else {}
    for s in a.sizes:
        img = inputs_for(s)
        t0 = time.time()
        preds, stage1, inter = oglue.process([img], lambda x: oswin.swinir_forward(sds["swin"], x), lambda x: ovae.vae_encode_mean(sds["vae"], x),
                                             lambda lat, t, yy, mm: odit.dit_forward(sds["dit"], lat, t, yy, mm), lambda z: ovae.vae_decode(sds["vae"], z),
                                             oglue.alphas_cumprod_diffusers(), y, mask, return_intermediates=True)
        dt = time.time() - t0
        pos = crop_positions(s)
        out[f"x0_{s}"] = inter["x0"][0].numpy().astype(np.float16)
        out[f"crops_{s}"] = take(preds[0], pos)
        out[f"stage1_{s}"] = take(stage1[0], pos[:N_STAGE1])
        out[f"pos_{s}"] = pos
        out[f"sum_{s}"] = np.int64(preds[0].astype(np.int64).sum())
        out[f"secs_{s}"] = np.float32(dt)
        print(f"{s}x{s}: oracle pass {dt:.1f} s, x0 rms {float(inter['x0'].pow(2).mean().sqrt()):.4f}, image std {preds[0].std():.2f}", flush=True)
        np.savez_compressed(a.out, **out)
    print("wrote", a.out, os.path.getsize(a.out), "bytes")


if __name__ == "__main__":
    main()
