#!/usr/bin/env python3
"""Batched evaluation harness of the one-step path — the role of the reference's test_scripts/test_dmd_general.py:112-192 (and
test_dmd.py:111-185 for the face weights): a folder of low-quality images goes through the network in batches of B and two folders come
out, the restored images and the stage-1 ("condition") images, each under its input's file name (`.jpg` -> `.png`, like save_batch,
test_dmd_general.py:37-51).

    python eval_batch.py --ckpt weights/InstaRevive_v1.ckpt --input DIR --output OUT_DIR --cond_output COND_DIR [--batch_size 4]
                         [--image_size 512] [--swinir_ckpt weights/face_swinir_v1.ckpt] [--prompt_embeds face_prompt.pth] ...

What the reference's harness builds with its dataset classes (synthetic degradation of ground-truth images, which SURVEY.md section 2.2
puts out of scope) is replaced by the file list of an EXISTING low-quality folder; every image is centre-cropped to image_size
(center_crop_arr, utils/image/common.py:12-36 — the reference's face / general evaluation feeds 512 x 512 crops), so a batch is
uniform and B images share every kernel launch. The per-batch arithmetic is the reference loop's: SwinIR -> VAE encode (mode) x scaling
factor -> generate_sample_1step at t = 400 -> VAE decode / 2 + 0.5 (:156-186), i.e. process() without tiling.
Model / prompt / scheduler artefacts are the ones of inference.py; the face variant differs only in --swinir_ckpt and --prompt_embeds.
With torchrun the file list is sharded over the ranks.
"""
import os
from argparse import ArgumentParser

import numpy as np
import torch
from PIL import Image

import inference as cli


def parse_args():
    ap = ArgumentParser()
    ap.add_argument("--ckpt", required=True)
    ap.add_argument("--input", required=True)
    ap.add_argument("--output", required=True)
    ap.add_argument("--cond_output", default=None, help="folder for the stage-1 (condition) images; default: <output>-cond")
    ap.add_argument("--batch_size", type=int, default=4)
    ap.add_argument("--image_size", type=int, default=512)
    ap.add_argument("--disable_preprocess_model", action="store_true")
    ap.add_argument("--swinir_ckpt", default="./weights/general_swinir_v1.ckpt")
    ap.add_argument("--swinir_config", default="./configs/swinir.yaml")
    ap.add_argument("--vae", default="stabilityai/sd-vae-ft-ema")
    ap.add_argument("--dit_config", default="PixArt-alpha/PixArt-Alpha-DMD-XL-2-512x512")
    ap.add_argument("--prompt_embeds", default=cli.DEFAULT_PROMPT)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--workers", type=int, default=-1, help="host threads for decoding / PNG encoding (inference.py --workers)")
    return ap.parse_args()


def out_name(folder, src_root, path):
    rel = os.path.relpath(path, src_root)
    stem, ext = os.path.splitext(rel)
    return os.path.join(folder, stem + ".png" if ext.lower() in (".jpg", ".jpeg") else rel)


def main():
    from instarevive_amd import parallel
    from instarevive_amd.pipeline import process_stream
    from instarevive_amd.utils import center_crop_arr, list_image_files
    args = parse_args()
    cli.check_device(args.device)
    rank, world, local = parallel.init_distributed()
    torch.cuda.set_device(local)
    m = cli.load_models(args, torch.device("cuda", local))
    cond_dir = args.cond_output or args.output.rstrip("/") + "-cond"
    # os.walk order is filesystem-dependent: every rank sorts its listing and takes rank 0's copy, so that ownership of a file follows from
    # ONE list (the same rule as inference.py)
    files = sorted(list_image_files(args.input, follow_links=True))
    if world > 1:
        files = parallel.agree_on_list(files)
    files = parallel.shard(files, rank, world)
    batches = [files[i:i + args.batch_size] for i in range(0, len(files), args.batch_size)]
    pools = cli.HostPools(cli.default_workers(int(os.environ.get("LOCAL_WORLD_SIZE", world))) if args.workers < 0 else args.workers)

    def feed():
        # decode + centre crop run ahead of the GPU on the reader threads, in file order
        crops = pools.read_ahead(lambda f: center_crop_arr(Image.open(f).convert("RGB"), args.image_size), files)
        for group in batches:
            yield [next(crops) for _ in group]

    def save(dst, img):
        os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
        Image.fromarray(np.ascontiguousarray(img)).save(dst)

    results = process_stream(m.model, feed(), "none", args.disable_preprocess_model, False, 512, 448, preprocess_model=m.preprocess_model, vae=m.vae,
                             y=m.y, y_mask=m.y_mask, noise_scheduler=m.noise_scheduler, return_stage1=True)
    for group, (preds, stage1) in zip(batches, results):
        for f, pred, cond in zip(group, preds, stage1):
            for folder, img in ((args.output, pred), (cond_dir, cond)):
                pools.write_behind(save, out_name(folder, args.input, f), img)
        print(f"[rank {rank}] queued {len(group)} images ({group[0]} ...)")
    pools.drain()
    print(f"[rank {rank}] saved {pools.written} files")


if __name__ == "__main__":
    main()
