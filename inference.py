#!/usr/bin/env python3
"""Drop-in for the reference's SR command line (test_scripts/inference.py, documented in README.md:57,63):

    python inference.py --ckpt weights/InstaRevive_v1.ckpt --input DIR --output DIR [--sr_scale F] [--tiled
        --tile_size 512 --tile_stride 448] [--color_fix_type wavelet|adain|none] [--disable_preprocess_model] ...

Same flags, defaults, file naming (`<stem>_<i>.png` under the input's relative path) and artefacts:
  * DiT:      flat state dict in diffusers Transformer2DModel key layout  (--ckpt; the reference parses --ckpt but
              hard-codes ./weights/InstaRevive_v1.ckpt, inference.py:239-240 — here --ckpt is honoured)
  * SwinIR:   ./weights/general_swinir_v1.ckpt with ./configs/swinir.yaml     (override: --swinir_ckpt / --swinir_config)
  * VAE:      diffusers folder 'stabilityai/sd-vae-ft-ema'                     (override: --vae)
  * prompt:   {'caption_embeds': [1,300,4096], 'emb_mask': [1,300]} .pth       (override: --prompt_embeds)
  * scheduler: only alphas_cumprod[400] is consumed                            (override: --dit_config folder)
All compute runs on the MI355X through hand-written HIP kernels; `--device cpu|mps` is rejected (no CPU path).
With torchrun (WORLD_SIZE > 1) the file list is sharded over the ranks, one process per GPU (images are independent).
"""
import math
import os
from argparse import ArgumentParser, Namespace

import numpy as np
import torch
from PIL import Image

DEFAULT_PROMPT = ("./output/tmp/real-world image, realistic, high quality, photograph, film, professional, 4k, "
                  "highly detailed_300token.pth")


def parse_args() -> Namespace:
    parser = ArgumentParser()
    parser.add_argument("--ckpt", required=True, type=str, help="full checkpoint path", default="./weights/InstaRevive_v1.ckpt")
    parser.add_argument("--input", type=str, required=True)
    parser.add_argument("--sr_scale", type=float, default=1)
    parser.add_argument("--repeat_times", type=int, default=1)
    parser.add_argument("--disable_preprocess_model", action="store_true")
    # patch-based sampling
    parser.add_argument("--tiled", action="store_true")
    parser.add_argument("--tile_size", type=int, default=512)
    parser.add_argument("--tile_stride", type=int, default=448)
    # latent image guidance (accepted and inert, like the reference)
    parser.add_argument("--use_guidance", action="store_true")
    parser.add_argument("--g_scale", type=float, default=0.0)
    parser.add_argument("--g_t_start", type=int, default=1001)
    parser.add_argument("--g_t_stop", type=int, default=-1)
    parser.add_argument("--g_space", type=str, default="latent")
    parser.add_argument("--g_repeat", type=int, default=5)
    parser.add_argument("--color_fix_type", type=str, default="wavelet", choices=["wavelet", "adain", "none"])
    parser.add_argument("--output", type=str, required=True)
    parser.add_argument("--show_lq", action="store_true")
    parser.add_argument("--skip_if_exist", action="store_true")
    parser.add_argument("--seed", type=int, default=231)
    parser.add_argument("--device", type=str, default="cuda", choices=["cpu", "cuda", "mps"])
    parser.add_argument("--use_prompt", action="store_true")
    parser.add_argument("--use_center_crop", action="store_true")
    # locations the reference hard-codes
    parser.add_argument("--swinir_ckpt", type=str, default="./weights/general_swinir_v1.ckpt")
    parser.add_argument("--swinir_config", type=str, default="./configs/swinir.yaml")
    parser.add_argument("--vae", type=str, default="stabilityai/sd-vae-ft-ema")
    parser.add_argument("--dit_config", type=str, default="PixArt-alpha/PixArt-Alpha-DMD-XL-2-512x512")
    parser.add_argument("--prompt_embeds", type=str, default=DEFAULT_PROMPT)
    return parser.parse_args()


def check_device(device: str) -> str:
    if device != "cuda" or not torch.cuda.is_available():
        raise SystemExit(f"device '{device}' requested / no GPU visible: this build runs on MI355X (ROCm) only and has no CPU or MPS path")
    print(f"using device {device}")
    return device


def main() -> None:
    from instarevive_amd.models import AutoencoderKL, DDPMScheduler, Transformer2DModel
    from instarevive_amd.pipeline import process
    from instarevive_amd.utils import (auto_resize, center_crop_arr, get_file_name_parts, instantiate_from_config, list_image_files,
                                       load_state_dict, load_yaml, pad)
    args = parse_args()
    torch.manual_seed(args.seed)  # the path is deterministic; kept for surface compatibility (pl.seed_everything)
    args.device = check_device(args.device)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    noise_scheduler = DDPMScheduler.from_pretrained(args.dit_config, subfolder="scheduler")
    vae = AutoencoderKL.from_pretrained(args.vae).to(torch.float32).to(device)
    model = Transformer2DModel.from_pretrained(args.dit_config, subfolder="transformer")
    model.load_state_dict(torch.load(args.ckpt, map_location="cpu"))
    preprocess_model = instantiate_from_config(load_yaml(args.swinir_config))
    load_state_dict(preprocess_model, torch.load(args.swinir_ckpt, map_location="cpu"), strict=True)
    model.to(device)
    preprocess_model.to(device)

    assert os.path.isdir(args.input)
    y_null_all = torch.load(args.prompt_embeds, map_location="cpu")
    y = y_null_all["caption_embeds"].to(device).to(torch.float32).reshape(1, -1, y_null_all["caption_embeds"].shape[-1])
    y_mask = y_null_all["emb_mask"].to(device).to(torch.float32).reshape(1, 1, -1)  # [1,1,L]: additive bias semantics (inference.py:274-277)

    files = sorted(list_image_files(args.input, follow_links=True))[rank::world]
    for file_path in files:
        lq = Image.open(file_path).convert("RGB")
        if args.sr_scale != 1:
            lq = lq.resize(tuple(math.ceil(x * args.sr_scale) for x in lq.size), Image.BICUBIC)
        if not args.tiled:
            if args.use_center_crop:
                lq_resized = center_crop_arr(lq, 512)
                x = np.array(lq_resized)
            else:
                lq_resized = auto_resize(lq, 512)
                x = pad(np.array(lq_resized), scale=64)
        else:
            lq_resized = auto_resize(lq, args.tile_size)
            x = pad(np.array(lq_resized), scale=64)
        for i in range(args.repeat_times):
            save_path = os.path.join(args.output, os.path.relpath(file_path, args.input))
            parent_path, stem, _ = get_file_name_parts(save_path)
            save_path = os.path.join(parent_path, f"{stem}_{i}.png")
            os.makedirs(parent_path, exist_ok=True)
            preds, stage1_preds = process(model, [x], strength=1, color_fix_type=args.color_fix_type,
                                          disable_preprocess_model=args.disable_preprocess_model, tiled=args.tiled, tile_size=args.tile_size,
                                          tile_stride=args.tile_stride, vae=vae, preprocess_model=preprocess_model, y=y, y_mask=y_mask,
                                          noise_scheduler=noise_scheduler)
            pred, stage1_pred = preds[0], stage1_preds[0]
            if not args.use_center_crop:
                height, width = (lq_resized.height, lq_resized.width)
                pred = pred[:height, :width, :]
                stage1_pred = stage1_pred[:height, :width, :]
            if args.show_lq:
                if not args.use_center_crop:
                    pred = np.array(Image.fromarray(pred).resize(lq.size, Image.LANCZOS))
                    stage1_pred = np.array(Image.fromarray(stage1_pred).resize(lq.size, Image.LANCZOS))
                    lq_arr = np.array(lq)
                else:
                    lq_arr = x
                images = [lq_arr, pred] if args.disable_preprocess_model else [lq_arr, stage1_pred, pred]
                Image.fromarray(np.concatenate(images, axis=1)).save(save_path)
            else:
                if not args.use_center_crop:
                    Image.fromarray(pred).resize(lq.size, Image.LANCZOS).save(save_path)
                else:
                    Image.fromarray(pred).save(save_path)
            print(f"save to {save_path}")


if __name__ == "__main__":
    main()
