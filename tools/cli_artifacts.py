"""Artefacts for a full-size run of the drop-in command line with SEEDED RANDOM weights (no checkpoint exists offline): every file
inference.py loads, in the format and key layout the reference's own files have:

    weights/InstaRevive_v1.ckpt        flat DiT state dict, diffusers Transformer2DModel keys       (test_scripts/inference.py:239-241)
    weights/general_swinir_v1.ckpt     {"state_dict": {"module.<key>": tensor}}                     (:243-246, utils/common.py load_state_dict)
    configs/swinir.yaml                target + params of the released general SwinIR               (configs/swinir.yaml)
    vae/{config.json, diffusion_pytorch_model.safetensors}                                          (:236, stabilityai/sd-vae-ft-ema layout)
    pixart/{transformer/config.json, scheduler/scheduler_config.json}                               (:238, PixArt-Alpha-DMD-XL-2-512x512 layout)
    prompt.pth                         {'caption_embeds': [1,300,4096], 'emb_mask': [1,300]}        (:254-256)

and a folder of synthetic LQ PNGs. Used by `bench.py --cli_files K` and tests/test_cli_gpu.py (throughput of the CLI as a child process)."""
import json
import os

import numpy as np
import torch

# the released general SwinIR's constructor arguments (the architecture bench.py times; the reference keeps them in configs/swinir.yaml)
SWINIR_PARAMS = dict(img_size=64, patch_size=1, in_chans=3, embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8, mlp_ratio=2, sf=8, img_range=1.0,
                     upsampler="nearest+conv", resi_connection="1conv", unshuffle=True, unshuffle_scale=8)


def write_full_artifacts(d, sds=None):
    """Write the six artefacts under folder d (created). sds: {'swin','vae','dit'} state dicts (bench.build_models' fifth result); seeded
    random ones of the full architectures are made when absent. Returns the inference.py arguments that point at them."""
    from safetensors.torch import save_file
    from instarevive_amd import weights as W
    if sds is None:
        import bench
        sds = dict(swin=bench.random_state_dict(W.swinir_shapes(dict(embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8, mlp_ratio=2)), 1),
                   vae=bench.random_state_dict(W.vae_shapes(dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2)), 2),
                   dit=bench.random_state_dict(W.dit_shapes(dict(num_layers=28, num_attention_heads=16, attention_head_dim=72, caption_channels=4096)), 3))
    d = str(d)
    for sub in ("weights", "configs", "vae", "pixart/transformer", "pixart/scheduler"):
        os.makedirs(os.path.join(d, sub), exist_ok=True)
    torch.save(sds["dit"], os.path.join(d, "weights", "InstaRevive_v1.ckpt"))
    torch.save({"state_dict": {"module." + k: v for k, v in sds["swin"].items()}}, os.path.join(d, "weights", "general_swinir_v1.ckpt"))
    with open(os.path.join(d, "configs", "swinir.yaml"), "w") as f:
        import yaml
        yaml.safe_dump({"target": "diffusion.model.swinir.SwinIR", "params": SWINIR_PARAMS}, f)
    with open(os.path.join(d, "vae", "config.json"), "w") as f:
        json.dump({"_class_name": "AutoencoderKL", "in_channels": 3, "out_channels": 3, "latent_channels": 4, "block_out_channels": [128, 256, 512, 512],
                   "layers_per_block": 2, "norm_num_groups": 32, "scaling_factor": 0.18215, "act_fn": "silu", "sample_size": 256}, f)
    save_file({k: v.contiguous() for k, v in sds["vae"].items()}, os.path.join(d, "vae", "diffusion_pytorch_model.safetensors"))
    with open(os.path.join(d, "pixart", "transformer", "config.json"), "w") as f:
        json.dump({"_class_name": "Transformer2DModel", "num_attention_heads": 16, "attention_head_dim": 72, "in_channels": 4, "out_channels": 8,
                   "num_layers": 28, "cross_attention_dim": 1152, "attention_bias": True, "sample_size": 64, "patch_size": 2,
                   "activation_fn": "gelu-approximate", "norm_type": "ada_norm_single", "norm_elementwise_affine": False, "norm_eps": 1e-6,
                   "caption_channels": 4096, "num_embeds_ada_norm": 1000}, f)
    with open(os.path.join(d, "pixart", "scheduler", "scheduler_config.json"), "w") as f:
        json.dump({"_class_name": "DDPMScheduler", "num_train_timesteps": 1000, "beta_start": 0.0001, "beta_end": 0.02, "beta_schedule": "linear"}, f)
    g = torch.Generator().manual_seed(1234)
    mask = torch.zeros(1, 300)
    mask[:, :25] = 1
    torch.save({"caption_embeds": torch.randn(1, 300, 4096, generator=g) * 0.1, "emb_mask": mask}, os.path.join(d, "prompt.pth"))
    return ["--ckpt", os.path.join(d, "weights", "InstaRevive_v1.ckpt"), "--swinir_ckpt", os.path.join(d, "weights", "general_swinir_v1.ckpt"),
            "--swinir_config", os.path.join(d, "configs", "swinir.yaml"), "--vae", os.path.join(d, "vae"), "--dit_config", os.path.join(d, "pixart"),
            "--prompt_embeds", os.path.join(d, "prompt.pth")]


def write_lq_pngs(folder, k, edge=512, seed=500):
    """k synthetic LQ images (uniform noise low-passed by a 3 x 3 box, the bench's image model) as PNG files f000.png ..."""
    from PIL import Image
    os.makedirs(str(folder), exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    for i in range(k):
        x = torch.rand(1, 3, edge, edge, generator=g)
        x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="replicate"), 3, stride=1)
        Image.fromarray((x[0].permute(1, 2, 0) * 255).to(torch.uint8).numpy()).save(os.path.join(str(folder), f"f{i:03d}.png"))


def parse_cli_rate(stdout):
    """The summary line inference.py prints per rank: '[rank R] wrote K files in T s = X files/s (W host threads); after the first result: ...'."""
    import re
    out = []
    for m in re.finditer(r"\[rank (\d+)\] wrote (\d+) files in ([0-9.]+) s = ([0-9.]+) files/s \((\d+) host threads\); after the first result: "
                         r"(\d+) files in ([0-9.]+) s = ([0-9.]+) files/s(?:; results left the GPU at ([0-9.]+) /s)?", stdout):
        out.append(dict(rank=int(m.group(1)), files=int(m.group(2)), seconds=float(m.group(3)), files_per_s=float(m.group(4)), workers=int(m.group(5)),
                        steady_files=int(m.group(6)), steady_seconds=float(m.group(7)), steady_files_per_s=float(m.group(8))))
        if m.group(9):
            out[-1]["result_rate"] = float(m.group(9))
    return out
