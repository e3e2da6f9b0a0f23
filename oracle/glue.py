"""CPU restatement of the pipeline glue around the four networks (test infrastructure; see oracle/__init__.py).

Follows /root/reference/test_scripts/inference.py:39-53 (_sliding_windows), :55-166 (process),
scripts/DMD/transformer_train/generate.py:22-87 (generate_sample_1step / forward_model / eps_to_mu),
diffusion/model/gaussian_diffusion.py:99-116 (linear beta schedule == the DDPMScheduler the CLI loads, :36),
utils/image/align_color.py:44-119 (colour fix), utils/image/common.py:229-249 (auto_resize / pad),
utils/metrics.py:8-38 (PSNR definition).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def sliding_windows(h, w, tile_size, tile_stride):
    hi_list = list(range(0, h - tile_size + 1, tile_stride))
    if (h - tile_size) % tile_stride != 0:
        hi_list.append(h - tile_size)
    wi_list = list(range(0, w - tile_size + 1, tile_stride))
    if (w - tile_size) % tile_stride != 0:
        wi_list.append(w - tile_size)
    return [(hi, hi + tile_size, wi, wi + tile_size) for hi in hi_list for wi in wi_list]


def alphas_cumprod(num_train_timesteps=1000, beta_start=1e-4, beta_end=2e-2):
    """Linear schedule in float64 like gaussian_diffusion.py:107-116, returned as float32 (diffusers stores float32)."""
    betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float64)
    return torch.from_numpy(np.cumprod(1.0 - betas, axis=0)).to(torch.float32)


def alphas_cumprod_diffusers(num_train_timesteps=1000, beta_start=1e-4, beta_end=2e-2):
    """diffusers DDPMScheduler(beta_schedule='linear'): betas = torch.linspace(..., dtype=float32); cumprod in float32."""
    betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)


def eps_to_mu(acp, model_output, sample, timesteps):
    a = acp.to(sample.dtype)[timesteps]
    while a.ndim < sample.ndim:
        a = a.unsqueeze(-1)
    return (sample - (1 - a) ** 0.5 * model_output) / a ** 0.5


def generate_sample_1step(model_fn, acp, latents, maxt, prompt_embeds, prompt_mask):
    """model_fn(latents, timestep[B], prompt_embeds, prompt_mask) -> [B,8,h,w]; keeps the eps half (generate.py:84-85)."""
    t = torch.full((1,), maxt).long()
    out = model_fn(latents, t.expand(latents.shape[0]), prompt_embeds, prompt_mask)
    if out.shape[1] // 2 == latents.shape[1]:
        out = out.chunk(2, dim=1)[0]
    return eps_to_mu(acp, out, latents, t)


def wavelet_blur(image, radius):
    k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]], dtype=image.dtype)[None, None].repeat(3, 1, 1, 1)
    return F.conv2d(F.pad(image, (radius,) * 4, mode="replicate"), k, groups=3, dilation=radius)


def wavelet_decomposition(image, levels=5):
    high = torch.zeros_like(image)
    low = image
    for i in range(levels):
        low = wavelet_blur(image, 2 ** i)
        high = high + (image - low)
        image = low
    return high, low


def wavelet_reconstruction(content, style):
    return wavelet_decomposition(content)[0] + wavelet_decomposition(style)[1]


def adaptive_instance_normalization(content, style, eps=1e-5):
    def ms(f):
        b, c = f.shape[:2]
        return f.reshape(b, c, -1).mean(2).reshape(b, c, 1, 1), (f.reshape(b, c, -1).var(2) + eps).sqrt().reshape(b, c, 1, 1)
    sm, ss = ms(style)
    cm, cs = ms(content)
    return (content - cm) / cs * ss + sm


def auto_resize(img, size):
    from PIL import Image
    short = min(img.size)
    if short < size:
        r = size / short
        return img.resize(tuple(math.ceil(x * r) for x in img.size), Image.BICUBIC)
    return img.copy()


def pad(img, scale):
    h, w = img.shape[:2]
    ph = 0 if h % scale == 0 else math.ceil(h / scale) * scale - h
    pw = 0 if w % scale == 0 else math.ceil(w / scale) * scale - w
    return np.pad(img, ((0, ph), (0, pw), (0, 0)), mode="constant", constant_values=0)


def psnr(a, b):
    """utils/metrics.py:8-38 (crop_border 0, RGB): a, b in [0,1], shape [N,3,H,W]."""
    mse = ((a.to(torch.float64) - b.to(torch.float64)) ** 2).mean(dim=[1, 2, 3])
    return 10.0 * torch.log10(1.0 / (mse + 1e-8))


@torch.no_grad()
def process(control_imgs, preprocess_fn, encode_fn, dit_fn, decode_fn, acp, y, y_mask, scaling_factor=0.18215, color_fix_type="wavelet",
            disable_preprocess_model=False, tiled=False, tile_size=512, tile_stride=448, return_intermediates=False):
    """inference.py:55-166 with the four networks passed as callables on fp32 NCHW tensors.
    Returns (preds uint8 [N,H,W,3], stage1 uint8 [N,H,W,3])."""
    n = len(control_imgs)
    control = torch.tensor(np.stack(control_imgs) / 255.0, dtype=torch.float32).clamp_(0, 1).permute(0, 3, 1, 2).contiguous()
    if not disable_preprocess_model:
        control = preprocess_fn(control)
    height, width = control.shape[-2:]
    h, w = height // 8, width // 8
    c_latent = encode_fn(control * 2 - 1).to(torch.float32)
    init_noise = c_latent * scaling_factor
    inter = {"control": control, "init_noise": init_noise}
    if not tiled:
        latents = generate_sample_1step(dit_fn, acp, init_noise, 400, y, y_mask)
        inter["x0"] = latents
        img = decode_fn(latents / scaling_factor) / 2 + 0.5
    else:
        wins = sliding_windows(h, w, tile_size // 8, tile_stride // 8)
        count = torch.zeros((n, 4, h, w))
        nb = torch.zeros_like(init_noise)
        for hi, he, wi, we in wins:
            nb[:, :, hi:he, wi:we] += generate_sample_1step(dit_fn, acp, init_noise[:, :, hi:he, wi:we], 400, y, y_mask)
            count[:, :, hi:he, wi:we] += 1
        nb.div_(count)
        inter["x0"] = nb
        img = torch.zeros_like(control)
        count = torch.zeros_like(control)
        for hi, he, wi, we in wins:
            t = decode_fn(nb[:, :, hi:he, wi:we] / scaling_factor) / 2 + 0.5
            cond = control[:, :, hi * 8:he * 8, wi * 8:we * 8]
            if color_fix_type == "adain":
                t = adaptive_instance_normalization(t, cond)
            elif color_fix_type == "wavelet":
                t = wavelet_reconstruction(t, cond)
            img[:, :, hi * 8:he * 8, wi * 8:we * 8] += t
            count[:, :, hi * 8:he * 8, wi * 8:we * 8] += 1
        img.div_(count)
    inter["img"] = img
    preds = (img.clamp(0, 1).permute(0, 2, 3, 1) * 255).numpy().clip(0, 255).astype(np.uint8)
    stage1 = (control.permute(0, 2, 3, 1) * 255).numpy().clip(0, 255).astype(np.uint8)
    if return_intermediates:
        return preds, stage1, inter
    return preds, stage1
