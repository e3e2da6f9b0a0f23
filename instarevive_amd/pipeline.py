"""process() and its helpers with the reference's names and argument meaning (test_scripts/inference.py:39-166,
scripts/DMD/transformer_train/generate.py:22-87), running on the HIP path.

When the four models are instarevive_amd objects sharing one context, process() issues ONE call through the C ABI
(ir_pipeline): uint8 HWC in, uint8 HWC out, everything in between stays on the GPU in NHWC bf16 / fp32 statistics.
The stage-by-stage form (the reference's literal sequence of Python calls) is kept for drop-in use and for tests.
"""
import ctypes as C
from typing import List, Tuple

import numpy as np
import torch

from . import _lib as L
from .models import AutoencoderKL, ControlTransformerHalf, DDPMScheduler, SwinIR, Transformer2DModel


def _sliding_windows(h: int, w: int, tile_size: int, tile_stride: int):
    hi_list = list(range(0, h - tile_size + 1, tile_stride))
    if (h - tile_size) % tile_stride != 0:
        hi_list.append(h - tile_size)
    wi_list = list(range(0, w - tile_size + 1, tile_stride))
    if (w - tile_size) % tile_stride != 0:
        wi_list.append(w - tile_size)
    return [(hi, hi + tile_size, wi, wi + tile_size) for hi in hi_list for wi in wi_list]


def eps_to_mu(scheduler, model_output, sample, timesteps):
    acp = scheduler.alphas_cumprod.to(device=sample.device, dtype=sample.dtype)
    a = acp[timesteps]
    while a.ndim < sample.ndim:
        a = a.unsqueeze(-1)
    return (sample - (1 - a) ** 0.5 * model_output) / a ** 0.5


def forward_model(model, latents, timestep, prompt_embeds, prompt_attention_masks=None, c=None):
    added = {"resolution": None, "aspect_ratio": None}
    timestep = timestep.expand(latents.shape[0])
    if c is None:
        noise_pred = model(latents, timestep=timestep, encoder_hidden_states=prompt_embeds, encoder_attention_mask=prompt_attention_masks,
                           added_cond_kwargs=added).sample
    else:  # ControlTransformerHalf returns the tensor itself (generate.py:74-82)
        noise_pred = model(latents, timestep=timestep, encoder_hidden_states=prompt_embeds, encoder_attention_mask=prompt_attention_masks,
                           added_cond_kwargs=added, c=c)
    if model.config.out_channels // 2 == latents.shape[1]:
        noise_pred = noise_pred.chunk(2, dim=1)[0]
    return noise_pred


def generate_sample_1step(model, scheduler, latents, maxt, prompt_embeds, prompt_attention_masks=None, c=None):
    if isinstance(model, Transformer2DModel) and c is None:  # fused epilogue: eps half + eps_to_mu inside the HIP path
        return model.step(latents, float(maxt), float(scheduler.alphas_cumprod[int(maxt)]), prompt_embeds, prompt_attention_masks)
    if isinstance(model, ControlTransformerHalf) and c is not None:
        return model.step(latents, float(maxt), float(scheduler.alphas_cumprod[int(maxt)]), prompt_embeds, prompt_attention_masks, c=c)
    t = torch.full((1,), maxt, device=latents.device).long()
    noise_pred = forward_model(model, latents=latents, timestep=t, prompt_embeds=prompt_embeds, prompt_attention_masks=prompt_attention_masks, c=c)
    return eps_to_mu(scheduler, noise_pred, latents, t)


def wavelet_reconstruction(content_feat, style_feat):
    return _color_fix(L.FLAG_FIX_WAVELET, content_feat, style_feat)


def adaptive_instance_normalization(content_feat, style_feat):
    return _color_fix(L.FLAG_FIX_ADAIN, content_feat, style_feat)


def _color_fix(kind, content, style):
    from .models import get_context
    ctx = get_context(content.device)
    content = content.to(torch.float32).contiguous()
    style = style.to(content.device, torch.float32).contiguous()
    n, ch, h, w = content.shape
    if ch != 3 or style.shape != content.shape:
        raise ValueError("colour fix expects two [B,3,H,W] tensors of equal shape")
    out = torch.empty_like(content)
    ws = ctx.workspace(ctx.ws_bytes(L.STAGE_COLORFIX, n, h, w))
    ctx.check(ctx.lib.ir_color_fix(ctx.h, ctx.stream(), kind, L.ptr(content), L.ptr(style), L.ptr(out), n, h, w, L.ptr(ws), ws.numel()),
              "ir_color_fix")
    return out


def _fused_ok(model, preprocess_model, vae, disable_preprocess_model):
    if not (isinstance(model, (Transformer2DModel, ControlTransformerHalf)) and isinstance(vae, AutoencoderKL)):
        return False
    if not disable_preprocess_model and not isinstance(preprocess_model, SwinIR):
        return False
    ctxs = {id(m.ctx) for m in (model, vae) if m.ctx is not None}
    if not disable_preprocess_model and preprocess_model.ctx is not None:
        ctxs.add(id(preprocess_model.ctx))
    return len(ctxs) == 1


@torch.no_grad()
def process(model, control_imgs: List[np.ndarray], strength: float, color_fix_type: str, disable_preprocess_model: bool, tiled: bool,
            tile_size: int, tile_stride: int, preprocess_model=None, vae=None, y=None, y_mask=None, noise_scheduler=None,
            fused: bool = True, graph: bool = False) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """test_scripts/inference.py:55-166. control_imgs: list of HWC uint8 RGB arrays of equal size (multiples of 64).
    Returns (preds, stage1_preds) as lists of HWC uint8 arrays.

    Extension (no reference counterpart: the reference's process() never passes c): when `model` is a ControlTransformerHalf, the
    one-step call becomes generate_sample_1step(..., c=<the scaled LQ latent the step starts from>), per tile under `tiled`.
    graph=True (fused form only): the launch sequence is recorded into a hipGraph per image size / flag set and replayed on later
    calls (staging buffers are kept per size so that the recorded addresses stay valid)."""
    noise_scheduler = noise_scheduler or DDPMScheduler()
    n = len(control_imgs)
    imgs = np.ascontiguousarray(np.stack(control_imgs))
    if imgs.dtype != np.uint8 or imgs.ndim != 4 or imgs.shape[-1] != 3:
        raise ValueError("control_imgs must be HWC uint8 RGB arrays")
    h, w = imgs.shape[1:3]
    device = model.device
    acp = float(noise_scheduler.alphas_cumprod[400])
    sf = float(vae.config.scaling_factor)
    if fused and _fused_ok(model, preprocess_model, vae, disable_preprocess_model):
        ctx = model.ctx
        with_c = isinstance(model, ControlTransformerHalf)
        model.set_prompt(y, y_mask)
        if tiled:
            model.ensure_pos(tile_size // 16, tile_size // 16)
        else:
            model.ensure_pos(h // 16, w // 16)
        flags = (L.FLAG_NO_PREPROCESS if disable_preprocess_model else 0) | (L.FLAG_TILED if tiled else 0)
        flags |= {"wavelet": L.FLAG_FIX_WAVELET, "adain": L.FLAG_FIX_ADAIN}.get(color_fix_type, 0) if tiled else 0
        flags |= L.FLAG_CONTROL_LQ if with_c else 0
        if graph:  # stable device addresses for the recorded graph: one set of staging buffers per call signature
            flags |= L.FLAG_GRAPH
            bufs = ctx.__dict__.setdefault("_graph_bufs", {})
            key = (n, h, w, flags, tile_size, tile_stride)
            if key not in bufs:
                bufs[key] = tuple(torch.empty((n, h, w, 3), dtype=torch.uint8, device=device) for _ in range(3))
            din, dout, dst1 = bufs[key]
            din.copy_(torch.from_numpy(imgs))
        else:
            din = torch.from_numpy(imgs).to(device)
            dout = torch.empty_like(din)
            dst1 = torch.empty_like(din)
        ws = ctx.workspace(ctx.ws_bytes(L.STAGE_PIPELINE, n, h, w, flags, tile_size, tile_stride))
        ctx.check(ctx.lib.ir_pipeline(ctx.h, ctx.stream(), L.ptr(din), L.ptr(dout), L.ptr(dst1), n, h, w, flags, tile_size, tile_stride, 400.0, acp,
                                      sf, L.ptr(ws), ws.numel()), "ir_pipeline")
        preds, stage1 = dout.cpu().numpy(), dst1.cpu().numpy()
        return [preds[i] for i in range(n)], [stage1[i] for i in range(n)]

    # ---- stage-by-stage form: the reference's literal call sequence on NCHW fp32 tensors
    control = torch.tensor(imgs / 255.0, dtype=torch.float32, device=device).clamp_(0, 1).permute(0, 3, 1, 2).contiguous()
    if not disable_preprocess_model:
        control = preprocess_model(control)
    height, width = control.shape[-2:]
    lh, lw = height // 8, width // 8
    c_latent = vae.encode(control * 2 - 1).latent_dist.mode().to(torch.float32)
    init_noise = c_latent * sf
    with_c = isinstance(model, ControlTransformerHalf)
    if not tiled:
        latents = generate_sample_1step(model, noise_scheduler, init_noise, 400, y, y_mask, c=init_noise if with_c else None)
        img_buffer = vae.decode(latents / sf).sample / 2 + 0.5
    else:
        wins = _sliding_windows(lh, lw, tile_size // 8, tile_stride // 8)
        count = torch.zeros((n, 4, lh, lw), device=device)
        noise_buffer = torch.zeros_like(init_noise)
        for hi, he, wi, we in wins:
            tile = init_noise[:, :, hi:he, wi:we].contiguous()
            noise_buffer[:, :, hi:he, wi:we] += generate_sample_1step(model, noise_scheduler, tile, 400, y, y_mask, c=tile if with_c else None)
            count[:, :, hi:he, wi:we] += 1
        noise_buffer.div_(count)
        img_buffer = torch.zeros_like(control)
        count = torch.zeros_like(control)
        for hi, he, wi, we in wins:
            tile = vae.decode((noise_buffer[:, :, hi:he, wi:we] / sf).contiguous()).sample / 2 + 0.5
            cond = control[:, :, hi * 8:he * 8, wi * 8:we * 8].contiguous()
            if color_fix_type == "adain":
                tile = adaptive_instance_normalization(tile, cond)
            elif color_fix_type == "wavelet":
                tile = wavelet_reconstruction(tile, cond)
            img_buffer[:, :, hi * 8:he * 8, wi * 8:we * 8] += tile
            count[:, :, hi * 8:he * 8, wi * 8:we * 8] += 1
        img_buffer.div_(count)
    x_samples = (img_buffer.clamp(0, 1).permute(0, 2, 3, 1) * 255).cpu().numpy().clip(0, 255).astype(np.uint8)
    control = (control.permute(0, 2, 3, 1) * 255).cpu().numpy().clip(0, 255).astype(np.uint8)
    return [x_samples[i] for i in range(n)], [control[i] for i in range(n)]
