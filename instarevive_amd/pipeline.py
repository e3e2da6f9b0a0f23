"""process() and its helpers with the reference's names and argument meaning (test_scripts/inference.py:39-166,
scripts/DMD/transformer_train/generate.py:22-87), running on the HIP path.

When the four models are instarevive_amd objects sharing one context, process() issues ONE call through the C ABI
(ir_pipeline): uint8 HWC in, uint8 HWC out, everything in between stays on the GPU in NHWC bf16 / fp32 statistics.
The stage-by-stage form (the reference's literal sequence of Python calls) is kept for drop-in use and for tests.
"""
import ctypes as C
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .models import AutoencoderKL, ControlTransformerHalf, DDPMScheduler, SwinIR, Transformer2DModel


class _Staging:
    """Host <-> device staging of one image batch shape: page-locked host buffers (so the 12.6 MB per 2048 x 2048 image cross PCIe
    by DMA at link rate instead of through a pageable bounce copy) and the matching device buffers, kept per shape on the context.
    `slots` independent sets let process_stream() upload batch i+1 and download batch i-1 while batch i computes."""

    def __init__(self, device, n, h, w, slots=1):
        shape = (n, h, w, 3)
        self.shape = shape
        self.h_in = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.h_out = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.h_st1 = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.d_in = [torch.empty(shape, dtype=torch.uint8, device=device) for _ in range(slots)]
        self.d_out = [torch.empty(shape, dtype=torch.uint8, device=device) for _ in range(slots)]
        self.d_st1 = [torch.empty(shape, dtype=torch.uint8, device=device) for _ in range(slots)]
        self.h2d_done = [None] * slots   # event behind the last asynchronous upload out of h_in[slot] (see upload())

    @staticmethod
    def get(ctx, n, h, w, slots=1, tag="sync"):
        pool = ctx.__dict__.setdefault("_staging", {})
        key = (tag, n, h, w, slots)
        if key not in pool:
            if len(pool) >= 8:  # bounded: drop the oldest shape
                pool.pop(next(iter(pool)))
            pool[key] = _Staging(ctx.device, n, h, w, slots)
        return pool[key]

    def fill(self, slot, control_imgs):
        """Copy the caller's HWC uint8 arrays into the pinned input buffer of `slot` (one pass, no intermediate np.stack)."""
        if self.h2d_done[slot] is not None:   # the previous upload out of this pinned buffer may still be queued behind earlier kernels
            self.h2d_done[slot].synchronize()
            self.h2d_done[slot] = None
        dst = self.h_in[slot].numpy()
        for i, im in enumerate(control_imgs):
            if im.dtype != np.uint8 or im.shape != self.shape[1:]:
                raise ValueError("control_imgs must be HWC uint8 RGB arrays of equal size")
            np.copyto(dst[i], im)

    def upload(self, slot, stream=None):
        """Asynchronous H2D copy of the pinned input buffer of `slot` on `stream` (default: the current one). The event recorded behind
        it is what the next fill() of the slot waits for: a caller that never synchronises with the device between two images (a
        rank that only contributes tiles under --shard_tiles) must not overwrite the pinned buffer while its copy is still queued."""
        self.d_in[slot].copy_(self.h_in[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream if stream is not None else torch.cuda.current_stream(self.d_in[slot].device))
        self.h2d_done[slot] = ev
        return ev


def _check_images(control_imgs):
    if len(control_imgs) == 0:
        raise ValueError("control_imgs is empty")
    first = np.asarray(control_imgs[0])
    if first.dtype != np.uint8 or first.ndim != 3 or first.shape[-1] != 3:
        raise ValueError("control_imgs must be HWC uint8 RGB arrays")
    return len(control_imgs), first.shape[0], first.shape[1]


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
    if model.config.sample_size == 128:   # generate.py:56-62: micro-conditioning on the latent's height / width
        bsz, _, height, width = latents.shape
        added = {"resolution": torch.tensor([height, width]).repeat(bsz, 1).to(prompt_embeds.dtype),
                 "aspect_ratio": torch.tensor([float(height / width)]).repeat(bsz, 1).to(prompt_embeds.dtype)}
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


def _pipeline_flags(model, color_fix_type, disable_preprocess_model, tiled):
    flags = (L.FLAG_NO_PREPROCESS if disable_preprocess_model else 0) | (L.FLAG_TILED if tiled else 0)
    if tiled:
        flags |= {"wavelet": L.FLAG_FIX_WAVELET, "adain": L.FLAG_FIX_ADAIN}.get(color_fix_type, 0)
    if isinstance(model, ControlTransformerHalf):
        flags |= L.FLAG_CONTROL_LQ
    return flags


def _prepare_fused(model, y, y_mask, h, w, tiled, tile_size, others=()):
    for m in others:  # the context must hold THESE models' weights (another instance of the family may have been loaded since)
        if m is not None:
            m._ready()
    model._ready()    # a ControlTransformerHalf re-binds its control branch here (set_prompt below resolves to the base model only)
    model.set_prompt(y, y_mask)
    if tiled:
        model.ensure_pos(tile_size // 16, tile_size // 16)
    else:
        model.ensure_pos(h // 16, w // 16)


def _launch_pipeline(ctx, st, slot, n, h, w, flags, tile_size, tile_stride, acp, sf, want_stage1=True):
    ws = ctx.workspace(ctx.ws_bytes(L.STAGE_PIPELINE, n, h, w, flags, tile_size, tile_stride))
    ctx.check(ctx.lib.ir_pipeline(ctx.h, ctx.stream(), L.ptr(st.d_in[slot]), L.ptr(st.d_out[slot]), L.ptr(st.d_st1[slot]) if want_stage1 else None,
                                  n, h, w, flags, tile_size, tile_stride, 400.0, acp, sf, L.ptr(ws), ws.numel()), "ir_pipeline")


@torch.no_grad()
def process(model, control_imgs: List[np.ndarray], strength: float, color_fix_type: str, disable_preprocess_model: bool, tiled: bool,
            tile_size: int, tile_stride: int, preprocess_model=None, vae=None, y=None, y_mask=None, noise_scheduler=None,
            fused: bool = True, graph: bool = False, return_stage1: bool = True, fp8: bool = False) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """test_scripts/inference.py:55-166. control_imgs: list of HWC uint8 RGB arrays of equal size (multiples of 64).
    Returns (preds, stage1_preds) as lists of HWC uint8 arrays (stage1_preds is empty with return_stage1=False, which skips its
    conversion and download).

    Extension (no reference counterpart: the reference's process() never passes c): when `model` is a ControlTransformerHalf, the
    one-step call becomes generate_sample_1step(..., c=<the scaled LQ latent the step starts from>), per tile under `tiled`.
    graph=True (fused form only): the launch sequence is recorded into a hipGraph per image size / flag set and replayed on later
    calls. Images travel through page-locked staging buffers kept per batch shape (which also gives a recorded graph stable
    device addresses). fp8=True (fused form, BASELINE.json configs[4]): fp8 MFMA operands in the VAE resnet convolutions
    (vae.enable_fp8() must have uploaded the fp8 weight forms)."""
    noise_scheduler = noise_scheduler or DDPMScheduler()
    n, h, w = _check_images(control_imgs)
    device = model.device
    acp = float(noise_scheduler.alphas_cumprod[400])
    sf = float(vae.config.scaling_factor)
    if fp8 and not (fused and _fused_ok(model, preprocess_model, vae, disable_preprocess_model)):
        raise ValueError("process(fp8=True) needs the fused form (instarevive_amd models sharing one context)")
    if fused and _fused_ok(model, preprocess_model, vae, disable_preprocess_model):
        if fp8 and not vae.__dict__.get("_fp8_uploaded"):
            raise RuntimeError("process(fp8=True): call vae.enable_fp8() first - without the fp8 weight forms every layer would silently run in bf16")
        ctx = model.ctx
        _prepare_fused(model, y, y_mask, h, w, tiled, tile_size, (vae, None if disable_preprocess_model else preprocess_model))
        flags = _pipeline_flags(model, color_fix_type, disable_preprocess_model, tiled) | (L.FLAG_GRAPH if graph else 0) | (L.FLAG_FP8 if fp8 else 0)
        st = _Staging.get(ctx, n, h, w)
        st.fill(0, control_imgs)
        st.upload(0)
        _launch_pipeline(ctx, st, 0, n, h, w, flags, tile_size, tile_stride, acp, sf, return_stage1)
        st.h_out[0].copy_(st.d_out[0], non_blocking=True)
        if return_stage1:
            st.h_st1[0].copy_(st.d_st1[0], non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
        preds = st.h_out[0].clone().numpy()   # the caller owns the result; the pinned buffer is reused by the next call
        stage1 = st.h_st1[0].clone().numpy() if return_stage1 else None
        return [preds[i] for i in range(n)], ([stage1[i] for i in range(n)] if return_stage1 else [])

    imgs = np.ascontiguousarray(np.stack(control_imgs))
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


@torch.no_grad()
def process_stream(model, batches: Iterable[Sequence[np.ndarray]], color_fix_type: str, disable_preprocess_model: bool, tiled: bool,
                   tile_size: int, tile_stride: int, preprocess_model=None, vae=None, y=None, y_mask=None, noise_scheduler=None,
                   return_stage1: bool = True, graph: bool = False, fp8: bool = False) -> Iterator[Tuple[List[np.ndarray], List[np.ndarray]]]:
    """process() over a sequence of image batches with the transfers hidden: while batch i computes on the current stream, batch
    i+1 is uploaded and batch i-1 downloaded on a copy stream (two staging slots per batch shape). Yields process()'s result for
    every batch, in order. Needs the fused form (all models instarevive_amd objects on one context). fp8 as in process() (cfg-5:
    vae.enable_fp8() first; the operand set is the context's ir_set_fp8_mask, by default the tolerance-chosen one)."""
    noise_scheduler = noise_scheduler or DDPMScheduler()
    if not _fused_ok(model, preprocess_model, vae, disable_preprocess_model):
        raise TypeError("process_stream needs instarevive_amd models sharing one context")
    ctx, device = model.ctx, model.device
    acp, sf = float(noise_scheduler.alphas_cumprod[400]), float(vae.config.scaling_factor)
    if fp8 and not vae.__dict__.get("_fp8_uploaded"):
        raise RuntimeError("process_stream(fp8=True): call vae.enable_fp8() first - without the fp8 weight forms every layer would silently run in bf16")
    base_flags = _pipeline_flags(model, color_fix_type, disable_preprocess_model, tiled) | (L.FLAG_GRAPH if graph else 0) | (L.FLAG_FP8 if fp8 else 0)
    main, copy = torch.cuda.current_stream(device), ctx.__dict__.setdefault("_copy_stream", torch.cuda.Stream(device))
    it = iter(batches)

    def upload(imgs, slot):
        n, h, w = _check_images(imgs)
        st = _Staging.get(ctx, n, h, w, slots=2, tag="stream")
        st.fill(slot, imgs)
        with torch.cuda.stream(copy):
            ev = st.upload(slot, copy)
        return st, slot, (n, h, w), ev

    def download(job):
        st, slot, (n, h, w), done = job
        done.synchronize()
        preds = st.h_out[slot].clone().numpy()
        stage1 = st.h_st1[slot].clone().numpy() if return_stage1 else None
        return [preds[i] for i in range(n)], ([stage1[i] for i in range(n)] if return_stage1 else [])

    slot, pending = 0, None
    nxt = next(it, None)
    up = upload(nxt, slot) if nxt is not None else None
    while up is not None:
        st, cur, (n, h, w), ready = up
        _prepare_fused(model, y, y_mask, h, w, tiled, tile_size, (vae, None if disable_preprocess_model else preprocess_model))
        main.wait_event(ready)
        _launch_pipeline(ctx, st, cur, n, h, w, base_flags, tile_size, tile_stride, acp, sf, return_stage1)
        computed = torch.cuda.Event()
        computed.record(main)
        # while this batch computes: fetch the previous result, stage the next input into the other slot. The other slot's device
        # buffers were last read by the previous batch's download, which download() has waited for by then.
        if pending is not None:
            yield download(pending)
        nxt = next(it, None)
        up = upload(nxt, cur ^ 1) if nxt is not None else None
        with torch.cuda.stream(copy):
            copy.wait_event(computed)
            st.h_out[cur].copy_(st.d_out[cur], non_blocking=True)
            if return_stage1:
                st.h_st1[cur].copy_(st.d_st1[cur], non_blocking=True)
            done = torch.cuda.Event()
            done.record(copy)
        pending = (st, cur, (n, h, w), done)
    if pending is not None:
        yield download(pending)


class HipTileEngine:
    """The five phases of tiled sampling (ir_tiled_* of the C ABI) on one GPU, in the form parallel.sharded_tiled_process() drives:
    every rank encodes, each rank runs the DiT / the decoder on ITS tiles, the per-tile results are exchanged between the phases."""

    def __init__(self, model, vae, preprocess_model, y, y_mask, color_fix_type="wavelet", disable_preprocess_model=False, tile_size=512,
                 tile_stride=448, noise_scheduler=None):
        if not _fused_ok(model, preprocess_model, vae, disable_preprocess_model):
            raise TypeError("HipTileEngine needs instarevive_amd models sharing one context")
        self.model, self.ctx, self.device = model, model.ctx, model.device
        self.others = (vae, None if disable_preprocess_model else preprocess_model)
        self.y, self.y_mask = y, y_mask
        self.tile_size, self.tile_stride = tile_size, tile_stride
        self.flags = _pipeline_flags(model, color_fix_type, disable_preprocess_model, True)
        sch = noise_scheduler or DDPMScheduler()
        self.acp, self.sf = float(sch.alphas_cumprod[400]), float(vae.config.scaling_factor)

    def _ws(self):
        n, h, w = self.shape
        return self.ctx.workspace(self.ctx.ws_bytes(L.STAGE_PIPELINE, n, h, w, self.flags, self.tile_size, self.tile_stride))

    def count(self, h, w):
        k = self.ctx.lib.ir_tiled_count(h, w, self.tile_size, self.tile_stride)
        if k <= 0:
            raise ValueError(f"bad tile geometry for a {h}x{w} image: tile {self.tile_size}, stride {self.tile_stride}")
        return k

    def encode(self, control_imgs):
        n, h, w = _check_images(control_imgs)
        self.shape = (n, h, w)
        _prepare_fused(self.model, self.y, self.y_mask, h, w, True, self.tile_size, self.others)
        st = _Staging.get(self.ctx, n, h, w)
        st.fill(0, control_imgs)
        st.upload(0)
        control = torch.empty((n, 3, h, w), dtype=torch.float32, device=self.device)
        init = torch.empty((n, 4, h // 8, w // 8), dtype=torch.float32, device=self.device)
        ws = self._ws()
        c = self.ctx
        c.check(c.lib.ir_tiled_encode(c.h, c.stream(), L.ptr(st.d_in[0]), L.ptr(st.d_st1[0]), L.ptr(control), L.ptr(init), n, h, w, self.flags,
                                      self.sf, L.ptr(ws), ws.numel()), "ir_tiled_encode")
        self._stage1 = st.d_st1[0]
        return control, init

    def can_shard_encode(self, control_imgs):
        """The encoder's mid-block attention can be split by query rows: one image, 512-channel mid block, h * w / 64 a multiple of 128."""
        n, h, w = _check_images(control_imgs)
        vae = self.others[0]
        return n == 1 and ((h // 8) * (w // 8)) % 128 == 0 and vae.config.block_out_channels[-1] == 512

    def encode_overflow(self):
        """1 when the fixed softmax reference of the last encode_part0 overflowed on THIS rank's rows (all rows of attn_o were then
        recomputed by the rescaling kernel), else 0. Synchronises the stream."""
        c = self.ctx
        v = c.lib.ir_tiled_encode_overflow(c.h, c.stream())
        if v < 0:
            c.check(v, "ir_tiled_encode_overflow")
        return int(v)

    def encode_part0(self, control_imgs, row0, row1, force_fallback=False):
        """ir_tiled_encode_part(part 0): everything up to the attention of query rows [row0, row1) of the encoder's mid block. Returns
        (control, attn_o, attn_res): this rank's rows of attn_o are filled; the exchange of the rows is the caller's (parallel.py)."""
        n, h, w = _check_images(control_imgs)
        self.shape = (n, h, w)
        _prepare_fused(self.model, self.y, self.y_mask, h, w, True, self.tile_size, self.others)
        st = _Staging.get(self.ctx, n, h, w)
        st.fill(0, control_imgs)
        st.upload(0)
        T = (h // 8) * (w // 8)
        control = torch.empty((n, 3, h, w), dtype=torch.float32, device=self.device)
        self._init = torch.empty((n, 4, h // 8, w // 8), dtype=torch.float32, device=self.device)
        attn_o = torch.empty((T, 512), dtype=torch.bfloat16, device=self.device)
        attn_res = torch.empty((T, 512), dtype=torch.bfloat16, device=self.device)
        ws, c = self._ws(), self.ctx
        c.check(c.lib.ir_tiled_encode_part(c.h, c.stream(), L.ptr(st.d_in[0]), L.ptr(st.d_st1[0]), L.ptr(control), L.ptr(self._init), n, h, w, self.flags,
                                           self.sf, 2 if force_fallback else 0, row0, row1, L.ptr(attn_o), L.ptr(attn_res), L.ptr(ws), ws.numel()),
                "ir_tiled_encode_part(0)")
        self._stage1 = st.d_st1[0]
        return control, attn_o, attn_res

    def encode_part1(self, control, attn_o, attn_res):
        """ir_tiled_encode_part(part 1): the rest of the encoder from all rows of attn_o. Returns init."""
        n, h, w = self.shape
        ws, c = self._ws(), self.ctx
        c.check(c.lib.ir_tiled_encode_part(c.h, c.stream(), None, None, L.ptr(control), L.ptr(self._init), n, h, w, self.flags, self.sf, 1, 0, 0,
                                           L.ptr(attn_o), L.ptr(attn_res), L.ptr(ws), ws.numel()), "ir_tiled_encode_part(1)")
        return self._init

    def stage1(self):
        n = self.shape[0]
        a = self._stage1.cpu().numpy()
        return [a[i] for i in range(n)]

    def dit_tiles(self, init, first, step):
        n, h, w = self.shape
        k = len(range(first, self.count(h, w), step))
        tl = self.tile_size // 8
        x0 = torch.empty((max(k, 1), n, 4, tl, tl), dtype=torch.float32, device=self.device)
        ws, c = self._ws(), self.ctx
        c.check(c.lib.ir_tiled_dit(c.h, c.stream(), L.ptr(init), L.ptr(x0), n, h, w, self.tile_size, self.tile_stride, first, step, 400.0, self.acp,
                                   self.flags, L.ptr(ws), ws.numel()), "ir_tiled_dit")
        return x0[:k]

    def blend_latent(self, x0_all):
        n, h, w = self.shape
        nb = torch.empty((n, 4, h // 8, w // 8), dtype=torch.float32, device=self.device)
        c = self.ctx
        c.check(c.lib.ir_tiled_blend_latent(c.h, c.stream(), L.ptr(x0_all.contiguous()), L.ptr(nb), n, h, w, self.tile_size, self.tile_stride),
                "ir_tiled_blend_latent")
        return nb

    def decode_tiles(self, nb, control, first, step):
        n, h, w = self.shape
        k = len(range(first, self.count(h, w), step))
        tp = (self.tile_size // 8) * 8
        px = torch.empty((max(k, 1), n, 3, tp, tp), dtype=torch.float32, device=self.device)
        ws, c = self._ws(), self.ctx
        c.check(c.lib.ir_tiled_decode(c.h, c.stream(), L.ptr(nb), L.ptr(control), L.ptr(px), n, h, w, self.tile_size, self.tile_stride, first, step,
                                      self.flags, self.sf, L.ptr(ws), ws.numel()), "ir_tiled_decode")
        return px[:k]

    def blend_pixels(self, px_all):
        n, h, w = self.shape
        out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=self.device)
        ws, c = self._ws(), self.ctx
        c.check(c.lib.ir_tiled_blend_pixels(c.h, c.stream(), L.ptr(px_all.contiguous()), L.ptr(out), n, h, w, self.tile_size, self.tile_stride,
                                            L.ptr(ws), ws.numel()), "ir_tiled_blend_pixels")
        a = out.cpu().numpy()
        return [a[i] for i in range(n)]
