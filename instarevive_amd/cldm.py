"""Host-side mirror of the ControlLDM one-step path: the objects diffusion/cldm.py builds from configs/cldm.yaml,
with the same names, constructor arguments and call signatures, every forward a call through the C ABI (no torch.nn arithmetic, no CPU
fallback).

  ControlledUnetModel  <- diffusion/cldm.py:32-55   (UNetModel of ldm/modules/diffusionmodules/openaimodel.py:411-786)
  ControlNet           <- diffusion/cldm.py:58-292
  Reflow_ControlLDM    <- diffusion/cldm.py:443-588 (apply_condition_encoder, apply_model, sample_log, decode_first_stage, log_images)

  FrozenOpenCLIPEmbedder <- ldm/modules/encoders/modules.py:134-196 (the `cond_stage_model`: open_clip's ViT-H-14 text tower; its BPE
                          tokenizer table is not in this image - the empty prompt the reference samples with is tokenised, other
                          prompts need a tokenizer callable)

The text conditioning of a call is `c_crossattn` [B, 77, context_dim] (from get_learned_conditioning / get_unconditional_conditioning or a
saved embedding). One context per call: all rows of a batch must carry the same embedding.
"""
import ctypes as C
from types import SimpleNamespace

import os

import torch

from . import _lib as L
from . import weights as W
from .models import AutoencoderKL, SwinIR, _DeviceModule, _ints

_IGNORED = dict(image_size=None, dropout=0, conv_resample=True, dims=2, use_checkpoint=False, use_fp16=False, num_heads=-1, num_heads_upsample=-1,
                use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False, n_embed=None, disable_self_attentions=None,
                num_attention_blocks=None, disable_middle_self_attn=False, num_classes=None)


class _UNetBase(_DeviceModule):
    CONTROL = False

    def __init__(self, image_size=32, in_channels=4, model_channels=320, out_channels=4, hint_channels=4, num_res_blocks=2,
                 attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 4), num_head_channels=64, use_spatial_transformer=True,
                 use_linear_in_transformer=True, transformer_depth=1, context_dim=1024, legacy=False, **kw):
        super().__init__()
        bad = [k for k, v in kw.items() if k not in _IGNORED or (k not in ("image_size", "use_checkpoint", "dropout") and v != _IGNORED[k])]
        levels = [l for l in range(len(channel_mult)) if 2 ** l in set(attention_resolutions)]
        if (bad or not use_spatial_transformer or not use_linear_in_transformer or transformer_depth != 1 or legacy or in_channels != 4 or
                out_channels != 4 or hint_channels != 4 or num_head_channels not in (32, 64) or not isinstance(num_res_blocks, int) or
                model_channels % 32 or context_dim % 32 or any(a not in [2 ** l for l in range(len(channel_mult))] for a in attention_resolutions)):
            raise NotImplementedError("the MI355X path implements the UNet / ControlNet variant of configs/cldm.yaml: spatial transformers of depth 1 "
                                      f"with linear projections, num_head_channels 32 or 64, 4 latent channels, legacy=False (unsupported: {bad})")
        self.cfg = dict(model_channels=model_channels, channel_mult=list(channel_mult), num_res_blocks=num_res_blocks,
                        attention_resolutions=list(attention_resolutions), num_head_channels=num_head_channels, context_dim=context_dim,
                        in_channels=4, hint_channels=4, out_channels=4)
        self.attention_levels = sum(1 << l for l in levels)
        self.model_channels = model_channels

    def _expected_keys(self):
        return W.unet_expected_keys(self.cfg, self.CONTROL)

    def load_state_dict(self, state_dict, strict=True):
        res = self._check_keys(state_dict, strict)
        self._sd = {k: v.detach().cpu() for k, v in state_dict.items()}
        if self.ctx is not None:
            self._upload()
        return res

    def _upload(self):
        c = self.cfg
        self.ctx.upload_all(W.pack_unet(self._sd, c, self.CONTROL))
        self.ctx.check(self.ctx.lib.ir_unet_configure(self.ctx.h, 1 if self.CONTROL else 0, c["model_channels"], len(c["channel_mult"]),
                                                      _ints(c["channel_mult"]), c["num_res_blocks"], self.attention_levels, c["num_head_channels"],
                                                      c["context_dim"], 8 if self.CONTROL else 4), "ir_unet_configure")
        self.ctx.__dict__.pop("_unet_context", None)   # the K / V caches of the cross-attentions went with the old binding
        self._mark_bound()


class ControlledUnetModel(_UNetBase):
    FAMILY = "unet"

    @torch.no_grad()
    def __call__(self, x, timesteps=None, context=None, control=None, only_mid_control=False, **kwargs):
        """UNet alone (control=None). A list of control tensors cannot be fed from the host: the controlled form runs inside
        Reflow_ControlLDM.sample_log / apply_model, where the ControlNet writes its residuals straight into the decoder's buffers."""
        if control is not None:
            raise NotImplementedError("pass c_latent through Reflow_ControlLDM.apply_model / sample_log: control residuals never leave the device")
        self._ready()
        return _sample(self.ctx, x, None, timesteps, context, add_x=False)

    forward = __call__


class ControlNet(_UNetBase):
    FAMILY = "cnet"
    CONTROL = True


def _set_context(ctx, context):
    """Bind c_crossattn [B, n_tok, context_dim] (all rows equal) as the context of every cross-attention. Cached by tensor OBJECT and version
    counter - the cache keeps the tensor alive, so its address cannot be handed to another tensor meanwhile (a (data_ptr, shape) key once took
    a new embedding that reused a freed tensor's memory for the old one); any other tensor, equal or not, rebuilds the K / V caches (~1 ms)."""
    cached = ctx.__dict__.get("_unet_context")
    if cached is not None and cached[0] is context and cached[1] == context._version:
        return
    c = context.detach().to("cpu", torch.float32)
    if c.dim() != 3 or not all(torch.equal(c[0], c[i]) for i in range(1, c.shape[0])):
        raise ValueError("c_crossattn must be [B, n_tok, context_dim] with the same embedding in every row (one prompt per call)")
    c0 = c[0].contiguous()
    ctx.check(ctx.lib.ir_unet_set_context(ctx.h, ctx.stream(), C.c_void_p(c0.data_ptr()), c0.shape[0]), "ir_unet_set_context")
    ctx.__dict__["_unet_context"] = (context, context._version)


def _cat1(tensors):
    """torch.cat(tensors, 1) without a copy for the usual one-element list (the copy would be a new object: a context-cache miss per call)."""
    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, 1)


def _timestep(timesteps):
    t = torch.as_tensor(timesteps, dtype=torch.float32).reshape(-1)
    if t.numel() > 1 and not bool((t == t[0]).all()):
        raise NotImplementedError("one timestep per call (the one-step sampler uses num_timesteps - 1 for the whole batch)")
    return float(t[0])


def _sample(ctx, x, c_latent, timesteps, context, add_x):
    dev = ctx.device
    x = x.to(dev, torch.float32).contiguous()
    n, ch, h, w = x.shape
    if ch != 4:
        raise ValueError(f"latent must be [B,4,h,w], got {tuple(x.shape)}")
    _set_context(ctx, context)
    hint = None if c_latent is None else c_latent.to(dev, torch.float32).contiguous()
    if hint is not None and hint.shape != x.shape:
        raise ValueError(f"c_latent {tuple(hint.shape)} must match the latent {tuple(x.shape)}")
    out = torch.empty_like(x)
    ws = ctx.workspace(ctx.ws_bytes(L.STAGE_CLDM, n, h, w))
    ctx.check(ctx.lib.ir_cldm_sample(ctx.h, ctx.stream(), L.ptr(x), L.ptr(hint) if hint is not None else None, L.ptr(out), n, h, w,
                                     _timestep(timesteps), 0 if add_x else 1, L.ptr(ws), ws.numel()), "ir_cldm_sample")
    return out   # zT + v (cldm.py:588), or v alone


class FrozenOpenCLIPEmbedder(_DeviceModule):
    """ldm/modules/encoders/modules.py:134-196 (cldm.yaml: cond_stage_config, layer "penultimate"): the text tower of open_clip's ViT-H-14.
    `arch` / `version` select nothing here: the weights come through load_state_dict with open_clip's parameter names (the checkpoint's
    `cond_stage_model.model.*` entries). open_clip's BPE table (bpe_simple_vocab_16e6.txt.gz) is not in this image: pass `tokenizer` = a
    folder holding it (or the Hugging Face CLIP vocab.json + merges.txt; instarevive_amd/clip_bpe.py restates open_clip.tokenize over it) or
    any callable texts -> LongTensor [B, 77]; without one only the empty prompt - the only one the reference's samplers use
    (cldm.py:357-358, positive_prompt="") - is tokenised (<start_of_text>, <end_of_text>, zero padding)."""
    FAMILY = "clip"
    SOT, EOT = 49406, 49407

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", max_length=77, freeze=True, layer="last", *, width=1024, heads=16, layers=24,
                 vocab_size=49408, mlp_ratio=4.0, tokenizer=None):
        super().__init__()
        if layer not in ("last", "penultimate"):
            raise AssertionError(layer)
        self.cfg = dict(width=width, heads=heads, layers=layers, vocab_size=vocab_size, context_length=max_length, mlp_ratio=mlp_ratio)
        if isinstance(tokenizer, (str, os.PathLike)):   # a folder with open_clip's BPE table (or the Hugging Face CLIP tokenizer files)
            from .clip_bpe import ClipBPETokenizer
            tokenizer = ClipBPETokenizer.from_folder(os.fspath(tokenizer), max_length)
        self.layer, self.layer_idx, self.max_length, self.tokenizer = layer, (0 if layer == "last" else 1), max_length, tokenizer

    def _expected_keys(self):
        return W.clip_text_expected_keys(self.cfg)

    def load_state_dict(self, state_dict, strict=True):
        sd = {k[6:] if k.startswith("model.") else k: v for k, v in state_dict.items()}
        res = self._check_keys(sd, strict, ignore=("visual.", "text_projection", "logit_scale", "attn_mask"))
        self._sd = {k: v.detach().cpu() for k, v in sd.items()}
        if self.ctx is not None:
            self._upload()
        return res

    def _upload(self):
        c = self.cfg
        n_run = c["layers"] - self.layer_idx
        self.ctx.upload_all(W.pack_clip_text(self._sd, c, n_run))
        self.ctx.check(self.ctx.lib.ir_clip_text_configure(self.ctx.h, n_run, c["width"], c["heads"], int(c["width"] * c["mlp_ratio"]), c["vocab_size"],
                                                           c["context_length"]), "ir_clip_text_configure")
        self._mark_bound()

    def tokenize(self, text):
        if self.tokenizer is not None:
            return self.tokenizer(text)
        if any(t != "" for t in text):
            raise NotImplementedError("open_clip's BPE table is not part of this build: construct FrozenOpenCLIPEmbedder(tokenizer=<folder with "
                                      "bpe_simple_vocab_16e6.txt.gz, or vocab.json + merges.txt>) for non-empty prompts")
        ids = torch.zeros(len(text), self.max_length, dtype=torch.long)
        ids[:, 0], ids[:, 1] = self.SOT, self.EOT
        return ids

    @torch.no_grad()
    def encode_with_transformer(self, tokens):
        self._ready()
        ids = tokens.to(self.device, torch.int32).contiguous()
        b, t = ids.shape
        if t != self.max_length:
            raise ValueError(f"tokens must be [B, {self.max_length}], got {tuple(ids.shape)}")
        out = torch.empty(b, t, self.cfg["width"], dtype=torch.float32, device=self.device)
        ws = self.ctx.workspace(self.ctx.ws_bytes(L.STAGE_CLIP_TEXT, b, 0, 0))
        self.ctx.check(self.ctx.lib.ir_clip_text_encode(self.ctx.h, self.ctx.stream(), L.ptr(ids), L.ptr(out), b, L.ptr(ws), ws.numel()), "ir_clip_text_encode")
        return out

    def __call__(self, text):
        return self.encode_with_transformer(self.tokenize(list(text)))

    forward = encode = __call__


class Reflow_ControlLDM:
    """Reflow_ControlLDM(control_stage_config, ..., unet_config, first_stage_config, preprocess_config) of configs/cldm.yaml. Each *_config is
    the yaml node ({'target': ..., 'params': {...}}) or its params dict."""

    def __init__(self, control_stage_config, unet_config, first_stage_config=None, preprocess_config=None, control_key="hint", sd_locked=True,
                 only_mid_control=False, learning_rate=None, lora_rank=None, output="./", timesteps=1000, scale_factor=0.18215, channels=4,
                 cond_stage_config=None, **unused):
        if only_mid_control:
            raise NotImplementedError("only_mid_control=True is not built (cldm.yaml: False)")
        par = lambda node: dict(node.get("params", node)) if node else {}
        self.model = SimpleNamespace(diffusion_model=ControlledUnetModel(**par(unet_config)))
        self.control_model = ControlNet(**par(control_stage_config))
        dd = par(first_stage_config).get("ddconfig", {})
        ch, mult = dd.get("ch", 128), list(dd.get("ch_mult", (1, 2, 4, 4)))
        # ONE device VAE binding: the encoder half holds cond_encoder.* (cldm.py:476-480), the decoder half first_stage_model.decoder
        self.first_stage_model = AutoencoderKL(block_out_channels=[ch * m for m in mult], layers_per_block=dd.get("num_res_blocks", 2),
                                               scaling_factor=scale_factor)
        self.cond_encoder = self.first_stage_model
        self.preprocess_model = SwinIR(**par(preprocess_config)) if preprocess_config else None
        self.cond_stage_model = FrozenOpenCLIPEmbedder(**par(cond_stage_config)) if cond_stage_config else None
        self.control_key, self.only_mid_control, self.control_scales = control_key, only_mid_control, [1.0] * 13
        self.num_timesteps, self.scale_factor, self.channels = timesteps, scale_factor, channels
        self.device, self.ctx = torch.device("cpu"), None

    # ---- weights
    def load_state_dict(self, state_dict, strict=True):
        """The checkpoint layout of the reference module: model.diffusion_model.*, control_model.*, cond_encoder.{encoder,quant_conv}.*,
        first_stage_model.{decoder,post_quant_conv}.*, preprocess_model.*. first_stage_model.encoder (training only), cond_stage_model
        (the text encoder, see the module docstring) and the diffusion schedule buffers are not on the path and are ignored."""
        def sub(prefix):
            return {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        self.model.diffusion_model.load_state_dict(sub("model.diffusion_model."), strict)
        self.control_model.load_state_dict(sub("control_model."), strict)
        vae = {**{k: v for k, v in sub("cond_encoder.").items()}, **{k: v for k, v in sub("first_stage_model.").items() if not k.startswith(("encoder.", "quant_conv."))}}
        nl = len(self.first_stage_model.cfg["ch_mult"])
        self.first_stage_model.load_state_dict(W.vae_ldm_to_diffusers(vae, nl, self.first_stage_model.cfg["num_res_blocks"]), strict)
        if self.preprocess_model is not None:
            self.preprocess_model.load_state_dict(sub("preprocess_model."), strict)
        if self.cond_stage_model is not None and any(k.startswith("cond_stage_model.") for k in state_dict):
            self.cond_stage_model.load_state_dict(sub("cond_stage_model."), strict)
        return SimpleNamespace(missing_keys=[], unexpected_keys=[])

    def to(self, device):
        for m in (self.model.diffusion_model, self.control_model, self.first_stage_model, self.preprocess_model, self.cond_stage_model):
            if m is not None:
                m.to(device)
        self.ctx, self.device = self.model.diffusion_model.ctx, self.model.diffusion_model.device
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def eval(self):
        return self

    def _ready(self):
        for m in (self.model.diffusion_model, self.control_model, self.first_stage_model):
            m._ready()

    # ---- the reference's methods
    def get_learned_conditioning(self, c):
        """LatentDiffusion.get_learned_conditioning: cond_stage_model.encode(c) (the frozen OpenCLIP text tower)."""
        if self.cond_stage_model is None or self.cond_stage_model._sd is None:
            raise NotImplementedError("no cond_stage_model weights loaded: pass the prompt embedding as c_crossattn")
        return self.cond_stage_model.encode(c)

    def get_unconditional_conditioning(self, N):   # cldm.py:529-530
        return self.get_learned_conditioning([""] * N)

    @torch.no_grad()
    def apply_condition_encoder(self, control):
        """cldm.py:486-490: mode of the condition encoder's posterior on control * 2 - 1, times scale_factor. control: [B,3,H,W] in [0,1]."""
        return self.cond_encoder.encode(control.to(self.device, torch.float32) * 2 - 1).latent_dist.mode() * self.scale_factor

    @torch.no_grad()
    def apply_model(self, x_noisy, t, cond, *args, **kwargs):
        """cldm.py:511-527: eps = diffusion_model(x, t, context, control = control_model(x, hint = c_latent, t, context) * control_scales)."""
        self._ready()
        c_latent = None if cond["c_latent"] is None else torch.cat(cond["c_latent"], 1)
        return _sample(self.ctx, x_noisy, c_latent, t, _cat1(cond["c_crossattn"]), add_x=False)

    @torch.no_grad()
    def sample_log(self, cond, steps=1, *, zT=None):
        """cldm.py:568-588: zT ~ N(0, I) of the latent shape of cond['c_concat'][0]; returns zT + v at t = num_timesteps - 1. `zT` (an
        extension) fixes the noise."""
        self._ready()
        b, _, h, w = cond["c_concat"][0].shape
        if zT is None:
            zT = torch.randn(b, self.channels, h // 8, w // 8, device=self.device)
        c_latent = None if cond["c_latent"] is None else torch.cat(cond["c_latent"], 1)
        return _sample(self.ctx, zT, c_latent, float(self.num_timesteps - 1), _cat1(cond["c_crossattn"]), add_x=True)

    @torch.no_grad()
    def decode_first_stage(self, z):
        """LatentDiffusion.decode_first_stage: first_stage_model.decode(z / scale_factor)."""
        return self.first_stage_model.decode(z.to(self.device, torch.float32) / self.scale_factor).sample

    @torch.no_grad()
    def log_images(self, batch, sample_steps=50, *, zT=None, c_crossattn=None, graph=False):
        """cldm.py:536-566 on the inference inputs: batch[control_key] is [B,H,W,3] in [0,1] (the LQ image; H, W multiples of 64);
        c_crossattn the prompt embedding ([77, D] or [B,77,D]; batch['c_crossattn'] if not given). The whole chain (SwinIR preprocess ->
        condition encoder -> ControlNet + UNet -> decoder) is ONE ir_cldm_pipeline call. Returns {'lq', 'control', 'samples'}.
        graph=True: the step's ~1000 launches are recorded into a hipGraph per batch shape and replayed on later calls (IR_FLAG_GRAPH; at 512 x 512
        18.7 instead of 20.5 ms per image). A recorded graph addresses fixed buffers, so inputs are copied into, and results returned as copies
        of, device buffers kept per shape; same bits as graph=False."""
        self._ready()
        if self.preprocess_model is None:
            raise RuntimeError("log_images needs the preprocess_model (SwinIR)")
        self.preprocess_model._ready()
        lq = batch[self.control_key].to(self.device, torch.float32).permute(0, 3, 1, 2).contiguous()
        n, ch, h, w = lq.shape
        if ch != 3 or h % 64 or w % 64:
            raise ValueError(f"{self.control_key} must be [B,H,W,3] with H, W multiples of 64, got {tuple(batch[self.control_key].shape)}")
        ctxt = c_crossattn if c_crossattn is not None else batch["c_crossattn"]
        ctxt = ctxt if ctxt.dim() == 3 else ctxt[None]
        _set_context(self.ctx, ctxt)
        if zT is None:
            zT = torch.randn(n, self.channels, h // 8, w // 8, device=self.device)
        zT = zT.to(self.device, torch.float32).contiguous()
        ws = self.ctx.workspace(self.ctx.ws_bytes(L.STAGE_CLDM_PIPELINE, n, h, w))
        if graph:
            bufs = self.__dict__.setdefault("_graph_bufs", {})
            key = (n, h, w)
            if key not in bufs:
                bufs[key] = (torch.empty_like(lq), torch.empty_like(zT), torch.empty_like(lq), torch.empty_like(lq))
            lq_s, zT_s, samples_s, control_s = bufs[key]
            lq_s.copy_(lq)
            zT_s.copy_(zT)
            self.ctx.check(self.ctx.lib.ir_cldm_pipeline(self.ctx.h, self.ctx.stream(), L.ptr(lq_s), L.ptr(zT_s), L.ptr(samples_s), L.ptr(control_s), n, h, w,
                                                         L.FLAG_GRAPH, float(self.num_timesteps - 1), float(self.scale_factor), L.ptr(ws), ws.numel()),
                           "ir_cldm_pipeline")
            return dict(lq=lq, control=control_s.clone(), samples=samples_s.clone())
        control, samples = torch.empty_like(lq), torch.empty_like(lq)
        self.ctx.check(self.ctx.lib.ir_cldm_pipeline(self.ctx.h, self.ctx.stream(), L.ptr(lq), L.ptr(zT), L.ptr(samples), L.ptr(control), n, h, w, 0,
                                                     float(self.num_timesteps - 1), float(self.scale_factor), L.ptr(ws), ws.numel()), "ir_cldm_pipeline")
        return dict(lq=lq, control=control, samples=samples)
