"""Host-side mirrors of the objects test_scripts/inference.py builds in main() (:236-252) and calls in process()
(:97,106-117): same constructor/loader names, argument meaning and error behaviour, but every forward is a call
through the C ABI into HIP kernels. No torch.nn arithmetic happens here and there is no CPU fallback.

  SwinIR            <- diffusion/model/swinir.py:629-905 (instantiate_from_config(configs/swinir.yaml) + load_state_dict)
  AutoencoderKL     <- diffusers.AutoencoderKL  (inference.py:34,236-237: .encode(x).latent_dist.mode(), .decode(z).sample,
                       .config.scaling_factor)
  Transformer2DModel<- diffusers.Transformer2DModel (inference.py:238-242; generate.py:56,67-73,84: .config.sample_size,
                       .config.out_channels, __call__(...).sample)
  DDPMScheduler     <- diffusers.DDPMScheduler (inference.py:36; generate.py:45: .alphas_cumprod)
"""
import ctypes as C
import json
import os
from types import SimpleNamespace

import torch

from . import _lib as L
from . import weights as W

_CONTEXTS = {}


def get_context(device=None) -> L.Context:
    """One ir_ctx per GPU, shared by all models on that device (so the fused pipeline sees all weights)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    device = torch.device(device)
    if device.type != "cuda":
        raise L.NativeLibraryError(f"instarevive_amd runs on MI355X GPUs only (asked for device '{device}'); there is no CPU path")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _CONTEXTS:
        _CONTEXTS[idx] = L.Context(idx)
    return _CONTEXTS[idx]


def _ints(v):
    return (C.c_int * len(v))(*v)


class _DeviceModule:
    """Minimal nn.Module-like surface: state dict on the host, .to(device) uploads and binds."""

    def __init__(self):
        self._sd = None
        self.ctx = None
        self.device = torch.device("cpu")
        self.training = False

    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise RuntimeError("instarevive_amd models are inference-only")
        return self

    def requires_grad_(self, flag=False):
        return self

    def parameters(self):
        return iter(self._sd.values()) if self._sd else iter(())

    def state_dict(self):
        return dict(self._sd) if self._sd is not None else {k: None for k in self._expected_keys()}

    def to(self, *args, **kwargs):
        device = kwargs.get("device")
        for a in args:
            if isinstance(a, (str, torch.device)) or isinstance(a, int):
                device = a
            # dtypes are accepted and ignored: storage precision is fixed by the kernels (bf16 weights, fp32 statistics)
        if device is not None:
            self.device = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
            self.ctx = get_context(self.device)
            self.device = self.ctx.device
            if self._sd is not None:
                self._upload()
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def _check_keys(self, sd, strict, ignore=()):
        exp = set(self._expected_keys())
        got = set(sd.keys())
        missing = sorted(k for k in exp - got if not any(s in k for s in ignore))
        unexpected = sorted(k for k in got - exp if not any(s in k for s in ignore))
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}:\n\tMissing key(s): {missing[:8]}"
                               f"{'...' if len(missing) > 8 else ''}\n\tUnexpected key(s): {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    FAMILY = None  # name of the weight set in the per-GPU context ("swin", "vae", "dit", ...): one binding per family and device

    def _mark_bound(self):
        self.ctx.__dict__.setdefault("_bound", {})[self.FAMILY] = self

    def _is_bound(self):
        return self.ctx.__dict__.get("_bound", {}).get(self.FAMILY) is self

    def _ready(self):
        if self.ctx is None or self._sd is None:
            raise RuntimeError(f"{type(self).__name__}: load_state_dict(...) and .to('cuda') must both be called before forward")
        if not self._is_bound():  # another model of the same family was loaded on this GPU since: the context holds ITS weights
            self._upload()


# ====================================================================================================== SwinIR
class SwinIR(_DeviceModule):
    FAMILY = "swin"

    def __init__(self, img_size=64, patch_size=1, in_chans=3, embed_dim=96, depths=(6, 6, 6, 6), num_heads=(6, 6, 6, 6), window_size=7,
                 mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=None,
                 ape=False, patch_norm=True, use_checkpoint=False, sf=4, img_range=1.0, upsampler="", resi_connection="1conv",
                 unshuffle=False, unshuffle_scale=None, hq_key="jpg", lq_key="hint", learning_rate=None, weight_decay=None):
        super().__init__()
        ok = (patch_size == 1 and in_chans == 3 and window_size == 8 and sf == 8 and upsampler == "nearest+conv" and
              resi_connection == "1conv" and unshuffle and unshuffle_scale == 8 and qkv_bias and qk_scale is None and not ape and patch_norm and
              len(set(num_heads)) == 1 and embed_dim % num_heads[0] == 0 and embed_dim // num_heads[0] <= 32)
        if not ok:
            raise NotImplementedError("the MI355X path implements the SwinIR variant of configs/swinir.yaml: window 8, sf 8, "
                                      "'nearest+conv', PixelUnshuffle(8), '1conv', head_dim <= 32")
        self.cfg = dict(embed_dim=embed_dim, depths=list(depths), num_heads=list(num_heads), window_size=8, mlp_ratio=mlp_ratio, sf=8,
                        img_range=img_range, unshuffle_scale=8)
        self.upscale, self.window_size = sf, window_size

    def _expected_keys(self):
        return W.swinir_expected_keys(self.cfg)

    def load_state_dict(self, state_dict, strict=True):
        # relative_position_index / attn_mask are buffers derived from the window geometry: the released checkpoint carries
        # them (732 entries) and they are accepted, but they are regenerated on the device and may be absent
        res = self._check_keys(state_dict, strict, ignore=("relative_position_index", "attn_mask"))
        self._sd = {k: v.detach().cpu() for k, v in state_dict.items()}
        if self.ctx is not None:
            self._upload()
        return res

    def _upload(self):
        c = self.cfg
        self.ctx.upload_all(W.pack_swinir(self._sd, c))
        mean = (C.c_float * 3)(*W.SWIN_MEAN)
        self.ctx.check(self.ctx.lib.ir_swinir_configure(self.ctx.h, c["embed_dim"], len(c["depths"]), _ints(c["depths"]), c["num_heads"][0],
                                                         int(c["embed_dim"] * c["mlp_ratio"]), 64, float(c["img_range"]), mean),
                       "ir_swinir_configure")
        self._mark_bound()

    @torch.no_grad()
    def __call__(self, x):
        """x: [B,3,H,W] fp32 cuda in [0,1], H and W multiples of 64 (the CLI pads to 64: inference.py:287,291)."""
        self._ready()
        x = x.to(self.device, torch.float32).contiguous()
        n, ch, h, w = x.shape
        if ch != 3 or h % 64 or w % 64:
            raise ValueError(f"SwinIR input must be [B,3,H,W] with H,W multiples of 64, got {tuple(x.shape)}")
        out = torch.empty_like(x)
        ws = self.ctx.workspace(self.ctx.ws_bytes(L.STAGE_SWINIR, n, h, w))
        self.ctx.check(self.ctx.lib.ir_swinir_forward(self.ctx.h, self.ctx.stream(), L.ptr(x), L.ptr(out), n, h, w, L.ptr(ws), ws.numel()),
                       "ir_swinir_forward")
        return out

    forward = __call__


# ====================================================================================================== VAE
def _load_weights_file(folder):
    """The state dict of a diffusers (diffusion_pytorch_model.*) or transformers (model.safetensors / pytorch_model.bin, single file or
    sharded behind an *.index.json as DeepFloyd/t5-v1_1-xxl ships it) weight folder."""
    def load_one(path):
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file
            return load_file(path)
        return torch.load(path, map_location="cpu")

    for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin", "diffusion_pytorch_model.pt", "model.safetensors",
                 "pytorch_model.bin"):
        path = os.path.join(folder, name)
        if os.path.exists(path):
            return load_one(path)
    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json", "diffusion_pytorch_model.safetensors.index.json"):
        path = os.path.join(folder, index)
        if os.path.exists(path):
            with open(path) as f:
                shards = sorted(set(json.load(f)["weight_map"].values()))
            sd = {}
            for shard in shards:
                sd.update(load_one(os.path.join(folder, shard)))
            return sd
    raise FileNotFoundError(f"no diffusion_pytorch_model.* / model.safetensors / pytorch_model.bin (or a sharded index) under {folder}")


def _hf_cache_dirs():
    """Where huggingface_hub keeps downloaded repositories, in its own order of precedence (no network access is attempted)."""
    env = os.environ
    dirs = [env.get("HF_HUB_CACHE"), env.get("HUGGINGFACE_HUB_CACHE")]
    if env.get("HF_HOME"):
        dirs.append(os.path.join(env["HF_HOME"], "hub"))
    dirs.append(os.path.join(env.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "huggingface", "hub"))
    return [d for i, d in enumerate(dirs) if d and d not in dirs[:i]]


def _hf_snapshots(repo_id):
    """Snapshot folders of a hub id (`org/name`) in the local hub cache layout `models--org--name/snapshots/<rev>/`: the revision
    `refs/main` names first (what from_pretrained() resolves without a revision argument), then the others, newest first."""
    found = []
    for cache in _hf_cache_dirs():
        repo = os.path.join(cache, "models--" + repo_id.strip("/").replace("/", "--"))
        snaps = os.path.join(repo, "snapshots")
        if not os.path.isdir(snaps):
            continue
        revs = sorted((d for d in os.listdir(snaps) if os.path.isdir(os.path.join(snaps, d))),
                      key=lambda d: os.path.getmtime(os.path.join(snaps, d)), reverse=True)
        ref = os.path.join(repo, "refs", "main")
        if os.path.isfile(ref):
            with open(ref) as f:
                main = f.read().strip()
            if main in revs:
                revs.remove(main)
                revs.insert(0, main)
        found += [os.path.join(snaps, d) for d in revs]
    return found


def _resolve_pretrained(name_or_path, subfolder=None):
    """A local folder, `./weights/<name>`, or a hub id (`stabilityai/sd-vae-ft-ema`, `PixArt-alpha/PixArt-Alpha-DMD-XL-2-512x512`: what
    the reference passes to from_pretrained, test_scripts/inference.py:36,236,238) resolved inside the local Hugging Face hub cache
    ($HF_HUB_CACHE, $HF_HOME/hub, ~/.cache/huggingface/hub) - the place the reference's own run left the weights. Nothing is downloaded."""
    cands = [name_or_path, os.path.join("weights", name_or_path), os.path.join("weights", os.path.basename(name_or_path))]
    if not os.path.isabs(name_or_path) and name_or_path.count("/") == 1 and not name_or_path.startswith("."):
        cands += _hf_snapshots(name_or_path)
    for c in cands:
        p = os.path.join(c, subfolder) if subfolder else c
        if os.path.isdir(p):
            return p
    raise FileNotFoundError(f"'{name_or_path}'{' / ' + subfolder if subfolder else ''} not found locally (looked in {cands[:3]} and the hub caches "
                            f"{_hf_cache_dirs()}); download the diffusers folder and pass its path")


class _LatentDist:
    def __init__(self, mean):
        self.mean = mean

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        raise NotImplementedError("the one-step path only uses latent_dist.mode() (test_scripts/inference.py:107)")


class AutoencoderKL(_DeviceModule):
    FAMILY = "vae"

    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4,
                 norm_num_groups=32, scaling_factor=0.18215, **unused):
        super().__init__()
        ch = block_out_channels[0]
        mult = [c // ch for c in block_out_channels]
        if in_channels != 3 or out_channels != 3 or latent_channels != 4 or norm_num_groups != 32 or ch % 32 or any(c % ch for c in block_out_channels):
            raise NotImplementedError("unsupported AutoencoderKL config for the MI355X path")
        self.cfg = dict(ch=ch, ch_mult=mult, num_res_blocks=layers_per_block, z_channels=4)
        self.config = SimpleNamespace(scaling_factor=scaling_factor, block_out_channels=list(block_out_channels), latent_channels=4,
                                      layers_per_block=layers_per_block)

    @classmethod
    def from_pretrained(cls, name_or_path, subfolder=None, **kw):
        folder = _resolve_pretrained(name_or_path, subfolder)
        with open(os.path.join(folder, "config.json")) as f:
            cfg = json.load(f)
        m = cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})
        m.load_state_dict(_load_weights_file(folder))
        return m

    def _expected_keys(self):
        keys = []
        c = self.cfg
        nl = len(c["ch_mult"])

        def res(p, sc):
            r = [p + f".{n}.{t}" for n in ("norm1", "conv1", "norm2", "conv2") for t in ("weight", "bias")]
            return r + ([p + ".conv_shortcut.weight", p + ".conv_shortcut.bias"] if sc else [])

        def attn(p):
            return [p + f".{n}.{t}" for n in ("group_norm", "to_q", "to_k", "to_v", "to_out.0") for t in ("weight", "bias")]

        for half in ("encoder", "decoder"):
            keys += [f"{half}.{n}.{t}" for n in ("conv_in", "conv_norm_out", "conv_out") for t in ("weight", "bias")]
            keys += res(f"{half}.mid_block.resnets.0", False) + res(f"{half}.mid_block.resnets.1", False) + attn(f"{half}.mid_block.attentions.0")
        cin = c["ch"]
        for l in range(nl):
            cout = c["ch"] * c["ch_mult"][l]
            for j in range(c["num_res_blocks"]):
                keys += res(f"encoder.down_blocks.{l}.resnets.{j}", cin != cout)
                cin = cout
            if l != nl - 1:
                keys += [f"encoder.down_blocks.{l}.downsamplers.0.conv.weight", f"encoder.down_blocks.{l}.downsamplers.0.conv.bias"]
        for i in range(nl):
            cout = c["ch"] * c["ch_mult"][nl - 1 - i]
            for j in range(c["num_res_blocks"] + 1):
                keys += res(f"decoder.up_blocks.{i}.resnets.{j}", cin != cout)
                cin = cout
            if i != nl - 1:
                keys += [f"decoder.up_blocks.{i}.upsamplers.0.conv.weight", f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"]
        keys += ["quant_conv.weight", "quant_conv.bias", "post_quant_conv.weight", "post_quant_conv.bias"]
        return keys

    def load_state_dict(self, state_dict, strict=True):
        sd = dict(state_dict)
        # accept the pre-0.18 attention names (query/key/value/proj_attn) some sd-vae-ft-ema snapshots still carry
        for k in list(sd):
            for old, new in ((".query.", ".to_q."), (".key.", ".to_k."), (".value.", ".to_v."), (".proj_attn.", ".to_out.0.")):
                if old in k and "attentions" in k:
                    sd[k.replace(old, new)] = sd.pop(k)
        res = self._check_keys(sd, strict)
        self._sd = {k: v.detach().cpu() for k, v in sd.items()}
        if self.ctx is not None:
            self._upload()
        return res

    def enable_fp8(self, on=True):
        """BASELINE.json configs[4]: run the ResnetBlock 3x3 convolutions on fp8 (OCP e4m3) operands. The fp8 weight forms are packed and
        uploaded on first use; process(..., fp8=True) switches the mode on per call, this switches it on for .encode() / .decode()."""
        self._fp8 = bool(on)
        if self.ctx is not None and self._sd is not None:
            if on and not self.__dict__.get("_fp8_uploaded"):
                self._upload()
            self.ctx.check(self.ctx.lib.ir_set_fp8(self.ctx.h, 1 if on else 0), "ir_set_fp8")
        return self

    def _upload(self):
        c = self.cfg
        want8 = bool(self.__dict__.get("_fp8"))
        self.ctx.upload_all(W.pack_vae(self._sd, c, fp8=want8))
        self._fp8_uploaded = want8
        self.ctx.check(self.ctx.lib.ir_vae_configure(self.ctx.h, c["ch"], len(c["ch_mult"]), _ints(c["ch_mult"]), c["num_res_blocks"], 1, 1),
                       "ir_vae_configure")
        self._mark_bound()

    @torch.no_grad()
    def encode(self, x):
        self._ready()
        x = x.to(self.device, torch.float32).contiguous()
        n, ch, h, w = x.shape
        if ch != 3 or h % 64 or w % 64:
            raise ValueError(f"VAE encode input must be [B,3,H,W] with H,W multiples of 64, got {tuple(x.shape)}")
        lat = torch.empty(n, 4, h // 8, w // 8, dtype=torch.float32, device=self.device)
        ws = self.ctx.workspace(self.ctx.ws_bytes(L.STAGE_VAE_ENCODE, n, h, w))
        self.ctx.check(self.ctx.lib.ir_vae_encode(self.ctx.h, self.ctx.stream(), L.ptr(x), L.ptr(lat), n, h, w, L.ptr(ws), ws.numel()),
                       "ir_vae_encode")
        return SimpleNamespace(latent_dist=_LatentDist(lat))

    @torch.no_grad()
    def decode(self, z):
        self._ready()
        z = z.to(self.device, torch.float32).contiguous()
        n, ch, h, w = z.shape
        if ch != 4 or h % 8 or w % 8:
            raise ValueError(f"VAE decode input must be [B,4,h,w] with h,w multiples of 8, got {tuple(z.shape)}")
        out = torch.empty(n, 3, h * 8, w * 8, dtype=torch.float32, device=self.device)
        ws = self.ctx.workspace(self.ctx.ws_bytes(L.STAGE_VAE_DECODE, n, h, w))
        self.ctx.check(self.ctx.lib.ir_vae_decode(self.ctx.h, self.ctx.stream(), L.ptr(z), L.ptr(out), n, h, w, L.ptr(ws), ws.numel()),
                       "ir_vae_decode")
        return SimpleNamespace(sample=out)


# ====================================================================================================== DiT
class Transformer2DModel(_DeviceModule):
    FAMILY = "dit"

    def __init__(self, num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, num_layers=28, cross_attention_dim=1152,
                 attention_bias=True, sample_size=64, patch_size=2, activation_fn="gelu-approximate", norm_type="ada_norm_single",
                 norm_elementwise_affine=False, norm_eps=1e-6, caption_channels=4096, interpolation_scale=None, mlp_ratio=4, kv_compress_config=None,
                 qk_norm=False, **unused):
        super().__init__()
        C_ = num_attention_heads * attention_head_dim
        if (in_channels != 4 or out_channels != 8 or patch_size != 2 or norm_type != "ada_norm_single" or activation_fn != "gelu-approximate" or
                norm_elementwise_affine or not attention_bias or attention_head_dim not in (32, 64, 72) or cross_attention_dim not in (None, C_)):
            raise NotImplementedError("the MI355X path implements the PixArt-alpha Transformer2DModel configuration "
                                      "(tools/convert_pixart_to_diffusers.py:163-180)")
        # sample_size 128 (PixArt-alpha 1024): diffusers builds the model with use_additional_conditions - `resolution` / `aspect_ratio` embeddings on top
        # of the timestep embedding (generate.py:56-62 passes them; in-tree twin: SizeEmbedder, PixArt_blocks.py:366-399). The library computes them from
        # the latent's height and width, which is what forward_model passes.
        # kv_compress_config / qk_norm: the optional branches of the in-tree self-attention (AttentionKVCompress, PixArt_blocks.py:60-158; PixArtMS.py:97-139
        # takes {'sampling': 'conv' | 'uniform' | 'ave', 'scale_factor': r, 'kv_compress_layer': [...]}); diffusers 0.30 has no counterpart, the keys follow
        # oracle.dit.pixart_to_diffusers (transformer_blocks.{d}.attn1.sr / norm / q_norm / k_norm)
        kvc = None
        if kv_compress_config and kv_compress_config.get("sampling") and int(kv_compress_config.get("scale_factor", 1)) > 1 and kv_compress_config.get("kv_compress_layer"):
            if kv_compress_config["sampling"] not in ("conv", "uniform", "ave"):
                raise NotImplementedError(f"kv_compress sampling {kv_compress_config['sampling']!r}: the MI355X path offers conv, uniform and ave (PixArt_blocks.py:97-121)")
            kvc = dict(sampling=kv_compress_config["sampling"], scale_factor=int(kv_compress_config["scale_factor"]),
                       layers=tuple(int(l) for l in kv_compress_config["kv_compress_layer"]))
        self.cfg = dict(num_layers=num_layers, micro=sample_size == 128, kv_compress=kvc, qk_norm=bool(qk_norm), num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                        sample_size=sample_size, caption_channels=caption_channels, mlp=mlp_ratio * C_,
                        interpolation_scale=float(interpolation_scale) if interpolation_scale is not None else float(max(sample_size // 64, 1)))
        self.config = SimpleNamespace(sample_size=sample_size, out_channels=out_channels, in_channels=in_channels, patch_size=patch_size,
                                      num_layers=num_layers, num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                                      caption_channels=caption_channels)
        self._prompt_key = None

    @classmethod
    def from_pretrained(cls, name_or_path, subfolder=None, **kw):
        folder = _resolve_pretrained(name_or_path, subfolder)
        with open(os.path.join(folder, "config.json")) as f:
            cfg = json.load(f)
        m = cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})
        try:
            m.load_state_dict(_load_weights_file(folder))
        except FileNotFoundError:
            pass  # the CLI overwrites the hub weights with InstaRevive_v1.ckpt anyway (inference.py:239-242)
        return m

    def _expected_keys(self):
        return W.dit_expected_keys(self.cfg)

    def load_state_dict(self, state_dict, strict=True):
        res = self._check_keys(state_dict, strict, ignore=("pos_embed.pos_embed",))
        self._sd = {k: v.detach().cpu() for k, v in state_dict.items()}
        if self.ctx is not None:
            self._upload()
        return res

    def _upload(self):
        c = self.cfg
        self.ctx.upload_all(W.pack_dit(self._sd, c))
        self.ctx.check(self.ctx.lib.ir_dit_configure(self.ctx.h, c["num_layers"], c["num_attention_heads"], c["attention_head_dim"], c["mlp"],
                                                      c["caption_channels"], c["sample_size"] // 2), "ir_dit_configure")
        self._prompt_key = None
        self.__dict__["_pos_done"] = set()
        self._mark_bound()
        self.ctx.__dict__["_bound"]["dit_ctrl"] = None  # ir_dit_configure drops a control branch bound to the previous model

    def ensure_pos(self, gh, gw):
        name = f"dit.pos.{gh}x{gw}"
        done = self.__dict__.setdefault("_pos_done", set())  # per model: the table depends on its width and base grid, not only on the name
        if (gh, gw) not in done:
            done.add((gh, gw))
            c = self.cfg
            self.ctx.upload(name, W.sincos_pos_embed(c["num_attention_heads"] * c["attention_head_dim"], gh, gw, c["sample_size"] // 2,
                                                     c["interpolation_scale"]))

    def set_prompt(self, encoder_hidden_states, encoder_attention_mask=None):
        """Project the caption and cache all layers' cross-attention K/V (constant across images and tiles)."""
        self._ready()
        y = encoder_hidden_states
        # Cache key: the tensor OBJECTS (held strongly, so their storage cannot be recycled for another prompt at the same
        # address) plus their version counters. Writes that bypass torch (raw pointers, ctypes) need invalidate_prompt().
        k = self._prompt_key
        if (k is not None and k[0] is y and k[1] == y._version and k[2] is encoder_attention_mask and
                k[3] == (None if encoder_attention_mask is None else encoder_attention_mask._version)):
            return
        key = (y, y._version, encoder_attention_mask, None if encoder_attention_mask is None else encoder_attention_mask._version)
        y = y.detach().to("cpu", torch.float32)
        y = y.reshape(-1, y.shape[-2], y.shape[-1])
        if y.shape[0] != 1:
            if not all(torch.equal(y[0], y[i]) for i in range(1, y.shape[0])):
                raise NotImplementedError("one shared prompt per batch (the CLI uses a single fixed prompt file, inference.py:256-259)")
        y = y[0].contiguous()
        n_tok = y.shape[0]
        if encoder_attention_mask is None:
            bias = torch.zeros(n_tok)
        else:
            m = encoder_attention_mask.detach().to("cpu", torch.float32)
            if m.ndim == 2:  # diffusers Transformer2DModel.forward: 2-D masks become (1 - m) * -10000
                bias = (1 - m[0]) * -10000.0
            else:            # ndim == 3 ([B,1,L], what the CLI passes): used as an additive bias as is
                bias = m.reshape(-1, m.shape[-1])[0]
        bias = bias.contiguous()
        self.ctx.check(self.ctx.lib.ir_dit_set_prompt(self.ctx.h, self.ctx.stream(), C.c_void_p(y.data_ptr()), C.c_void_p(bias.data_ptr()), n_tok),
                       "ir_dit_set_prompt")
        self._prompt_key = key

    def invalidate_prompt(self):
        """Forget the cached prompt: the next call projects encoder_hidden_states again even if it is the same tensor object."""
        self._prompt_key = None

    def _check_added_cond(self, added, h, w):
        """added_cond_kwargs as generate.py:56-62 builds them: None entries for the 512 model, the latent's (height, width) and height / width for the
        sample_size 128 one. The library derives exactly these from the latent it is given; anything else is a conditioning the path does not offer."""
        res = None if added is None else added.get("resolution")
        ar = None if added is None else added.get("aspect_ratio")
        if not self.cfg["micro"]:
            if res is not None or ar is not None:
                raise NotImplementedError("added_cond_kwargs given to a model without micro-conditioning (sample_size != 128)")
            return
        if res is None and ar is None:
            return   # the fused entry points (step / ir_pipeline) never pass them: the library computes them itself
        r = torch.as_tensor(res, dtype=torch.float32).reshape(-1, 2).cpu()
        a = torch.as_tensor(ar, dtype=torch.float32).reshape(-1).cpu()
        if not (bool((r == torch.tensor([float(h), float(w)])).all()) and bool(((a - float(h) / float(w)).abs() <= 1e-6).all())):
            raise NotImplementedError("micro-conditioning is computed from the latent's own height / width (generate.py:58-60); other values are not offered")

    @staticmethod
    def _scalar_timestep(timestep):
        t = torch.as_tensor(timestep).detach().reshape(-1).to("cpu", torch.float32)
        if t.numel() > 1 and not bool((t == t[0]).all()):
            raise NotImplementedError("one timestep per batch (generate.py:65 expands a single value)")
        return float(t[0])

    @torch.no_grad()
    def __call__(self, hidden_states, timestep=None, encoder_hidden_states=None, encoder_attention_mask=None, added_cond_kwargs=None,
                 return_dict=True, **unused):
        self._ready()
        self.set_prompt(encoder_hidden_states, encoder_attention_mask)
        x = hidden_states.to(self.device, torch.float32).contiguous()
        n, ch, h, w = x.shape
        if ch != 4 or h % 2 or w % 2:
            raise ValueError(f"latents must be [B,4,h,w] with even h,w, got {tuple(x.shape)}")
        self._check_added_cond(added_cond_kwargs, h, w)
        self.ensure_pos(h // 2, w // 2)
        out = torch.empty(n, 8, h, w, dtype=torch.float32, device=self.device)
        ws = self.ctx.workspace(self.ctx.ws_bytes(L.STAGE_DIT, n, h, w))
        self.ctx.check(self.ctx.lib.ir_dit_forward(self.ctx.h, self.ctx.stream(), L.ptr(x), self._scalar_timestep(timestep), L.ptr(out), n, h, w,
                                                    L.ptr(ws), ws.numel()), "ir_dit_forward")
        return SimpleNamespace(sample=out) if return_dict else (out,)

    forward = __call__

    @torch.no_grad()
    def step(self, latents, timestep, alpha_cumprod, encoder_hidden_states, encoder_attention_mask=None):
        """Fused generate_sample_1step (generate.py:22-51): returns x0 without materialising the 8-channel output."""
        self._ready()
        self.set_prompt(encoder_hidden_states, encoder_attention_mask)
        x = latents.to(self.device, torch.float32).contiguous()
        n, ch, h, w = x.shape
        self.ensure_pos(h // 2, w // 2)
        out = torch.empty_like(x)
        ws = self.ctx.workspace(self.ctx.ws_bytes(L.STAGE_DIT, n, h, w))
        self.ctx.check(self.ctx.lib.ir_dit_step(self.ctx.h, self.ctx.stream(), L.ptr(x), L.ptr(out), n, h, w, self._scalar_timestep(timestep),
                                                 float(alpha_cumprod), L.ptr(ws), ws.numel()), "ir_dit_step")
        return out


class ControlTransformerHalf(_DeviceModule):
    """ControlNet-Half over the DiT (diffusion/model/nets/transformer_controlnet.py:58-173): copies of the
    first copy_blocks_num blocks run on the patch-embedded condition latent `c`, and block i (1..copy_blocks_num) of the base model
    receives x + after_proj(copy i-1). Same constructor, state-dict layout (`base_model.*`, `controlnet.{i}.copied_block.*`,
    `controlnet.{i}.after_proj`, `controlnet.0.before_proj`) and call signature as the reference class; attributes it does not
    define fall through to the base model, as the reference's __getattr__ does (:78-84)."""

    FAMILY = "dit_ctrl"

    def __init__(self, base_model, copy_blocks_num=13):
        super().__init__()
        if not isinstance(base_model, Transformer2DModel):
            raise TypeError("base_model must be an instarevive_amd Transformer2DModel")
        if not 1 <= copy_blocks_num < base_model.cfg["num_layers"]:
            raise ValueError(f"copy_blocks_num must be in 1..{base_model.cfg['num_layers'] - 1}")
        if base_model.cfg.get("kv_compress") or base_model.cfg.get("qk_norm"):
            raise NotImplementedError("ControlTransformerHalf over a base model with KV compression / qk_norm (the copies would carry those branches too): not offered")
        self.base_model = base_model
        self.copy_blocks_num = copy_blocks_num
        self.total_blocks_num = base_model.cfg["num_layers"]
        if base_model._sd is not None:  # the copies start as clones of the base blocks, the projections as zeros (:23-39)
            C_ = base_model.cfg["num_attention_heads"] * base_model.cfg["attention_head_dim"]
            sd = {}
            for i in range(copy_blocks_num):
                for k in W._dit_block_keys(""):
                    sd[f"controlnet.{i}.copied_block.{k}"] = base_model._sd[f"transformer_blocks.{i}.{k}"].clone()
                for n_ in (("before_proj",) if i == 0 else ()) + ("after_proj",):
                    sd[f"controlnet.{i}.{n_}.weight"], sd[f"controlnet.{i}.{n_}.bias"] = torch.zeros(C_, C_), torch.zeros(C_)
            self._sd = sd
        if base_model.ctx is not None:
            self.to(base_model.device)

    def __getattr__(self, name):  # only reached for names not set on the wrapper
        if name in ("base_model", "_sd", "ctx"):
            raise AttributeError(name)
        return getattr(self.base_model, name)

    def _expected_keys(self):
        return W.dit_control_expected_keys(self.copy_blocks_num)

    def state_dict(self):
        sd = {"base_model." + k: v for k, v in self.base_model.state_dict().items()}
        sd.update(super().state_dict())
        return sd

    def load_state_dict(self, state_dict, strict=True):
        if not all(k.startswith(("base_model", "controlnet")) for k in state_dict):  # a bare base checkpoint (:154-166 of the in-tree twin)
            res = self.base_model.load_state_dict(state_dict, strict)
            if self.ctx is not None and self._sd is not None:
                self._upload()  # re-binding the base model dropped the device-side control branch
            return res
        base = {k[len("base_model."):]: v for k, v in state_dict.items() if k.startswith("base_model.")}
        ctrl = {k: v for k, v in state_dict.items() if k.startswith("controlnet.")}
        res = None
        if base or strict:
            res = self.base_model.load_state_dict(base, strict)
        res2 = self._check_keys(ctrl, strict)
        self._sd = {k: v.detach().cpu() for k, v in ctrl.items()}
        if self.ctx is not None:
            self._upload()
        return res2 if res is None else res

    def to(self, *args, **kwargs):
        self.base_model.to(*args, **kwargs)
        return super().to(*args, **kwargs)

    def _upload(self):
        if self.base_model.ctx is not self.ctx or self.base_model._sd is None:
            raise RuntimeError("the base model must be loaded and on the same device before the control branch is uploaded")
        self.ctx.upload_all(W.pack_dit_control(self._sd, self.copy_blocks_num))
        self.base_model._ready()           # the base model's weights are the ones bound on this GPU (re-uploads them if not)
        self.ctx.check(self.ctx.lib.ir_dit_control_configure(self.ctx.h, self.copy_blocks_num), "ir_dit_control_configure")
        self.base_model._prompt_key = None  # the control copies' prompt K/V caches are built by the next set_prompt
        self._mark_bound()

    def _ready(self):
        if self.ctx is None or self._sd is None:
            raise RuntimeError("ControlTransformerHalf: load_state_dict(...) and .to('cuda') must both be called before forward")
        self.base_model._ready()
        if not self._is_bound():
            self._upload()

    def _prep(self, hidden_states, c, encoder_hidden_states, encoder_attention_mask):
        if c is None:
            raise ValueError("ControlTransformerHalf needs the condition latent c (the reference dereferences it unconditionally, :44)")
        self._ready()
        self.base_model.set_prompt(encoder_hidden_states, encoder_attention_mask)
        x = hidden_states.to(self.device, torch.float32).contiguous()
        cc = c.to(self.device, torch.float32).contiguous()
        n, ch, h, w = x.shape
        if ch != 4 or h % 2 or w % 2 or cc.shape != x.shape:
            raise ValueError(f"latents and c must both be [B,4,h,w] with even h,w, got {tuple(x.shape)} and {tuple(cc.shape)}")
        self.base_model.ensure_pos(h // 2, w // 2)
        return x, cc, self.ctx.workspace(self.ctx.ws_bytes(L.STAGE_DIT, n, h, w))

    @torch.no_grad()
    def __call__(self, hidden_states, encoder_hidden_states=None, timestep=None, added_cond_kwargs=None, class_labels=None,
                 cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None, c=None, return_dict=True):
        x, cc, ws = self._prep(hidden_states, c, encoder_hidden_states, encoder_attention_mask)
        n, _, h, w = x.shape
        out = torch.empty(n, 8, h, w, dtype=torch.float32, device=self.device)
        self.ctx.check(self.ctx.lib.ir_dit_forward_control(self.ctx.h, self.ctx.stream(), L.ptr(x), L.ptr(cc), self._scalar_timestep(timestep),
                                                            L.ptr(out), n, h, w, L.ptr(ws), ws.numel()), "ir_dit_forward_control")
        return out if return_dict else (out,)  # the reference returns the bare tensor here (:169-172), not a .sample holder

    forward = __call__

    @torch.no_grad()
    def step(self, latents, timestep, alpha_cumprod, encoder_hidden_states, encoder_attention_mask=None, c=None):
        """Fused generate_sample_1step(..., c=c) (generate.py:22-51): x0 from the eps half, inside the HIP path."""
        x, cc, ws = self._prep(latents, c, encoder_hidden_states, encoder_attention_mask)
        n, _, h, w = x.shape
        out = torch.empty_like(x)
        self.ctx.check(self.ctx.lib.ir_dit_step_control(self.ctx.h, self.ctx.stream(), L.ptr(x), L.ptr(cc), L.ptr(out), n, h, w,
                                                         self._scalar_timestep(timestep), float(alpha_cumprod), L.ptr(ws), ws.numel()),
                       "ir_dit_step_control")
        return out


# ====================================================================================================== T5 encoder (prompt producer)
class T5EncoderModel(_DeviceModule):
    """transformers.T5EncoderModel as the reference's T5Embedder uses it (diffusion/model/t5.py:80,95-100): T5 v1.1 encoder
    (gated-GELU feed-forward, shared relative-position bias, no biases), state dict in the transformers key layout. The call returns
    a dict with 'last_hidden_state' [B, T, d_model] (fp32), as the reference indexes it."""
    FAMILY = "t5"

    def __init__(self, d_model=4096, d_kv=64, num_heads=64, d_ff=10240, num_layers=24, vocab_size=32128, relative_attention_num_buckets=32,
                 relative_attention_max_distance=128, feed_forward_proj="gated-gelu", layer_norm_epsilon=1e-6, **unused):
        super().__init__()
        if feed_forward_proj != "gated-gelu" or abs(layer_norm_epsilon - 1e-6) > 1e-12:
            raise NotImplementedError("the MI355X path implements the T5 v1.1 encoder (gated-gelu, eps 1e-6) of DeepFloyd/t5-v1_1-xxl")
        self.cfg = dict(d_model=d_model, d_kv=d_kv, num_heads=num_heads, d_ff=d_ff, num_layers=num_layers, vocab_size=vocab_size,
                        buckets=relative_attention_num_buckets, max_distance=relative_attention_max_distance)
        self.config = SimpleNamespace(**self.cfg)

    @classmethod
    def from_pretrained(cls, name_or_path, subfolder=None, **kw):
        folder = _resolve_pretrained(name_or_path, subfolder)
        with open(os.path.join(folder, "config.json")) as f:
            cfg = json.load(f)
        m = cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})
        m.load_state_dict(_load_weights_file(folder), strict=False)
        return m

    def _expected_keys(self):
        return W.t5_expected_keys(self.cfg)

    def load_state_dict(self, state_dict, strict=True):
        sd = dict(state_dict)
        if "shared.weight" not in sd and "encoder.embed_tokens.weight" in sd:
            sd["shared.weight"] = sd["encoder.embed_tokens.weight"]
        res = self._check_keys(sd, strict, ignore=("encoder.embed_tokens.weight", "decoder.", "lm_head."))
        self._sd = {k: v.detach().cpu() for k, v in sd.items() if k in set(self._expected_keys())}
        if self.ctx is not None:
            self._upload()
        return res

    def _upload(self):
        c = self.cfg
        self.ctx.upload_all(W.pack_t5(self._sd, c))
        self.ctx.check(self.ctx.lib.ir_t5_configure(self.ctx.h, c["num_layers"], c["d_model"], c["num_heads"], c["d_kv"], c["d_ff"], c["vocab_size"]),
                       "ir_t5_configure")
        self._bias_lengths = set()  # position-bias tables depend on the weights: rebuilt per sequence length after every upload
        self._mark_bound()

    @torch.no_grad()
    def __call__(self, input_ids=None, attention_mask=None, **unused):
        self._ready()
        ids = input_ids.to(self.device, torch.int32).contiguous()
        if ids.ndim != 2 or ids.shape[1] > 512:
            raise ValueError(f"input_ids must be [B, T] with T <= 512, got {tuple(ids.shape)}")
        b, t = ids.shape
        name = f"t5.bias.{t}"
        if t not in self._bias_lengths:
            self._bias_lengths.add(t)
            self.ctx.upload(name, W.t5_position_bias(self._sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], t,
                                                     self.cfg["buckets"], self.cfg["max_distance"]))
        mask = None if attention_mask is None else attention_mask.to(self.device, torch.float32).contiguous()
        if mask is not None and tuple(mask.shape) != (b, t):
            raise ValueError("attention_mask must have the shape of input_ids")
        out = torch.empty(b, t, self.cfg["d_model"], dtype=torch.float32, device=self.device)
        ws = self.ctx.workspace(self.ctx.ws_bytes(L.STAGE_T5, b, t, 0))
        self.ctx.check(self.ctx.lib.ir_t5_encode(self.ctx.h, self.ctx.stream(), L.ptr(ids), L.ptr(mask) if mask is not None else None, L.ptr(out), b, t,
                                                  L.ptr(ws), ws.numel()), "ir_t5_encode")
        return {"last_hidden_state": out}

    forward = __call__


class T5Embedder:
    """diffusion/model/t5.py:13-101 with the same constructor keywords that matter here and the same get_text_embeddings(texts)
    -> (embeddings [B, L, 4096], attention_mask [B, L]). The tokenizer is transformers' own (AutoTokenizer over the model folder, as in
    the reference); the encoder is the HIP T5EncoderModel above. Caption cleaning (clean_caption twice under use_text_preprocessing,
    t5.py:106-233) is instarevive_amd.captions, pinned by tests/golden/captions.json."""

    def __init__(self, device, dir_or_name="t5-v1_1-xxl", *, tokenizer=None, model=None, model_max_length=120, use_text_preprocessing=True,
                 **unused):
        self.device = torch.device(device)
        self.model_max_length = model_max_length
        self.use_text_preprocessing = use_text_preprocessing
        if tokenizer is None:
            from transformers import AutoTokenizer  # host-side text -> ids only; needs the tokenizer files in dir_or_name
            tokenizer = AutoTokenizer.from_pretrained(dir_or_name)
        self.tokenizer = tokenizer
        self.model = (model or T5EncoderModel.from_pretrained(dir_or_name)).to(self.device)

    def text_preprocessing(self, text):
        from .captions import text_preprocessing
        return text_preprocessing(text, self.use_text_preprocessing)

    def get_text_embeddings(self, texts):
        texts = [self.text_preprocessing(t) for t in texts]
        tok = self.tokenizer(texts, max_length=self.model_max_length, padding="max_length", truncation=True, return_attention_mask=True,
                             add_special_tokens=True, return_tensors="pt")
        mask = tok["attention_mask"].to(self.device)
        embs = self.model(input_ids=tok["input_ids"].to(self.device), attention_mask=mask)["last_hidden_state"].detach()
        return embs, mask


# ====================================================================================================== scheduler
class DDPMScheduler:
    """Only what the path consumes: alphas_cumprod (generate.py:45). Linear / scaled_linear betas like diffusers."""

    def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=2e-2, beta_schedule="linear", **unused):
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule)

    @classmethod
    def from_pretrained(cls, name_or_path, subfolder=None, **kw):
        try:
            folder = _resolve_pretrained(name_or_path, subfolder)
            with open(os.path.join(folder, "scheduler_config.json")) as f:
                cfg = json.load(f)
            return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})
        except FileNotFoundError:
            return cls()  # PixArt-alpha's scheduler: linear betas 1e-4..2e-2, T=1000 (diffusion/model/gaussian_diffusion.py:107-116)
