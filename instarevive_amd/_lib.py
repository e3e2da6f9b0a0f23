"""ctypes binding of csrc/libinstarevive_hip.so (the C ABI declared in include/instarevive_hip.h).

There is deliberately no fallback: if the shared library is missing or a call fails, an exception is raised.
PyTorch is used only as the owner of device memory / streams; tensors cross the ABI as raw device pointers.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# INSTAREVIVE_HIP_LIB points the loader at another build of the same library (A/B timing of kernel changes)
LIB_PATH = os.environ.get("INSTAREVIVE_HIP_LIB") or os.path.join(_HERE, "csrc", "libinstarevive_hip.so")

# every symbol include/instarevive_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "ir_abi_version", "ir_init", "ir_destroy", "ir_last_error", "ir_upload", "ir_drop_optional", "ir_has_tensor",
    "ir_swinir_configure", "ir_vae_configure", "ir_dit_configure", "ir_dit_control_configure", "ir_dit_set_prompt", "ir_t5_configure", "ir_t5_encode", "ir_workspace_bytes",
    "ir_swinir_forward", "ir_vae_encode", "ir_dit_forward", "ir_dit_step", "ir_dit_forward_control", "ir_dit_step_control", "ir_vae_decode", "ir_color_fix",
    "ir_pipeline", "ir_u8_to_nchw", "ir_nchw_to_u8", "ir_profile_begin", "ir_profile_select", "ir_profile_end", "ir_profile_end_kernels", "ir_profile_kernel_count",
    "ir_profile_kernel_name",
    "ir_op_conv", "ir_op_conv_splitk", "ir_op_conv_groupnorm", "ir_op_linear", "ir_op_groupnorm", "ir_op_layernorm", "ir_op_attention", "ir_op_swin_attention",
    "ir_op_softmax_rows", "ir_op_nchw_to_nhwc", "ir_op_nhwc_to_nchw",
    "ir_tiled_count", "ir_tiled_encode", "ir_tiled_encode_part", "ir_tiled_encode_overflow", "ir_op_conv_up2x2", "ir_op_conv_norm", "ir_op_vae_conv_in", "ir_op_vae_norm_conv_out", "ir_op_conv64", "ir_op_conv64_to3", "ir_tiled_dit", "ir_tiled_blend_latent", "ir_tiled_decode", "ir_tiled_blend_pixels", "ir_set_plain_kernels", "ir_set_fp8", "ir_set_fp8_mask", "ir_attn_fallback_count", "ir_op_conv_fp8", "ir_op_conv_fp8_up", "ir_op_conv_fp8_route", "ir_fp8_features", "ir_op_attention_fp8", "ir_op_attention_d512_fp8",
    "ir_unet_configure", "ir_unet_set_context", "ir_cldm_sample", "ir_cldm_pipeline", "ir_clip_text_configure", "ir_clip_text_encode", "ir_op_groupnorm_any", "ir_op_geglu",
]

STAGE_SWINIR, STAGE_VAE_ENCODE, STAGE_DIT, STAGE_VAE_DECODE, STAGE_PIPELINE, STAGE_COLORFIX, STAGE_T5, STAGE_CLDM, STAGE_CLDM_PIPELINE, STAGE_CLIP_TEXT = range(10)
FLAG_NO_PREPROCESS, FLAG_TILED, FLAG_FIX_WAVELET, FLAG_FIX_ADAIN, FLAG_CONTROL_LQ, FLAG_GRAPH, FLAG_FP8 = 1, 2, 4, 8, 16, 32, 64
# ir_set_fp8_mask (include/instarevive_hip.h): QUALIFIED = the set qualified against the fp32 oracle on flat-softmax weights (>= 46.3 dB at 2048 x
# 2048: the three attention parts + decoder level-0 / level-2 ResnetBlock convs); DEFAULT (the context's, ABI v3) = the same without the DiT
# self-attention, whose cost explodes on peaky rows; fp8_select.auto_mask() calibrates the set on the loaded weights; ALL = every part, outside
# north_star's 0.1 dB above a 25.8 dB reference
FP8_MASK_DEFAULT, FP8_MASK_QUALIFIED, FP8_MASK_ALL, FP8_MASK_ATTENTION = 0x5006, 0x5007, 0xffffffff, 0b111
FP8_CONV_BITS = sum(1 << b for b in (4, 5, 6, 7, 8, 12, 13, 14, 15, 16))
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_LRELU, ACT_SILU = range(5)

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load_library():
    """Load the HIP library; raises NativeLibraryError when it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -m instarevive_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the MI355X path.")
    lib = C.CDLL(LIB_PATH)
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.ir_abi_version.restype = i
    lib.ir_init.argtypes = [i, C.POINTER(vp)]
    lib.ir_destroy.argtypes = [vp]
    lib.ir_destroy.restype = None
    lib.ir_last_error.argtypes = [vp]
    lib.ir_last_error.restype = C.c_char_p
    lib.ir_upload.argtypes = [vp, C.c_char_p, vp, sz]
    lib.ir_drop_optional.argtypes = [vp, C.c_char_p]
    lib.ir_has_tensor.argtypes = [vp, C.c_char_p]
    lib.ir_swinir_configure.argtypes = [vp, i, i, C.POINTER(i), i, i, i, f, C.POINTER(f)]
    lib.ir_vae_configure.argtypes = [vp, i, i, C.POINTER(i), i, i, i]
    lib.ir_dit_configure.argtypes = [vp, i, i, i, i, i, i]
    lib.ir_dit_control_configure.argtypes = [vp, i]
    lib.ir_dit_set_prompt.argtypes = [vp, vp, vp, vp, i]
    lib.ir_t5_configure.argtypes = [vp, i, i, i, i, i, i]
    lib.ir_t5_encode.argtypes = [vp, vp, vp, vp, vp, i, i, vp, sz]
    lib.ir_workspace_bytes.argtypes = [vp, i, i, i, i, i, i, i]
    lib.ir_workspace_bytes.restype = sz
    lib.ir_swinir_forward.argtypes = [vp, vp, vp, vp, i, i, i, vp, sz]
    lib.ir_vae_encode.argtypes = [vp, vp, vp, vp, i, i, i, vp, sz]
    lib.ir_dit_forward.argtypes = [vp, vp, vp, f, vp, i, i, i, vp, sz]
    lib.ir_dit_step.argtypes = [vp, vp, vp, vp, i, i, i, f, f, vp, sz]
    lib.ir_dit_forward_control.argtypes = [vp, vp, vp, vp, f, vp, i, i, i, vp, sz]
    lib.ir_dit_step_control.argtypes = [vp, vp, vp, vp, vp, i, i, i, f, f, vp, sz]
    lib.ir_vae_decode.argtypes = [vp, vp, vp, vp, i, i, i, vp, sz]
    lib.ir_color_fix.argtypes = [vp, vp, i, vp, vp, vp, i, i, i, vp, sz]
    lib.ir_pipeline.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, f, f, f, vp, sz]
    lib.ir_profile_begin.argtypes = [vp]
    lib.ir_profile_select.argtypes = [vp, i]
    lib.ir_profile_end.argtypes = [vp, vp, i, vp, vp, vp, vp]
    lib.ir_profile_end_kernels.argtypes = [vp, vp, i, vp, vp, vp, vp]
    lib.ir_profile_kernel_count.argtypes = []
    lib.ir_profile_kernel_name.argtypes = [i]
    lib.ir_profile_kernel_name.restype = C.c_char_p
    lib.ir_u8_to_nchw.argtypes = [vp, vp, vp, vp, i, i, i]
    lib.ir_nchw_to_u8.argtypes = [vp, vp, vp, vp, i, i, i]
    lib.ir_op_conv.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, i, i, f, vp, i, i]
    lib.ir_op_conv_splitk.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, vp, i, i, vp, sz, C.POINTER(C.c_int)]
    lib.ir_op_conv_groupnorm.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, vp, i, vp, sz, C.POINTER(C.c_int)]
    lib.ir_op_linear.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp, vp, i, i, f]
    lib.ir_op_groupnorm.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, f, i, vp, sz]
    lib.ir_op_layernorm.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, f]
    lib.ir_op_attention.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, f, vp, vp, sz]
    lib.ir_op_swin_attention.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, f]
    lib.ir_op_softmax_rows.argtypes = [vp, vp, vp, vp, i, i]
    lib.ir_op_nchw_to_nhwc.argtypes = [vp, vp, vp, vp, i, i, C.c_long, i, f, f]
    lib.ir_op_nhwc_to_nchw.argtypes = [vp, vp, vp, i, vp, i, i, C.c_long, f, f, i]
    lib.ir_tiled_count.argtypes = [i, i, i, i]
    lib.ir_tiled_encode.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, f, vp, sz]
    lib.ir_tiled_encode_part.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, f, i, i, i, vp, vp, vp, sz]
    lib.ir_tiled_encode_overflow.argtypes = [vp, vp]
    lib.ir_op_conv_up2x2.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i]
    lib.ir_op_conv_norm.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i]
    lib.ir_op_vae_conv_in.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, f, f, vp]
    lib.ir_op_vae_norm_conv_out.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i]
    lib.ir_op_conv64.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, f]
    lib.ir_op_conv64_to3.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i]
    lib.ir_tiled_dit.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, i, f, f, i, vp, sz]
    lib.ir_tiled_blend_latent.argtypes = [vp, vp, vp, vp, i, i, i, i, i]
    lib.ir_tiled_decode.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, f, vp, sz]
    lib.ir_tiled_blend_pixels.argtypes = [vp, vp, vp, vp, i, i, i, i, i, vp, sz]
    lib.ir_set_plain_kernels.argtypes = [vp, i]
    lib.ir_set_fp8.argtypes = [vp, i]
    lib.ir_set_fp8_mask.argtypes = [vp, C.c_uint]
    lib.ir_attn_fallback_count.argtypes = [vp, vp, C.c_int]
    lib.ir_fp8_features.argtypes = []
    lib.ir_op_conv_fp8_route.argtypes = [vp, i, i, i, i, i, i]
    lib.ir_op_attention_fp8.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, f, vp, sz]
    lib.ir_op_conv_fp8.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp]
    lib.ir_op_conv_fp8_up.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i]
    lib.ir_op_attention_d512_fp8.argtypes = [vp, vp, vp, vp, vp, vp, i, i, f, vp, sz]
    lib.ir_unet_configure.argtypes = [vp, i, i, i, C.POINTER(i), i, i, i, i, i]
    lib.ir_unet_set_context.argtypes = [vp, vp, vp, i]
    lib.ir_cldm_sample.argtypes = [vp, vp, vp, vp, vp, i, i, i, f, i, vp, sz]
    lib.ir_clip_text_configure.argtypes = [vp, i, i, i, i, i, i]
    lib.ir_clip_text_encode.argtypes = [vp, vp, vp, vp, i, vp, sz]
    lib.ir_cldm_pipeline.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, f, f, vp, sz]
    lib.ir_op_groupnorm_any.argtypes = [vp, vp, vp, vp, vp, vp, i, C.c_long, i, i, f, i, vp, sz]
    lib.ir_op_geglu.argtypes = [vp, vp, vp, vp, C.c_long, i]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ("ir_abi_version",):
            pass
    _lib = lib
    return lib


def ptr(t):
    """Device (or host) pointer of a contiguous tensor, or None."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensors crossing the C ABI must be contiguous"
    return C.c_void_p(t.data_ptr())


def bf16_bits(t: torch.Tensor) -> torch.Tensor:
    """fp32 tensor -> raw bf16 bit pattern (round-to-nearest-even) as int16 view, same device."""
    return t.to(torch.bfloat16).contiguous().view(torch.int16)


def from_bf16_bits(t: torch.Tensor) -> torch.Tensor:
    return t.view(torch.bfloat16).to(torch.float32)


class Context:
    """One ir_ctx on one GPU. Owns uploaded weights; not thread safe (mirrors the reference's single host loop)."""

    def __init__(self, device=0):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise NativeLibraryError("no GPU visible: the MI355X path cannot run (and has no CPU fallback)")
        self.device = torch.device("cuda", device if isinstance(device, int) else (device.index or 0))
        h = C.c_void_p()
        rc = self.lib.ir_init(self.device.index, C.byref(h))
        if rc != 0:
            raise NativeLibraryError(f"ir_init failed ({rc})")
        self.h = h
        self._ws = None
        self._names = set()

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ir_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- helpers
    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.ir_last_error(self.h)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def upload(self, name, t: torch.Tensor):
        """Copy a HOST tensor (any dtype, contiguous) into the named device buffer of the context."""
        t = t.detach().cpu().contiguous()
        self.check(self.lib.ir_upload(self.h, name.encode(), C.c_void_p(t.data_ptr()), t.numel() * t.element_size()),
                   f"ir_upload({name})")
        self._names.add(name)

    def upload_all(self, tensors: dict):
        """Upload one model's packed tensors. The optional conv forms of the families involved (first name component: "vae", "swin", ...) are
        dropped first: the table is keyed by name, and a form the previous model of that family had but this one has not would otherwise be bound."""
        for fam in sorted({k.split(".", 1)[0] for k in tensors}):
            if self.lib.ir_drop_optional(self.h, fam.encode()) < 0:
                self.check(-1, f"ir_drop_optional({fam})")
        for k, v in tensors.items():
            self.upload(k, v)

    def has(self, name):
        return bool(self.lib.ir_has_tensor(self.h, name.encode()))

    def workspace(self, nbytes):
        """Grow-only scratch buffer owned by torch's allocator; handed to the C ABI as caller-owned workspace."""
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    PROFILE_CLASSES = ["conv3x3", "linear", "flash_attn", "swin_attn", "groupnorm", "layernorm", "softmax_rows", "transpose", "other"]

    def profile_begin(self, only=None):
        """only: a "class/kernel" name of profile_end_kernels() - events around that kernel's launches alone (ir_profile_select)."""
        kid = -1
        if only is not None:
            names = [self.lib.ir_profile_kernel_name(i).decode() for i in range(int(self.lib.ir_profile_kernel_count()))]
            kid = names.index(only)
        self.check(self.lib.ir_profile_select(self.h, kid), "ir_profile_select")
        self.check(self.lib.ir_profile_begin(self.h), "ir_profile_begin")

    def profile_end(self):
        """-> {class: dict(ms, flops, bytes, launches)} summed over every launch since profile_begin()."""
        n = len(self.PROFILE_CLASSES)
        ms, fl, by, la = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_longlong * n)()
        self.check(self.lib.ir_profile_end(self.h, self.stream(), n, ms, fl, by, la), "ir_profile_end")
        return {k: dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=int(la[i])) for i, k in enumerate(self.PROFILE_CLASSES)}

    def profile_end_kernels(self):
        """-> {"class/kernel": dict(ms, flops, bytes, launches)} of the same measurement, one row per kernel that ran (algorithmic FLOPs / bytes)."""
        n = int(self.lib.ir_profile_kernel_count())
        ms, fl, by, la = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_longlong * n)()
        self.check(self.lib.ir_profile_end_kernels(self.h, self.stream(), n, ms, fl, by, la), "ir_profile_end_kernels")
        return {self.lib.ir_profile_kernel_name(i).decode(): dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=int(la[i])) for i in range(n) if la[i]}

    def ws_bytes(self, stage, n, h, w, flags=0, tile_size=0, tile_stride=0):
        return int(self.lib.ir_workspace_bytes(self.h, stage, n, h, w, flags, tile_size, tile_stride))
