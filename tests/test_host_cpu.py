"""CPU-only checks: the C-ABI library builds/loads and exports every symbol include/instarevive_hip.h declares (no compute
calls), weight packing layouts, host glue (_sliding_windows, loaders, CLI parsing), failure behaviour without a GPU."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from instarevive_amd import _lib
    from instarevive_amd.build import build
    build()
    lib = _lib.load_library()
    header = open(os.path.join(ROOT, "include", "instarevive_hip.h")).read()
    declared = set(re.findall(r"\b(ir_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.ir_abi_version() == 1


def test_no_cpu_fallback():
    from instarevive_amd import models
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        models.get_context(torch.device("cuda", 0))
    with pytest.raises(Exception):
        models.get_context(torch.device("cpu"))
    m = models.AutoencoderKL(block_out_channels=(32, 64, 128, 128))
    with pytest.raises(RuntimeError):
        m.encode(torch.zeros(1, 3, 64, 64))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "instarevive_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
    assert not re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(ROOT, "inference.py")).read(), re.M)


def test_pack_conv_and_linear_layouts():
    from instarevive_amd import weights as W
    w = torch.arange(2 * 3 * 9, dtype=torch.float32).reshape(2, 3, 3, 3)
    p = W.pack_conv3x3(w, 8, 32).view(torch.bfloat16).float().reshape(32, 9, 8)
    assert torch.equal(p[1, 4, :3], w[1, :, 1, 1]) and p[2:].abs().sum() == 0 and p[:, :, 3:].abs().sum() == 0
    lw = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    q = W.pack_linear(lw, 4, 8, row_map=torch.tensor([3, 0]), col_map=torch.tensor([1, 2, 5])).view(torch.bfloat16).float()
    assert q[3, 1] == 0 and q[3, 2] == 1 and q[3, 5] == 2 and q[0, 1] == 3 and q.abs().sum() == lw.sum()


def test_swin_packing_head_padding():
    from instarevive_amd import weights as W
    cfg = dict(embed_dim=60, depths=[1], num_heads=[6], window_size=8, mlp_ratio=2, img_range=1.0)
    sd = {k: torch.randn(*s) for k, s in W.swinir_shapes(cfg).items()}
    p = W.pack_swinir(sd, cfg)
    qkv = p["swin.l0.b0.qkv.w"].view(torch.bfloat16).float()  # [3*192][192]
    assert qkv.shape == (576, 192)
    # row for (k, head 2, d 7) comes from original row 60 + 2*10 + 7
    torch.testing.assert_close(qkv[192 + 2 * 32 + 7, :60], sd["layers.0.residual_group.blocks.0.attn.qkv.weight"][60 + 27].bfloat16().float())
    assert qkv[192 + 2 * 32 + 10: 192 + 3 * 32].abs().sum() == 0  # padded head dims are exact zeros
    assert p["swin.l0.b0.biasT"].shape == (6, 64, 64)
    assert set(W.swinir_expected_keys(cfg)) >= set(sd)


def test_sliding_windows_and_loaders():
    from instarevive_amd.pipeline import _sliding_windows
    from instarevive_amd import utils
    assert _sliding_windows(64, 64, 64, 56) == [(0, 64, 0, 64)]
    assert len(_sliding_windows(272, 480, 64, 56)) == 45
    assert _sliding_windows(70, 64, 64, 56) == [(0, 64, 0, 64), (6, 70, 0, 64)]
    m = utils.instantiate_from_config({"target": "diffusion.model.swinir.SwinIR", "params": dict(
        img_size=64, patch_size=1, in_chans=3, embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8, mlp_ratio=2, sf=8, img_range=1.0,
        upsampler="nearest+conv", resi_connection="1conv", unshuffle=True, unshuffle_scale=8)})
    from instarevive_amd.models import SwinIR
    assert isinstance(m, SwinIR) and len(m.state_dict()) == 732  # same key count as the reference checkpoint
    with pytest.raises(RuntimeError):
        m.load_state_dict({"bogus": torch.zeros(1)}, strict=True)
    img = np.zeros((50, 37, 3), np.uint8)
    assert utils.pad(img, 64).shape == (64, 64, 3)


def test_cli_flag_surface():
    sys.argv = ["inference.py", "--ckpt", "x.ckpt", "--input", "in", "--output", "out", "--tiled", "--sr_scale", "4", "--color_fix_type", "adain",
                "--use_guidance", "--g_scale", "1.0", "--show_lq", "--skip_if_exist", "--use_center_crop", "--repeat_times", "2"]
    sys.path.insert(0, ROOT)
    import importlib
    inf = importlib.import_module("inference")
    a = inf.parse_args()
    assert a.tiled and a.tile_size == 512 and a.tile_stride == 448 and a.sr_scale == 4 and a.color_fix_type == "adain" and a.seed == 231
    assert a.device == "cuda" and a.repeat_times == 2


def test_scheduler_constant():
    from instarevive_amd.models import DDPMScheduler
    assert abs(float(DDPMScheduler().alphas_cumprod[400]) - 0.193572) < 1e-6
