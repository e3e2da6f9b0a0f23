"""CPU-only checks: the C-ABI library builds/loads and exports every symbol include/instarevive_hip.h declares (no compute
calls), weight packing layouts, host glue (_sliding_windows, loaders, CLI parsing), failure behaviour without a GPU."""
import json
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
    assert lib.ir_abi_version() == 3


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


def test_swin_masked_bias_classes():
    """weights.swin_masked_bias against calculate_mask of the reference (swinir.py:227-248) restated on a 3 x 4 window grid: every window's mask must be
    the table of its class minus the plain table, class = 2 * (last window row) + (last window column)."""
    import math
    from instarevive_amd import weights as W
    ws, shift, nh, nw = 8, 4, 3, 4
    H, Wd = nh * ws, nw * ws
    img = torch.zeros(H, Wd)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.view(nh, ws, nw, ws).permute(0, 2, 1, 3).reshape(nh * nw, ws * ws)          # window_partition
    ref = mw[:, None, :] - mw[:, :, None]
    ref = torch.where(ref != 0, torch.tensor(-100.0), torch.tensor(0.0))                  # [window][query][key]
    base = torch.randn(6, 64, 64)
    t = W.swin_masked_bias(base, ws, shift)
    assert t.shape == (4, 6, 64, 64) and torch.equal(t[0], base)
    for wy in range(nh):
        for wx in range(nw):
            cls = 2 * (wy == nh - 1) + (wx == nw - 1)
            got = (t[cls] - base)[0] / math.log2(math.e)                                  # [key][query]
            torch.testing.assert_close(got, ref[wy * nw + wx].t(), atol=1e-4, rtol=0)
    cfg = dict(embed_dim=60, depths=[2], num_heads=[6], window_size=8, mlp_ratio=2, img_range=1.0)
    p = W.pack_swinir({k: torch.randn(*sh) for k, sh in W.swinir_shapes(cfg).items()}, cfg)
    assert "swin.l0.b0.biasM" not in p and p["swin.l0.b1.biasM"].shape == (4, 6, 64, 64)


def test_swin_qkv_ring_tiles():
    """pack_swin_qkv_tiles: slot t // 2, tile t % 2 (13312 B apart), row r (400 B apart), k position p holds the padded qkv weight
    [32 t + r][_acc_order(192)[p]] - the layout swin_mlp_kernel<true, true> streams through its LDS ring; only blocks 1.. of an RSTB get one."""
    from instarevive_amd import weights as W
    cfg = dict(embed_dim=60, depths=[2], num_heads=[6], window_size=8, mlp_ratio=2, img_range=1.0)
    sd = {k: torch.randn(*s) for k, s in W.swinir_shapes(cfg).items()}
    p = W.pack_swinir(sd, cfg)
    assert "swin.l0.b0.qkv_t" not in p
    t, w = p["swin.l0.b1.qkv_t"], p["swin.l0.b1.qkv.w"]
    assert t.dtype == torch.uint8 and t.shape == (9, 28672)
    order = W._acc_order(192)
    for tile, r, pos in ((0, 0, 0), (5, 17, 9), (17, 31, 191), (8, 3, 100)):
        off = (tile % 2) * 13312 + r * 400 + pos * 2
        got = t[tile // 2, off:off + 2].view(torch.int16)[0]
        assert got == w[32 * tile + r, order[pos]]
    assert t[:, 26624:].abs().sum() == 0   # the slot's tail (the MLP's W2 area) is unused


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


def test_list_image_files_and_name_parts(tmp_path):
    from instarevive_amd import utils
    for rel in ("b.PNG", "a.jpg", "notes.txt", "sub/c.jpeg", "sub/deep/d.arw", "sub/e.bmp"):
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(b"x")
    got = utils.list_image_files(str(tmp_path))
    walk = [os.path.join(d, f) for d, _, fs in os.walk(str(tmp_path)) for f in fs if os.path.splitext(f)[1].lower() in (".jpg", ".png", ".jpeg", ".arw")]
    assert got == walk and len(got) == 4                      # os.walk order (the reference does not sort), lower-cased extension filter
    assert utils.list_image_files(str(tmp_path), max_size=2) == walk[:2] and utils.list_image_files(str(tmp_path), max_size=0) == []
    assert utils.list_image_files(str(tmp_path), exts=(".bmp",)) == [os.path.join(str(tmp_path), "sub", "e.bmp")]
    assert utils.get_file_name_parts("out/x/name.tar.png") == ("out/x", "name.tar", ".png")
    assert utils.get_file_name_parts("plain") == ("", "plain", "")


def test_load_state_dict_prefix_handling():
    from instarevive_amd import utils

    class Fake:
        def __init__(self, keys):
            self.keys, self.loaded, self.strict = keys, None, None

        def state_dict(self):
            return {k: None for k in self.keys}

        def load_state_dict(self, sd, strict=False):
            self.loaded, self.strict = dict(sd), strict

    w = {"a.weight": 1, "b.bias": 2}
    m = Fake(["a.weight", "b.bias"])
    utils.load_state_dict(m, {"state_dict": {"module." + k: v for k, v in w.items()}}, strict=True)   # wrapped + DDP prefix -> stripped
    assert m.loaded == w and m.strict is True
    m = Fake(["module.a.weight", "module.b.bias"])
    utils.load_state_dict(m, w)                                                                       # bare keys -> prefix added
    assert m.loaded == {"module." + k: v for k, v in w.items()} and m.strict is False
    m = Fake(["a.weight", "b.bias"])
    utils.load_state_dict(m, w)
    assert m.loaded == w
    with pytest.raises(KeyError):
        utils.instantiate_from_config({"params": {}})


def test_cli_job_pre_and_post_processing(tmp_path):
    """read_job / write_job / batches_of of inference.py around a stand-in for process(): the reference's per-file arithmetic
    (test_scripts/inference.py:263-291, 323-346) without a GPU."""
    from argparse import Namespace
    from PIL import Image
    sys.path.insert(0, ROOT)
    import importlib
    inf = importlib.import_module("inference")
    from tests.golden._det import det_input
    src = tmp_path / "in" / "sub"
    src.mkdir(parents=True)
    img = (det_input(3, (40, 56, 3)) * 255).numpy().astype(np.uint8)
    Image.fromarray(img).save(src / "x.png")
    base = dict(input=str(tmp_path / "in"), output=str(tmp_path / "out"), sr_scale=1, tiled=False, tile_size=512, use_center_crop=False,
                show_lq=False, disable_preprocess_model=False)
    # default: short edge 40 -> 512 (bicubic, ceil), pad to multiples of 64
    a = Namespace(**base)
    job = inf.read_job(str(src / "x.png"), 0, a)
    assert job.save_path == os.path.join(a.output, "sub", "x_0.png") and job.lq.size == (56, 40)
    assert job.valid_hw == (512, 717) and job.net_in.shape == (512, 768, 3) and job.net_in[:, 717:].max() == 0
    inf.write_job(job, job.net_in, job.net_in, a)                       # the "prediction" is the network input itself
    saved = np.array(Image.open(job.save_path))
    want = np.array(Image.fromarray(job.net_in[:512, :717]).resize((56, 40), Image.LANCZOS))
    assert np.array_equal(saved, want)
    # --sr_scale 2 --tiled --tile_size 64 --show_lq: LQ | stage-1 | result strip at the up-scaled LQ size
    b = Namespace(**dict(base, sr_scale=2.0, tiled=True, tile_size=64, show_lq=True))
    job = inf.read_job(str(src / "x.png"), 1, b)
    assert job.lq.size == (112, 80) and job.valid_hw == (80, 112) and job.net_in.shape == (128, 128, 3) and job.save_path.endswith("x_1.png")
    inf.write_job(job, job.net_in, 255 - job.net_in, b)
    strip = np.array(Image.open(job.save_path))
    assert strip.shape == (80, 336, 3) and np.array_equal(strip[:, :112], np.array(job.lq)) and np.array_equal(strip[:, 224:], np.array(job.lq))
    # --use_center_crop: 512 x 512 crop, nothing removed or resized afterwards
    c = Namespace(**dict(base, use_center_crop=True))
    job = inf.read_job(str(src / "x.png"), 0, c)
    assert job.valid_hw == () and job.net_in.shape == (512, 512, 3)
    inf.write_job(job, job.net_in, None, c)
    assert np.array_equal(np.array(Image.open(job.save_path)), job.net_in)
    # batching: consecutive equal shapes only, at most `limit`
    mk = lambda h, w: inf.Job("p", None, np.zeros((h, w, 3), np.uint8), ())
    groups = list(inf.batches_of([mk(64, 64), mk(64, 64), mk(64, 64), mk(64, 128), mk(64, 64)], 2))
    assert [len(g) for g in groups] == [2, 1, 1, 1]
    assert [len(g) for g in inf.batches_of([mk(64, 64)] * 3, 1)] == [1, 1, 1]


def test_clean_caption_matches_the_reference_fixture():
    """instarevive_amd.captions against tests/golden/captions.json = outputs of the reference's own T5Embedder.clean_caption /
    text_preprocessing (diffusion/model/t5.py:106-233) on captions covering every rewrite (URLs, tags, entities, CJK blocks, dashes,
    quotes, ids, file names, boilerplate, punctuation runs, quote stripping), one pass and the two passes the reference applies."""
    import json
    from instarevive_amd.captions import clean_caption, text_preprocessing
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "captions.json"), encoding="utf-8"))
    assert len(fx["captions"]) >= 15
    for c, once, twice, plain in zip(fx["captions"], fx["once"], fx["twice"], fx["plain"]):
        assert clean_caption(c) == once, (c, clean_caption(c), once)
        assert clean_caption(c, fix=lambda t: t) == once            # the fixture's ftfy stand-in is the identity: these captions do not need repair
        assert text_preprocessing(c) == twice, (c, text_preprocessing(c), twice)
        assert text_preprocessing(c, use_text_preprocessing=False) == plain


# ---------------------------------------------------------------------------------------------- ControlLDM (N4) host logic
def test_unet_layout_keys_and_packing():
    """weights.unet_expected_keys == the parameter names of the reference's UNetModel / ControlNet (via oracle.cldm.state_dict_shapes, which
    the fixture generator loads into the reference modules with strict=True); pack_unet emits the tensors ir_unet_configure binds."""
    import torch
    from instarevive_amd import weights as W
    from oracle import cldm as ocldm
    from tests.golden._det import det_state_dict
    for cfg, n_u, n_c in ((dict(ocldm.DEFAULT_CFG), 686, 324), (dict(ocldm.DEFAULT_CFG, model_channels=64, num_head_channels=32, context_dim=64), 686, 324)):
        for control, n in ((False, n_u), (True, n_c)):
            shapes = ocldm.state_dict_shapes(cfg, control)
            assert set(W.unet_expected_keys(cfg, control)) == set(shapes) and len(shapes) == n
    cfg = dict(ocldm.DEFAULT_CFG, model_channels=64, num_head_channels=32, context_dim=64)
    inb, mid, outb, skips = W.unet_layout(cfg)
    assert len(inb) == 12 and len(outb) == 12 and len(skips) == 12 and mid == 256
    assert [L[0][1] for L in outb] == [512, 512, 512, 512, 512, 384, 384, 256, 192, 192, 128, 128]   # h + skip channels of each decoder block
    assert [len(L) for L in outb] == [1, 1, 2, 2, 2, 3, 2, 2, 3, 2, 2, 2]                              # res [+ xf] [+ up]
    sd = det_state_dict(ocldm.state_dict_shapes(cfg), seed=5)
    p = W.pack_unet(sd, cfg)
    assert p["unet.in0.conv.w"].shape == (64, 9 * 32) and p["unet.out.conv.w"].shape == (32, 9 * 64)
    assert p["unet.in1.res.emb.w"].shape == (64, 256) and p["unet.in1.xf.qkv.w"].shape == (192, 64) and p["unet.in1.xf.ff1.w"].shape == (512, 64)
    assert p["unet.in1.xf.ckv.w"].shape == (128, 64) and p["unet.out2.up.w"].shape == (256, 9 * 256) and p["unet.in3.down.w"].shape == (64, 9 * 64)
    torch.testing.assert_close(p["unet.in1.res.emb.b"], sd["input_blocks.1.0.emb_layers.1.bias"] + sd["input_blocks.1.0.in_layers.2.bias"])
    assert float(p["unet.in1.xf.qkv.b"].abs().max()) == 0.0   # to_q / to_k / to_v have no bias
    pc = W.pack_unet(det_state_dict(ocldm.state_dict_shapes(cfg, True), seed=6), cfg, control=True)
    assert pc["cnet.in0.conv.w"].shape == (64, 9 * 32) and "cnet.zero11.w" in pc and "cnet.midzero.w" in pc and "cnet.out.conv.w" not in pc


def test_vae_ldm_names_round_trip():
    """weights.vae_ldm_to_diffusers inverts the diffusers -> LDM renaming the fixtures use (ldm/modules/diffusionmodules/model.py names)."""
    import torch
    from instarevive_amd import weights as W
    from oracle import vae as ovae
    from tests.golden._det import det_state_dict
    from tests.test_cldm_gpu import diffusers_to_ldm
    sd = det_state_dict(ovae.state_dict_shapes(dict(ch=32)), seed=9)
    ldm = diffusers_to_ldm(sd)
    assert "encoder.down.0.block.0.norm1.weight" in ldm and "decoder.up.3.upsample.conv.weight" in ldm and ldm["encoder.mid.attn_1.q.weight"].dim() == 4
    back = W.vae_ldm_to_diffusers(ldm)
    assert set(back) == set(sd)
    for k in sd:
        torch.testing.assert_close(back[k].reshape(sd[k].shape), sd[k], rtol=0, atol=0)
    enc_only = W.vae_ldm_to_diffusers({k: v for k, v in ldm.items() if k.startswith(("encoder.", "quant_conv."))})
    assert all(k.startswith(("encoder.", "quant_conv.")) for k in enc_only) and len(enc_only) == sum(k.startswith(("encoder.", "quant_conv.")) for k in sd)


def test_unet_shapes_match_the_reference_parameter_table():
    from instarevive_amd import weights as W
    from oracle import cldm as ocldm
    for cfg in (dict(ocldm.DEFAULT_CFG), dict(ocldm.DEFAULT_CFG, model_channels=64, num_head_channels=32, context_dim=64)):
        for control in (False, True):
            assert {k: tuple(v) for k, v in W.unet_shapes(cfg, control).items()} == {k: tuple(v) for k, v in ocldm.state_dict_shapes(cfg, control).items()}


def test_clip_text_keys_and_packing():
    import torch
    from instarevive_amd import weights as W
    from oracle import clip_text as oclip
    from tests.golden._det import det_state_dict
    cfg = dict(width=64, heads=2, layers=3, vocab_size=100, context_length=77, mlp_ratio=4.0)
    shapes = oclip.state_dict_shapes(cfg)
    assert set(W.clip_text_expected_keys(cfg)) == set(shapes) and {k: tuple(v) for k, v in W.clip_text_shapes(cfg).items()} == {k: tuple(v) for k, v in shapes.items()}
    full = oclip.state_dict_shapes()
    assert len(full) == 4 + 24 * 12 and full["transformer.resblocks.23.mlp.c_fc.weight"] == (4096, 1024)   # ViT-H-14 text tower
    sd = det_state_dict(shapes, seed=3)
    p = W.pack_clip_text(sd, cfg, n_run=2)
    assert "clip.l1.qkv.w" in p and "clip.l2.qkv.w" not in p and p["clip.causal"].shape == (2, 77, 77) and p["clip.embed"].shape == (100, 64)
    assert float(p["clip.causal"][0, 5, 6]) < -1e37 and float(p["clip.causal"][0, 6, 5]) == 0.0
    torch.testing.assert_close(p["clip.l0.qkv.b"][:64], sd["transformer.resblocks.0.attn.in_proj_bias"][:64] * 32 ** -0.5)
    torch.testing.assert_close(p["clip.l0.qkv.b"][64:], sd["transformer.resblocks.0.attn.in_proj_bias"][64:])


def test_evaluate_pairs_psnr_ssim_on_y(tmp_path):
    """tools/evaluate_pairs.py (the paired half of the reference's evaluate_img.py:30-33,40-57; pyiqa's definitions restated): identical
    folders score the eps-limited PSNR and SSIM 1; a uniform grey shift gives the analytic PSNR on BT.601 Y; Gaussian noise lowers both; files
    pair up in sorted order."""
    import importlib.util
    import numpy as np
    from PIL import Image
    spec = importlib.util.spec_from_file_location("evaluate_pairs", os.path.join(ROOT, "tools", "evaluate_pairs.py"))
    ep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ep)
    rng = np.random.default_rng(3)
    base = np.clip(rng.normal(128, 40, (2, 48, 64, 3)), 20, 230).astype(np.uint8)
    for sub in ("gt", "same", "shift", "noise"):
        os.makedirs(tmp_path / sub)
    for i in range(2):
        Image.fromarray(base[i]).save(tmp_path / "gt" / f"im{i}.png")
        Image.fromarray(base[i]).save(tmp_path / "same" / f"im{i}.png")
        Image.fromarray(base[i] + 4).save(tmp_path / "shift" / f"im{i}.png")
        Image.fromarray(np.clip(base[i] + rng.normal(0, 12, base[i].shape), 0, 255).astype(np.uint8)).save(tmp_path / "noise" / f"im{i}.png")
    quiet = lambda *a: None
    same = ep.evaluate(tmp_path / "same", tmp_path / "gt", log=quiet)
    assert same["ssim"] == 1.0 and abs(same["psnr"] - 80.0) < 1e-6            # eps 1e-8 on the UNIT-scale MSE (pyiqa psnr, data_range 1)
    # +4 grey levels on every channel: Y moves by exactly 219 / 255 * 4 / 255 on the unit scale (no rounding on pyiqa's default path)
    shift = ep.evaluate(tmp_path / "shift", tmp_path / "gt", log=quiet)
    assert abs(shift["psnr"] - 10 * np.log10(1.0 / ((219.0 / 255.0 * 4.0 / 255.0) ** 2 + 1e-8))) < 1e-6 and shift["ssim"] > 0.99
    noise = ep.evaluate(tmp_path / "noise", tmp_path / "gt", log=quiet)
    assert 24.0 < noise["psnr"] < 32.0 and 0.3 < noise["ssim"] < 0.97
    # white on black: Y spans 16 .. 235 (studio swing), as pyiqa's color_space='ycbcr'
    assert ep.to_y(np.zeros((1, 1, 3)))[0, 0] == 16 and ep.to_y(np.ones((1, 1, 3)))[0, 0] == 235
    assert abs(ep.to_y(np.full((1, 1, 3), 0.5), 1.0)[0, 0] - (16 + 219 * 0.5) / 255) < 1e-12       # unit scale: not rounded


def test_evaluate_pairs_lpips_against_an_independent_construction(tmp_path):
    """LPIPS v0.1 / alex as tools/evaluate_pairs.py restates it (utils/metrics.py:41-66, evaluate_img.py:32) against the same network built
    here from torch.nn modules in torchvision's `features` order (Conv 0, ReLU 1, MaxPool 2, Conv 3, ReLU 4, MaxPool 5, Conv 6, ReLU 7, Conv 8,
    ReLU 9, Conv 10, ReLU 11) with seeded random weights - the pretrained files do not exist offline (parity unpinned for the weights, not for
    the arithmetic). Both weight layouts load (torchvision + lin heads; one full lpips state dict); d(x, x) = 0; symmetry; the folder
    evaluation reports the mean."""
    import importlib.util
    from PIL import Image
    spec = importlib.util.spec_from_file_location("evaluate_pairs", os.path.join(ROOT, "tools", "evaluate_pairs.py"))
    ep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ep)
    nn = torch.nn
    torch.manual_seed(5)
    feats = nn.Sequential(nn.Conv2d(3, 64, 11, 4, 2), nn.ReLU(), nn.MaxPool2d(3, 2), nn.Conv2d(64, 192, 5, padding=2), nn.ReLU(), nn.MaxPool2d(3, 2),
                          nn.Conv2d(192, 384, 3, padding=1), nn.ReLU(), nn.Conv2d(384, 256, 3, padding=1), nn.ReLU(), nn.Conv2d(256, 256, 3, padding=1), nn.ReLU())
    lins = [torch.rand(1, c, 1, 1) for c in (64, 192, 384, 256, 256)]
    alex_sd = {"features." + k: v for k, v in feats.state_dict().items()}
    alex_sd.update({"classifier.1.weight": torch.zeros(2, 2)})                                   # the real file also carries the classifier
    lin_sd = {f"lin{k}.model.1.weight": w for k, w in enumerate(lins)}
    torch.save(alex_sd, tmp_path / "alexnet.pth")
    torch.save(lin_sd, tmp_path / "alex_lin.pth")
    full = dict(lin_sd)
    for k, key in enumerate(ep.LPIPS_SLICE_KEY):
        idx = ep.ALEX_CONVS[k][0]
        full[key + ".weight"], full[key + ".bias"] = alex_sd[f"features.{idx}.weight"], alex_sd[f"features.{idx}.bias"]
    torch.save(full, tmp_path / "lpips_full.pth")
    a, b = torch.rand(2, 3, 96, 80), torch.rand(2, 3, 96, 80)

    def want(x, y):
        shift, scale = torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1), torch.tensor([.458, .448, .450]).view(1, 3, 1, 1)
        tot = torch.zeros(x.shape[0])
        fx, fy = (2 * x - 1 - shift) / scale, (2 * y - 1 - shift) / scale
        k = 0
        with torch.no_grad():
            for layer in feats:
                fx, fy = layer(fx), layer(fy)
                if isinstance(layer, nn.ReLU):
                    nx = fx / (fx.norm(dim=1, keepdim=True) + 1e-10)
                    ny = fy / (fy.norm(dim=1, keepdim=True) + 1e-10)
                    tot += torch.nn.functional.conv2d((nx - ny) ** 2, lins[k]).mean((1, 2, 3))
                    k += 1
        return tot

    for net in (ep.LPIPS(str(tmp_path / "alexnet.pth"), str(tmp_path / "alex_lin.pth")), ep.LPIPS(None, str(tmp_path / "lpips_full.pth"))):
        got = net(a, b)
        torch.testing.assert_close(got, want(a, b), rtol=1e-4, atol=1e-6)
        assert float(net(a, a).abs().max()) == 0.0 and float(got.min()) > 0
        torch.testing.assert_close(net(b, a), got, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(net(2 * a - 1, 2 * b - 1, normalize=False), got, rtol=1e-4, atol=1e-6)
    with pytest.raises(KeyError):
        ep.LPIPS(str(tmp_path / "alexnet.pth"), None)
    for sub in ("x", "y"):
        os.makedirs(tmp_path / sub)
    for i in range(2):
        Image.fromarray((a[i].permute(1, 2, 0) * 255).to(torch.uint8).numpy()).save(tmp_path / "x" / f"{i}.png")
        Image.fromarray((b[i].permute(1, 2, 0) * 255).to(torch.uint8).numpy()).save(tmp_path / "y" / f"{i}.png")
    res = ep.evaluate(tmp_path / "x", tmp_path / "y", log=lambda *a_: None, lpips=net)
    q = lambda t: (t * 255).to(torch.uint8).float() / 255
    assert abs(res["lpips"] - float(want(q(a), q(b)).mean())) < 1e-4 and "psnr" in res and "ssim" in res


def test_clip_bpe_tokenizer_matches_transformers_and_open_clip_layout(tmp_path):
    """instarevive_amd/clip_bpe.py (open_clip.tokenize restated; ldm/modules/encoders/modules.py:171) on a BPE table built on the spot:
    (1) the vocabulary layout open_clip derives from its merges file (bytes, bytes + </w>, merges, the two specials last); (2) the ids agree
    with transformers.CLIPTokenizer - an independent implementation of the same byte-level BPE, installed here - loaded from the SAME table
    written in the Hugging Face form; (3) row layout: <start_of_text>, ids, <end_of_text>, zero padding, truncation keeps the end token."""
    import gzip
    import json
    from instarevive_amd.clip_bpe import ClipBPETokenizer, bytes_to_unicode
    words = "a photo of a cat the quick brown fox jumps over lazy dog restoration high quality image".split()
    # a small deterministic merge table: the character pairs of these words, most frequent first (any ranked list is a valid BPE table)
    seq = [tuple(w[:-1]) + (w[-1] + "</w>",) for w in words]
    merges = []
    for _ in range(60):
        cnt = {}
        for w in seq:
            for p in zip(w[:-1], w[1:]):
                cnt[p] = cnt.get(p, 0) + 1
        if not cnt:
            break
        best = max(sorted(cnt), key=lambda p: cnt[p])
        merges.append(best)
        seq = [tuple(_merge(w, best)) for w in seq]
    oc = tmp_path / "oc"
    os.makedirs(oc)
    with gzip.open(oc / "bpe_simple_vocab_16e6.txt.gz", "wt", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    tok = ClipBPETokenizer.from_folder(str(oc))
    bu = list(bytes_to_unicode().values())
    assert tok.encoder[bu[0]] == 0 and tok.encoder[bu[0] + "</w>"] == 256 and tok.encoder["".join(merges[0])] == 512
    assert tok.sot == 512 + len(merges) and tok.eot == tok.sot + 1
    texts = ["a photo of a cat", "The quick  brown fox's image, 42 dogs!", "", "restoration " * 100]
    ids = tok(texts)
    assert ids.shape == (4, 77) and ids.dtype == torch.long
    assert ids[2].tolist() == [tok.sot, tok.eot] + [0] * 75                    # the empty prompt = the embedder's built-in row
    assert ids[3, 0] == tok.sot and ids[3, -1] == tok.eot and (ids[3] != 0).all()  # truncated, end token kept
    n0 = int((ids[0] != 0).sum())
    assert ids[0, 0] == tok.sot and ids[0, n0 - 1] == tok.eot and (ids[0, n0:] == 0).all()
    # (2) the same table through transformers' CLIP tokenizer
    from transformers import CLIPTokenizer
    hf = tmp_path / "hf"
    os.makedirs(hf)
    enc = {k: v for k, v in tok.encoder.items()}
    enc["<|startoftext|>"] = enc.pop("<start_of_text>")
    enc["<|endoftext|>"] = enc.pop("<end_of_text>")
    json.dump(enc, open(hf / "vocab.json", "w"))
    with open(hf / "merges.txt", "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    ref = CLIPTokenizer(str(hf / "vocab.json"), str(hf / "merges.txt"))
    for t in texts[:2] + ["high quality image restoration", "jumps over the lazy dog"]:
        want = ref(t)["input_ids"]
        got = [int(v) for v in tok(t)[0] if v != 0]
        assert got == want, (t, got, want)
    # the Hugging Face form of the folder loads to the same tokenizer
    tok2 = ClipBPETokenizer.from_folder(str(hf))
    assert torch.equal(tok2(texts), ids)
    # and the embedder takes the folder
    from instarevive_amd.cldm import FrozenOpenCLIPEmbedder
    emb = FrozenOpenCLIPEmbedder(layer="penultimate", tokenizer=str(oc))
    assert torch.equal(emb.tokenize(texts), ids)


def _merge(word, pair):
    out, i = [], 0
    while i < len(word):
        if i < len(word) - 1 and (word[i], word[i + 1]) == pair:
            out.append(word[i] + word[i + 1])
            i += 2
        else:
            out.append(word[i])
            i += 1
    return out


def test_caption_mojibake_repair():
    """captions.fix_mojibake: the dominant case of ftfy's encoding repair (UTF-8 read as cp1252 / Latin-1, one to three layers), restated
    because ftfy is absent (diffusion/model/t5.py:118-124 calls ftfy.fix_text first); text that is not mojibake must pass untouched."""
    from instarevive_amd.captions import clean_caption, fix_mojibake, fix_text
    assert fix_mojibake("caf\u00c3\u00a9 au lait") == "caf\u00e9 au lait"
    assert fix_text("it\u00e2\u20ac\u2122s a dog\u00e2\u20ac\u00a6") == "it's a dog\u2026"                 # -> curly apostrophe -> uncurled; the ellipsis stays
    assert fix_mojibake("\u00c3\u00a2\u00e2\u201a\u00ac\u00e2\u201e\u00a2 twice") == "\u2019 twice"       # two layers
    assert fix_mojibake("\u00e2\u20ac\u0153quoted\u00e2\u20ac\u009d") == "\u201cquoted\u201d"             # 0x9D: a byte cp1252 leaves undefined
    for clean in ("na\u00efve caf\u00e9 \u00e9\u00e0 \u00fcn\u00ef", "plain ascii", "\u00c3", "\u4f60\u597d \u00df\u00fc", "a \u00d7 b"):
        assert fix_mojibake(clean) == clean
    # inside clean_caption the repair sees what the EARLIER steps left (t5.py:124-197 lower-case the caption and strip (c) / (tm) / (r) first, exactly
    # as in the reference, where ftfy sits at the same place): "A caf\u00c3\u00a9" arrives as "a caf\u00e3" and stays that way there, too
    assert clean_caption("A caf\u00c3\u00a9 in Paris") == "a caf\u00e3 in paris"
    text = "a photo of a cat"   # ASCII captions (every fixture caption) never reach the repair
    assert clean_caption(text) == text


def test_no_compiler_placed_hazard_in_front_of_asm_mfmas():
    """The pinned MFMA streams are `asm` statements, which hipcc's hazard recogniser does not look into: a VALU instruction it schedules directly in
    front of one (the zeroed C operand of a score MFMA: flash_attn_x72_kernel's round-4 bug, wrong results in some lanes from run to run) or an asm
    VALU instruction directly behind a transcendental is silent on every functional test that happens not to hit the timing. tools/mfma_hazard_scan.py
    cross-compiles every kernel source with asm MFMAs to ISA (no GPU needed) and looks for such pairs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mfma_hazard_scan", os.path.join(ROOT, "tools", "mfma_hazard_scan.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    res = m.scan_all()
    assert sum(n for _, n in res.values()) > 4000, "the scan must see the kernels' MFMAs"
    bad = {os.path.basename(f): [(k[:40], w, ins) for k, ins, w, d in fl][:4] for f, (fl, n) in res.items() if fl}
    assert not bad, bad


def test_host_pools_order_bounds_and_errors(tmp_path):
    """inference.HostPools: reads come back in submission order however long each takes, at most `depth` run ahead, writes are bounded,
    a failed write surfaces at drain(), workers = 0 is the inline (reference) form, and the threaded run writes the same PNG bytes'
    pixels as the inline one."""
    import importlib
    import threading
    import time
    from argparse import Namespace
    from PIL import Image
    sys.path.insert(0, ROOT)
    inf = importlib.import_module("inference")
    pools = inf.HostPools(4)
    started, lock = [], threading.Lock()

    def slow(i):
        with lock:
            started.append(i)
        time.sleep(0.02 * ((7 - i) % 4))
        return i * i

    out = []
    for v in pools.read_ahead(slow, range(40)):
        out.append(v)
        with lock:
            assert len(started) - len(out) <= pools.depth          # bounded look-ahead
    assert out == [i * i for i in range(40)]
    running, peak, done = [0], [0], []

    def w(i):
        with lock:
            running[0] += 1
            peak[0] = max(peak[0], running[0])
        time.sleep(0.005)
        with lock:
            running[0] -= 1
            done.append(i)

    for i in range(50):
        pools.write_behind(w, i)
    pools.drain()
    assert sorted(done) == list(range(50)) and peak[0] <= 4 and pools.written == 50
    bad = inf.HostPools(2)
    bad.write_behind(lambda: (_ for _ in ()).throw(OSError("disk full")))
    with pytest.raises(OSError):
        bad.drain()
    inline = inf.HostPools(0)
    assert list(inline.read_ahead(lambda i: i + 1, range(5))) == [1, 2, 3, 4, 5]
    seen = []
    inline.write_behind(seen.append, 3)
    assert seen == [3]
    inline.drain()
    assert inf.default_workers(1) >= 1 and inf.default_workers(10 ** 6) == 1
    # same files from the threaded and the inline form
    from tests.golden._det import det_input
    src = tmp_path / "in"
    src.mkdir()
    for i in range(6):
        Image.fromarray((det_input(10 + i, (40 + 8 * i, 56, 3)) * 255).numpy().astype(np.uint8)).save(src / f"f{i}.png")
    outs = {}
    for workers in (0, 3):
        a = Namespace(input=str(src), output=str(tmp_path / f"out{workers}"), sr_scale=1.5, tiled=False, tile_size=512, use_center_crop=False,
                      show_lq=True, disable_preprocess_model=False)
        hp = inf.HostPools(workers)
        files = sorted(str(p) for p in src.iterdir())
        for job in hp.read_ahead(lambda f: inf.read_job(f, 0, a), files):
            hp.write_behind(inf.write_job, job, 255 - job.net_in, job.net_in, a)
        hp.drain()
        outs[workers] = {p.name: np.array(Image.open(p)) for p in sorted((tmp_path / f"out{workers}").iterdir())}
    assert list(outs[0]) == list(outs[3]) and len(outs[0]) == 6
    assert all(np.array_equal(outs[0][k], outs[3][k]) for k in outs[0])
    # --png_compress_level changes the file, not the pixels
    a1 = Namespace(**{**vars(a), "output": str(tmp_path / "out_l1"), "png_compress_level": 1})
    for job in inf.HostPools(0).read_ahead(lambda f: inf.read_job(f, 0, a1), files):
        inf.write_job(job, 255 - job.net_in, job.net_in, a1)
    l1 = {p.name: np.array(Image.open(p)) for p in sorted((tmp_path / "out_l1").iterdir())}
    assert list(l1) == list(outs[0]) and all(np.array_equal(l1[k], outs[0][k]) for k in l1)


def test_default_workers_on_a_shared_host(monkeypatch):
    """8 ranks on a 64-core host. Each rank's part is 8 cores: one is kept for the feeding thread (two from 9 cores up), and the CLI
    says at start-up when that many encoders cannot keep ahead of one GPU at the output size - and what to do about it."""
    import importlib
    inf = importlib.import_module("inference")
    monkeypatch.delenv("IR_WORKERS", raising=False)
    monkeypatch.setattr(inf, "cpu_share", lambda: 64)
    assert inf.default_workers(8) == 7 and inf.default_workers(4) == 14 and inf.default_workers(1) == 16 and inf.default_workers(64) == 1
    monkeypatch.setattr(inf, "cpu_share", lambda: 16)   # the one-GPU box of this pool
    assert inf.default_workers(1) == 14
    note = inf.host_keeps_up(7, 2048 * 2048, None)
    assert note.startswith("host-bound") and "--png_compress_level" in note
    assert inf.host_keeps_up(14, 2048 * 2048, None) == "" and inf.host_keeps_up(7, 2048 * 2048, 1) == ""
    assert inf.host_keeps_up(0, 2048 * 2048, None) == ""   # inline mode: the reference's behaviour, nothing to size
    monkeypatch.setenv("IR_WORKERS", "3")
    assert inf.default_workers(8) == 3


def test_hub_ids_resolve_through_the_hf_cache(tmp_path, monkeypatch):
    """The reference loads `stabilityai/sd-vae-ft-ema` and `PixArt-alpha/PixArt-Alpha-DMD-XL-2-512x512` (subfolders transformer / scheduler)
    by hub id (test_scripts/inference.py:36,236,238), i.e. from <cache>/models--org--name/snapshots/<rev>/ with blobs behind symlinks.
    _resolve_pretrained must find them there without a network: refs/main wins, else the newest snapshot; $HF_HUB_CACHE before $HF_HOME/hub
    before ~/.cache/huggingface/hub; a missing id raises FileNotFoundError naming the caches."""
    import time
    from instarevive_amd import models
    home = tmp_path / "home"
    hub = home / ".cache" / "huggingface" / "hub"
    repo = hub / "models--PixArt-alpha--PixArt-Alpha-DMD-XL-2-512x512"
    old, new = repo / "snapshots" / "aaaa", repo / "snapshots" / "bbbb"
    for snap, beta_end in ((old, 0.03), (new, 0.02)):
        (snap / "scheduler").mkdir(parents=True)
        blob = repo / "blobs" / f"blob{beta_end}"
        blob.parent.mkdir(exist_ok=True)
        blob.write_text(json.dumps({"_class_name": "DDPMScheduler", "num_train_timesteps": 1000, "beta_start": 1e-4, "beta_end": beta_end,
                                    "beta_schedule": "linear"}))
        os.symlink(blob, snap / "scheduler" / "scheduler_config.json")      # the cache stores files as links into blobs/
        (snap / "transformer").mkdir()
    past = time.time() - 1000
    os.utime(old, (past, past))
    for var in ("HF_HOME", "HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE", "XDG_CACHE_HOME"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("HOME", str(home))
    monkeypatch.chdir(tmp_path)                                              # no ./weights here
    hub_id = "PixArt-alpha/PixArt-Alpha-DMD-XL-2-512x512"
    assert models._resolve_pretrained(hub_id, "transformer") == str(new / "transformer")        # newest snapshot
    assert float(models.DDPMScheduler.from_pretrained(hub_id, subfolder="scheduler").config.beta_end) == 0.02
    (repo / "refs").mkdir()
    (repo / "refs" / "main").write_text("aaaa\n")
    assert models._resolve_pretrained(hub_id, "scheduler") == str(old / "scheduler")            # refs/main decides
    assert float(models.DDPMScheduler.from_pretrained(hub_id, subfolder="scheduler").config.beta_end) == 0.03
    # $HF_HOME/hub and $HF_HUB_CACHE take precedence over the home cache, in huggingface_hub's order
    alt = tmp_path / "alt" / "hub" / "models--stabilityai--sd-vae-ft-ema" / "snapshots" / "cccc"
    alt.mkdir(parents=True)
    with pytest.raises(FileNotFoundError) as e:
        models._resolve_pretrained("stabilityai/sd-vae-ft-ema")
    assert "huggingface" in str(e.value)
    monkeypatch.setenv("HF_HOME", str(tmp_path / "alt"))
    assert models._resolve_pretrained("stabilityai/sd-vae-ft-ema") == str(alt)
    direct = tmp_path / "direct" / "models--stabilityai--sd-vae-ft-ema" / "snapshots" / "dddd"
    direct.mkdir(parents=True)
    monkeypatch.setenv("HF_HUB_CACHE", str(tmp_path / "direct"))
    assert models._resolve_pretrained("stabilityai/sd-vae-ft-ema") == str(direct)
    # a local folder of the same name still wins (the reference's from_pretrained does the same)
    (tmp_path / "stabilityai" / "sd-vae-ft-ema").mkdir(parents=True)
    assert models._resolve_pretrained("stabilityai/sd-vae-ft-ema") == "stabilityai/sd-vae-ft-ema"


def test_bench_helpers_of_round_5(tmp_path, monkeypatch):
    """bench.py's round-5 helpers without a GPU: (1) roofline.traffic comes only from a PMC file collected on the CURRENT kernel sources
    (instarevive_amd.build.source_hash) - a file without the hash or with another one is refused, with the reason; (2) tools/power_sampler.summarise picks
    the card by PCI address (falling back to the busiest one) and averages over the timed window only; (3) tools/cli_artifacts.parse_cli_rate reads the
    summary line inference.py prints; (4) tests/support/stress_weights scales exactly the advertised rows and leaves the input dicts alone."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from instarevive_amd.build import source_hash
    h = source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", h) and h == source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    f, why = bench.pmc_file_for_this_tree()
    assert f is None and "none found" in why
    (prof / "r04_pmc_kernels.json").write_text(json.dumps({"per_kernel": {}}))
    (prof / "r05_pmc_kernels.json").write_text(json.dumps({"csrc_sha16": "0" * 16, "per_kernel": {}}))
    f, why = bench.pmc_file_for_this_tree()
    assert f is None and "stale" in why and "r05_pmc_kernels.json" in why and h in why
    (prof / "r06_pmc_kernels.json").write_text(json.dumps({"csrc_sha16": h, "per_kernel": {}}))
    f, why = bench.pmc_file_for_this_tree()
    assert f.endswith("r06_pmc_kernels.json") and why is None
    # (2)
    from tools.power_sampler import summarise
    tr = tmp_path / "trace.txt"
    lines = ["# cards ['/sys/devices/pci0000:00/0000:05:00.0', '/sys/devices/pci0000:c0/0000:dc:00.0']; columns: t_unix card sclk_mhz power_w mclk_mhz temp_c"]
    for i in range(100):
        t = 1000.0 + 0.02 * i
        lines += [f"{t:.4f} 0 95 237.0 2000 nan", f"{t:.4f} 1 {2100 if 20 <= i < 60 else 300} {1250.0 if 20 <= i < 60 else 240.0} 2000 55.0"]
    tr.write_text("\n".join(lines) + "\n")
    s1 = summarise(str(tr), 1000.4, 1001.18, "0000:dc:00.0")
    assert s1["card"] == 1 and s1["clock_mhz"] == 2100.0 and s1["power_w"] == 1250.0 and s1["samples"] == 40 and "PCI address" in s1["selection"]
    assert summarise(str(tr), 1000.4, 1001.18, "0000:05:00.0")["clock_mhz"] == 95.0
    assert summarise(str(tr), 1000.4, 1001.18)["card"] == 1                                 # no address: the busiest card
    assert summarise(str(tr), 5.0, 6.0) is None and summarise(str(tmp_path / "absent"), 0, 1) is None
    # (3)
    from tools.cli_artifacts import parse_cli_rate
    txt = "save to x\n[rank 0] wrote 48 files in 6.500 s = 7.385 files/s (14 host threads); after the first result: 47 files in 6.100 s = 7.705 files/s\n"
    r = parse_cli_rate(txt)
    assert r == [dict(rank=0, files=48, seconds=6.5, files_per_s=7.385, workers=14, steady_files=47, steady_seconds=6.1, steady_files_per_s=7.705)]
    assert parse_cli_rate("nothing") == []
    # (4)
    from tests.support.stress_weights import stress_state_dicts
    g = torch.Generator().manual_seed(0)
    dit = {}
    for l in range(2):
        p = f"transformer_blocks.{l}."
        for n, shp in (("attn1.to_q", (8, 8)), ("attn1.to_k", (8, 8)), ("attn1.to_v", (8, 8)), ("attn1.to_out.0", (8, 8)), ("attn2.to_out.0", (8, 8)), ("ff.net.0.proj", (32, 8)),
                       ("ff.net.2", (8, 32))):
            dit[p + n + ".weight"], dit[p + n + ".bias"] = torch.rand(shp, generator=g), torch.rand(shp[0], generator=g)
    vae = {"decoder.up_blocks.0.resnets.0.conv1.weight": torch.rand(8, 8, 3, 3, generator=g), "decoder.up_blocks.0.resnets.0.conv1.bias": torch.rand(8, generator=g),
           "decoder.mid_block.attentions.0.to_q.weight": torch.rand(8, 8, generator=g), "decoder.mid_block.attentions.0.to_q.bias": torch.rand(8, generator=g),
           "decoder.conv_out.weight": torch.rand(3, 8, 3, 3, generator=g)}
    sds = {"swin": {}, "vae": vae, "dit": dit}
    keep = {k: v.clone() for k, v in dit.items()}
    out = stress_state_dicts(sds, frac=0.125, gain=30.0, logit_gain={"dit": [4.0, 1.0], "vae_encoder": 1.0, "vae_decoder": 9.0})
    assert all(torch.equal(dit[k], keep[k]) for k in dit), "the input state dicts must stay untouched"
    w0, w1 = dit["transformer_blocks.0.attn1.to_out.0.weight"], out["dit"]["transformer_blocks.0.attn1.to_out.0.weight"]
    ratio = (w1 / w0)[:, 0]
    assert int((ratio > 29).sum()) == 1 and int(((ratio - 1).abs() < 1e-6).sum()) == 7                       # one of eight stream channels x30 ...
    r2 = (out["dit"]["transformer_blocks.1.ff.net.2.weight"] / dit["transformer_blocks.1.ff.net.2.weight"])[:, 0]
    assert torch.equal(r2 > 29, ratio > 29)                                                                  # ... the SAME channel in every block and writer
    assert int(((out["dit"]["transformer_blocks.0.ff.net.0.proj.bias"] / dit["transformer_blocks.0.ff.net.0.proj.bias"]) > 29).sum()) == 4   # 1/8 of 32 hidden units
    torch.testing.assert_close(out["dit"]["transformer_blocks.0.attn1.to_q.weight"], dit["transformer_blocks.0.attn1.to_q.weight"] * 2.0)   # sqrt(4) on q and on k
    torch.testing.assert_close(out["dit"]["transformer_blocks.0.attn1.to_k.bias"], dit["transformer_blocks.0.attn1.to_k.bias"] * 2.0)
    assert torch.equal(out["dit"]["transformer_blocks.1.attn1.to_q.weight"], dit["transformer_blocks.1.attn1.to_q.weight"])                 # block 1: gain 1
    assert torch.equal(out["dit"]["transformer_blocks.0.attn1.to_v.weight"], dit["transformer_blocks.0.attn1.to_v.weight"])
    torch.testing.assert_close(out["vae"]["decoder.mid_block.attentions.0.to_q.weight"], vae["decoder.mid_block.attentions.0.to_q.weight"] * 3.0)
    assert torch.equal(out["vae"]["decoder.conv_out.weight"], vae["decoder.conv_out.weight"])
    assert int(((out["vae"]["decoder.up_blocks.0.resnets.0.conv1.bias"] / vae["decoder.up_blocks.0.resnets.0.conv1.bias"]) > 29).sum()) == 1
