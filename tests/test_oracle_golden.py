"""The oracle (oracle/*.py, a CPU restatement) is pinned against outputs of the reference's own modules, captured by
tests/golden/make_golden.py in the build container. CPU only; runs in seconds."""
import os

import numpy as np
import pytest
import torch

from oracle import dit as odit
from oracle import glue as oglue
from oracle import swinir as oswin
from oracle import vae as ovae
from tests.golden._det import checksum, det_state_dict

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return {k: (torch.from_numpy(v) if v.dtype.kind == "f" else v) for k, v in np.load(os.path.join(G, name)).items()}


def test_swinir_matches_reference():
    fx = load("swinir_small.npz")
    cfg = dict(embed_dim=60, depths=[2, 2], num_heads=[6, 6])
    sd = det_state_dict(oswin.state_dict_shapes(cfg), seed=101)
    assert abs(checksum(sd) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])  # deterministic weights reproduced
    for k in ("x64", "x128x192"):  # 64^2: one window per image (8x8 grid); 128x192: on-the-fly shift masks
        out = oswin.swinir_forward(sd, fx[k], cfg)
        assert out.shape == fx[k + "_out"].shape
        torch.testing.assert_close(out, fx[k + "_out"], rtol=1e-4, atol=2e-5)


def test_vae_matches_reference():
    fx = load("vae_small.npz")
    cfg = dict(ch=32)
    sd = det_state_dict(ovae.state_dict_shapes(cfg), seed=202)
    assert abs(checksum(sd) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
    for k in ("x64", "x64x128"):
        torch.testing.assert_close(ovae.vae_encode_mean(sd, fx[k], cfg), fx[k + "_mean"], rtol=1e-4, atol=2e-5)
    for k in ("z8", "z8x16"):
        torch.testing.assert_close(ovae.vae_decode(sd, fx[k], cfg), fx[k + "_dec"], rtol=1e-4, atol=2e-5)


def _pixart_block_shapes(p, C):
    return {p + "scale_shift_table": (6, C), p + "attn.qkv.weight": (3 * C, C), p + "attn.qkv.bias": (3 * C,),
            p + "attn.proj.weight": (C, C), p + "attn.proj.bias": (C,), p + "cross_attn.q_linear.weight": (C, C),
            p + "cross_attn.q_linear.bias": (C,), p + "cross_attn.kv_linear.weight": (2 * C, C), p + "cross_attn.kv_linear.bias": (2 * C,),
            p + "cross_attn.proj.weight": (C, C), p + "cross_attn.proj.bias": (C,), p + "mlp.fc1.weight": (4 * C, C),
            p + "mlp.fc1.bias": (4 * C,), p + "mlp.fc2.weight": (C, 4 * C), p + "mlp.fc2.bias": (C,)}


def _pixart_shapes(depth, C, cap, prefix=""):
    """in-tree PixArtMS parameter names (what make_golden.py seeded)"""
    shapes = {"x_embedder.proj.weight": (C, 4, 2, 2), "x_embedder.proj.bias": (C,),
              "t_embedder.mlp.0.weight": (C, 256), "t_embedder.mlp.0.bias": (C,), "t_embedder.mlp.2.weight": (C, C), "t_embedder.mlp.2.bias": (C,),
              "t_block.1.weight": (6 * C, C), "t_block.1.bias": (6 * C,),
              "y_embedder.y_proj.fc1.weight": (C, cap), "y_embedder.y_proj.fc1.bias": (C,),
              "y_embedder.y_proj.fc2.weight": (C, C), "y_embedder.y_proj.fc2.bias": (C,),
              "final_layer.linear.weight": (32, C), "final_layer.linear.bias": (32,), "final_layer.scale_shift_table": (2, C)}
    for d in range(depth):
        shapes.update(_pixart_block_shapes(f"blocks.{d}.", C))
    return {prefix + k: v for k, v in shapes.items()}


def _dit_small():
    depth, heads, hidden, cap = 2, 2, 144, 64
    sd = det_state_dict(_pixart_shapes(depth, hidden, cap), seed=303)  # converted below with the restated key map
    cfg = dict(num_layers=depth, num_attention_heads=heads, attention_head_dim=hidden // heads, sample_size=16, caption_channels=cap)
    return sd, odit.pixart_to_diffusers(sd, depth), cfg


def _dit_control_small():
    depth, heads, hidden, cap, ncopy = 4, 4, 288, 64, 2
    shapes = _pixart_shapes(depth, hidden, cap, "base_model.")
    for i in range(ncopy):
        shapes.update(_pixart_block_shapes(f"controlnet.{i}.copied_block.", hidden))
        for n in ("after_proj",) + (("before_proj",) if i == 0 else ()):
            shapes.update({f"controlnet.{i}.{n}.weight": (hidden, hidden), f"controlnet.{i}.{n}.bias": (hidden,)})
    sd = det_state_dict(shapes, seed=404)
    bsd = {k[len("base_model."):]: v for k, v in sd.items() if k.startswith("base_model.")}
    dsd = dict(odit.pixart_to_diffusers(bsd, depth), **odit.control_to_diffusers(sd, ncopy))
    cfg = dict(num_layers=depth, num_attention_heads=heads, attention_head_dim=hidden // heads, sample_size=16, caption_channels=cap,
               copy_blocks_num=ncopy)
    return sd, dsd, cfg


def test_dit_control_matches_reference_wiring():
    """ControlNet-Half branch (pixart_controlnet.py:40-128) against the fixture made from the reference."""
    fx = load("dit_control_small.npz")
    sd, dsd, cfg = _dit_control_small()
    assert abs(checksum(sd) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
    assert abs(checksum(dsd) - float(fx["wsum_diffusers"])) < 1e-6 * float(fx["wsum_diffusers"])
    out = odit.dit_forward(dsd, fx["lat"], 400.0, fx["y"], None, cfg, c=fx["c"])
    torch.testing.assert_close(out, fx["out_c"], rtol=2e-4, atol=2e-5)
    out0 = odit.dit_forward(dsd, fx["lat"], 400.0, fx["y"], None, cfg)
    torch.testing.assert_close(out0, fx["out_0"], rtol=2e-4, atol=2e-5)
    assert (fx["out_c"] - fx["out_0"]).abs().max() > 1e-2  # the control branch really contributes in the fixture


def test_dit_matches_reference_wiring():
    fx = load("dit_small.npz")
    sd, dsd, cfg = _dit_small()
    assert abs(checksum(sd) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
    assert abs(checksum(dsd) - float(fx["wsum_diffusers"])) < 1e-6 * float(fx["wsum_diffusers"])
    y, mask = fx["y"], fx["mask"]  # y [1,L,cap], mask [1,L]
    for k in ("lat16", "lat16x24"):  # square and non-square latent (pos-emb regeneration)
        lat = fx[k]
        out = odit.dit_forward(dsd, lat, 400.0, y, None, cfg)
        torch.testing.assert_close(out, fx[k + "_nomask"], rtol=2e-4, atol=2e-5)
        # the in-tree twin DROPS padded tokens; a 2-D mask in diffusers becomes -10000 on them: identical in fp32
        out = odit.dit_forward(dsd, lat, 400.0, y, mask, cfg)
        torch.testing.assert_close(out, fx[k + "_mask2d"], rtol=2e-4, atol=2e-5)


def test_micro_conditioning_matches_the_reference_size_embedders():
    """sample_size-128 models (generate.py:56-62): what oracle.dit.micro_condition adds to the timestep embedding equals what the reference's two
    SizeEmbedder modules produce, wired as controlnet.py:189-191 (fixture from the imported modules, tests/golden/make_golden_r6.py). The diffusers
    class that the CLI would execute (PixArtAlphaCombinedTimestepSizeEmbeddings) is not in the tree: that equality is UNPINNED."""
    fx = load("size_embedder.npz")
    S = 96
    shapes = {"mlp.0.weight": (S, 256), "mlp.0.bias": (S,), "mlp.2.weight": (S, S), "mlp.2.bias": (S,)}
    sd_c, sd_a = det_state_dict(shapes, seed=611), det_state_dict(shapes, seed=612)
    assert abs(checksum(sd_c) - float(fx["wsum_c"])) < 1e-6 * abs(float(fx["wsum_c"])) and abs(checksum(sd_a) - float(fx["wsum_a"])) < 1e-6 * abs(float(fx["wsum_a"]))
    dsd = {}
    for pre, sd in (("adaln_single.emb.resolution_embedder.", sd_c), ("adaln_single.emb.aspect_ratio_embedder.", sd_a)):
        for a, b in (("linear_1", "mlp.0"), ("linear_2", "mlp.2")):
            dsd[pre + a + ".weight"], dsd[pre + a + ".bias"] = sd[b + ".weight"], sd[b + ".bias"]
    for name, (h, w) in (("16x24", (16, 24)), ("64x64", (64, 64)), ("128x96", (128, 96))):
        torch.testing.assert_close(odit.micro_condition(dsd, 2, h, w), fx["add_" + name], rtol=1e-5, atol=1e-6)


KVC_VARIANTS = {   # name -> (kv_compress_config of the reference's PixArtMS, qk_norm)
    "conv": ({"sampling": "conv", "scale_factor": 2, "kv_compress_layer": [1]}, False),
    "uniform_qknorm": ({"sampling": "uniform", "scale_factor": 2, "kv_compress_layer": [0, 1]}, True),
    "ave": ({"sampling": "ave", "scale_factor": 2, "kv_compress_layer": [0]}, False)}


def _dit_kvc_small(name):
    """The seeded in-tree state dict make_golden_r6.py::golden_dit_kvc built for variant `name`, its diffusers-keyed form and the oracle's config."""
    depth, heads, hidden, cap = 2, 4, 288, 64
    kvc, qkn = KVC_VARIANTS[name]
    shapes = _pixart_shapes(depth, hidden, cap)
    for d in range(depth):
        q = f"blocks.{d}.attn."
        if kvc["sampling"] == "conv" and d in kvc["kv_compress_layer"]:
            shapes.update({q + "sr.weight": (hidden, 1, 2, 2), q + "sr.bias": (hidden,), q + "norm.weight": (hidden,), q + "norm.bias": (hidden,)})
        if qkn:
            shapes.update({q + "q_norm.weight": (hidden,), q + "q_norm.bias": (hidden,), q + "k_norm.weight": (hidden,), q + "k_norm.bias": (hidden,)})
    sd = det_state_dict(shapes, seed=909)
    for k in sd:
        if k.endswith(("attn.norm.weight", "attn.q_norm.weight", "attn.k_norm.weight")):
            sd[k] = sd[k] + 1.0
    cfg = dict(num_layers=depth, num_attention_heads=heads, attention_head_dim=hidden // heads, sample_size=16, caption_channels=cap, qk_norm=qkn,
               kv_compress=dict(sampling=kvc["sampling"], scale_factor=kvc["scale_factor"], layers=tuple(kvc["kv_compress_layer"])))
    return sd, odit.pixart_to_diffusers(sd, depth), cfg


def test_dit_kv_compression_and_qk_norm_match_the_reference():
    """AttentionKVCompress (PixArt_blocks.py:60-158) inside the reference's PixArtMS: KV token compression by the depthwise 'conv' sampler + LayerNorm
    in one block, by 'uniform' picking with LayerNorm on q and k (qk_norm) in both blocks, by 'ave' in the first block - the oracle against the
    outputs of the imported model (dit_kvc_small.npz). VERDICT r05 missing 3."""
    fx = load("dit_kvc_small.npz")
    for name in KVC_VARIANTS:
        sd, dsd, cfg = _dit_kvc_small(name)
        assert abs(checksum(sd) - float(fx["wsum_" + name])) < 1e-6 * float(fx["wsum_" + name]), name
        assert abs(checksum(dsd) - float(fx["wsum_diffusers_" + name])) < 1e-6 * float(fx["wsum_diffusers_" + name]), name
        out = odit.dit_forward(dsd, fx["lat"], 400.0, fx["y"], None, cfg)
        torch.testing.assert_close(out, fx["out_" + name], rtol=2e-4, atol=2e-5)
        plain = odit.dit_forward(dsd, fx["lat"], 400.0, fx["y"], None, dict(cfg, kv_compress=None, qk_norm=False))
        assert (out - plain).abs().max() > 1e-2, name   # the branch really changes the result in the fixture


def test_dit_3d_mask_is_additive_not_dropping():
    """diffusers semantics the CLI triggers (inference.py:274-277): a [B,1,L] float mask is added to the logits."""
    fx = load("dit_small.npz")
    _, dsd, cfg = _dit_small()
    lat, y, mask = fx["lat16"], fx["y"], fx["mask"]
    a = odit.dit_forward(dsd, lat, 400.0, y, mask[:, None, :], cfg)
    b = odit.dit_forward(dsd, lat, 400.0, y, None, cfg)
    c = odit.dit_forward(dsd, lat, 400.0, y, mask, cfg)
    assert not torch.allclose(a, b, atol=1e-5) and not torch.allclose(a, c, atol=1e-5)


T5_SMALL = dict(d_model=128, d_kv=32, num_heads=4, d_ff=256, num_layers=2, vocab_size=100)


def _t5_small():
    from oracle import t5 as ot5
    sd = det_state_dict(ot5.state_dict_shapes(T5_SMALL), seed=707)
    sd["shared.weight"] = sd["shared.weight"] * 8.0
    for k in sd:  # as tests/golden/make_golden.py::golden_t5: q carries the 1/sqrt(d_kv) T5 folds into its initialisation
        if k.endswith("SelfAttention.q.weight"):
            sd[k] = sd[k] * T5_SMALL["d_kv"] ** -0.5
    return sd


def test_t5_encoder_matches_transformers():
    """the prompt producer's T5 v1.1 encoder (diffusion/model/t5.py:82-101 -> transformers.T5EncoderModel)
    against outputs of the installed transformers package on a reduced configuration."""
    from oracle import t5 as ot5
    fx = load("t5_small.npz")
    sd = _t5_small()
    assert abs(checksum(sd) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
    ids, mask = torch.from_numpy(fx["ids"]).long(), torch.from_numpy(fx["mask"]).long()
    torch.testing.assert_close(ot5.t5_encode(sd, ids, mask, T5_SMALL), fx["out"], rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(ot5.t5_encode(sd, ids, None, T5_SMALL), fx["out_nomask"], rtol=2e-4, atol=2e-5)
    # bucket function spot checks (bidirectional, 32 buckets, max distance 128): exact up to +-7, log-spaced beyond, saturating
    b = ot5.relative_position_bucket(torch.tensor([0, 1, -1, 7, -7, 8, 127, 128, 1000, -1000]))
    assert b.tolist() == [0, 17, 1, 23, 7, 24, 31, 31, 31, 15]


def test_glue_matches_reference():
    fx = load("glue.npz")
    acp = oglue.alphas_cumprod()
    assert abs(float(acp[400]) - float(fx["acp400"])) < 1e-7
    assert abs(float(oglue.alphas_cumprod_diffusers()[400]) - float(fx["acp400"])) < 2e-7
    mu = oglue.eps_to_mu(acp, fx["eps_eps"], fx["eps_x"], torch.full((1,), 400).long())
    torch.testing.assert_close(mu, fx["eps_mu"], rtol=1e-6, atol=1e-6)
    for key, v in fx.items():
        if key.startswith("win_"):
            h, w, t, s = map(int, key.split("_")[1:])
            assert np.array_equal(np.array(oglue.sliding_windows(h, w, t, s), dtype=np.int64), v), key
    torch.testing.assert_close(oglue.wavelet_reconstruction(fx["cf_content"], fx["cf_style"]), fx["cf_wavelet"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(oglue.adaptive_instance_normalization(fx["cf_content"], fx["cf_style"]), fx["cf_adain"], rtol=1e-5, atol=1e-6)
    from PIL import Image
    rs = oglue.auto_resize(Image.fromarray(fx["resize_in"]), 64)
    assert np.array_equal(np.array(rs), fx["resize_out"])
    assert np.array_equal(oglue.pad(np.array(rs), 64), fx["pad_out"])


def test_sliding_windows_cover_and_snap():
    for h, w, t, s in [(64, 64, 64, 56), (272, 480, 64, 56), (70, 100, 64, 56)]:
        wins = oglue.sliding_windows(h, w, t, s)
        cover = np.zeros((h, w), int)
        for a, b, c, d in wins:
            assert 0 <= a and b <= h and 0 <= c and d <= w and b - a == t and d - c == t
            cover[a:b, c:d] += 1
        assert cover.min() >= 1


# ---------------------------------------------------------------------------------------------------------------------
# Fixtures added in round 2: unit blocks, released-width blocks, position tables, the reference's process().
def test_vae_unit_blocks_match_reference():
    """ResnetBlock (with nin_shortcut), AttnBlock, Downsample, Upsample of ldm/modules/diffusionmodules/model.py:52-205, one at a time."""
    import torch.nn.functional as F
    fx = load("units.npz")
    shapes = {}
    # same insertion order as make_golden.golden_units builds `shapes` in: det_state_dict seeds per tensor by position
    for a, shp in (("norm1", (32,)), ("conv1", (64, 32, 3, 3)), ("norm2", (64,)), ("conv2", (64, 64, 3, 3)), ("conv_shortcut", (64, 32, 1, 1))):
        for t in ("weight", "bias"):
            shapes[f"r.{a}.{t}"] = shp if t == "weight" else (shp[0],)
    for a in ("group_norm", "to_q", "to_k", "to_v", "to_out.0"):
        for t in ("weight", "bias"):
            shapes[f"a.{a}.{t}"] = (64,) if (a == "group_norm" or t == "bias") else (64, 64)
    shapes.update({"d.conv.weight": (64, 64, 3, 3), "d.conv.bias": (64,), "u.conv.weight": (64, 64, 3, 3), "u.conv.bias": (64,)})
    sd = det_state_dict(shapes, seed=808)
    assert abs(checksum(sd) - float(fx["vae_wsum"])) < 1e-6 * abs(float(fx["vae_wsum"]))
    torch.testing.assert_close(ovae._resnet(sd, "r", fx["res_in"]), fx["res_out"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(ovae._attn(sd, "a", fx["attn_in"]), fx["attn_out"], rtol=1e-4, atol=2e-5)
    x = fx["attn_in"]
    torch.testing.assert_close(ovae._conv(sd, "d.conv", F.pad(x, (0, 1, 0, 1)), stride=2, padding=0), fx["down_out"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(ovae._conv(sd, "u.conv", F.interpolate(x, scale_factor=2.0, mode="nearest")), fx["up_out"], rtol=1e-4, atol=2e-5)


def test_swin_block_full_width_matches_reference():
    """One SwinTransformerBlock at the released width (dim 180, 6 heads, window 8, shift 4) on a 16 x 24 grid (swinir.py:175-290)."""
    fx = load("units.npz")
    C, hid = 180, 360
    shapes = {"norm1.weight": (C,), "norm1.bias": (C,), "attn.relative_position_bias_table": (225, 6), "attn.qkv.weight": (3 * C, C),
              "attn.qkv.bias": (3 * C,), "attn.proj.weight": (C, C), "attn.proj.bias": (C,), "norm2.weight": (C,), "norm2.bias": (C,),
              "mlp.fc1.weight": (hid, C), "mlp.fc1.bias": (hid,), "mlp.fc2.weight": (C, hid), "mlp.fc2.bias": (C,)}
    sd = det_state_dict(shapes, seed=809)
    assert abs(checksum(sd) - float(fx["swin_wsum"])) < 1e-6 * abs(float(fx["swin_wsum"])), "key order differs from the reference module's"
    out = oswin._block(sd, "", fx["swin_in"], 16, 24, 6, 8, 4, oswin._relative_position_index(8))
    torch.testing.assert_close(out, fx["swin_out"], rtol=1e-4, atol=2e-5)


def test_dit_block_released_width_matches_reference():
    """A PixArtMS of depth 1 at hidden size 1152 / 16 heads of 72 (PixArtMS.py:22-79): the released-width block inside its model."""
    fx = load("units.npz")
    sd = det_state_dict(_pixart_shapes(1, 1152, 64), seed=810)
    assert abs(checksum(sd) - float(fx["dit_wsum"])) < 1e-6 * abs(float(fx["dit_wsum"]))
    dsd = odit.pixart_to_diffusers(sd, 1)
    assert abs(checksum(dsd) - float(fx["dit_wsum_diffusers"])) < 1e-6 * abs(float(fx["dit_wsum_diffusers"]))
    cfg = dict(num_layers=1, num_attention_heads=16, attention_head_dim=72, sample_size=16, caption_channels=64)
    torch.testing.assert_close(odit.dit_forward(dsd, fx["dit_lat"], 400.0, fx["dit_y"], None, cfg), fx["dit_out"], rtol=2e-4, atol=3e-5)


def test_sincos_tables_match_reference():
    fx = load("units.npz")
    for gh, gw in ((32, 32), (32, 48)):
        tab = odit.sincos_pos_embed(1152, (gh, gw), 32, 1.0)
        assert tab.shape == (gh * gw, 1152)
        np.testing.assert_allclose(tab[::37], fx[f"pos_{gh}x{gw}_rows"].numpy() if hasattr(fx[f"pos_{gh}x{gw}_rows"], "numpy") else fx[f"pos_{gh}x{gw}_rows"], rtol=0, atol=1e-12)
        assert abs(float(tab.astype(np.float64).sum()) - float(fx[f"pos_{gh}x{gw}_sum"])) < 1e-6


def _process_small_models():
    from tests.golden._det import det_input  # noqa: F401
    sws = det_state_dict(oswin.state_dict_shapes(dict(embed_dim=60, depths=[2, 2], num_heads=[6, 6])), seed=101)
    svae = det_state_dict(ovae.state_dict_shapes(dict(ch=32)), seed=202)
    sdit = det_state_dict(_pixart_shapes(2, 288, 64), seed=303)
    return sws, svae, sdit, odit.pixart_to_diffusers(sdit, 2)


PROCESS_CASES = {"untiled": ("img_small", dict(color_fix_type="wavelet", tiled=False)),
                 "nopre": ("img_small", dict(color_fix_type="wavelet", tiled=False, disable_preprocess_model=True)),
                 "tiled_wavelet": ("img_big", dict(color_fix_type="wavelet", tiled=True, tile_size=64, tile_stride=40)),
                 "tiled_adain": ("img_big", dict(color_fix_type="adain", tiled=True, tile_size=64, tile_stride=40)),
                 "tiled_none": ("img_big", dict(color_fix_type="none", tiled=True, tile_size=64, tile_stride=40))}
PROCESS_DIT_CFG = dict(num_layers=2, num_attention_heads=4, attention_head_dim=72, sample_size=16, caption_channels=64)


@pytest.mark.parametrize("case", sorted(PROCESS_CASES))
def test_oracle_process_matches_reference_process(case):
    """oracle/glue.py::process against the output of the reference's own process() (test_scripts/inference.py:55-166, executed from the
    reference file on reference modules by make_golden.golden_process): untiled, --disable_preprocess_model, and --tiled with snapped
    last tiles (latent 24 x 32, tile 8, stride 5) under all three colour-fix modes. The uint8 results must agree exactly except for the
    pixels whose pre-truncation value sits within float rounding of an integer (<= 0.05 % of the values, off by one grey level; measured <= 0.011 %)."""
    fx = np.load(os.path.join(G, "process_small.npz"))
    sws, svae, sdit, dsd = _process_small_models()
    for key, sd in (("wsum_swin", sws), ("wsum_vae", svae), ("wsum_dit", sdit), ("wsum_dit_diffusers", dsd)):
        assert abs(checksum(sd) - float(fx[key])) < 1e-6 * abs(float(fx[key])), key
    img_key, kw = PROCESS_CASES[case]
    imgs = list(fx[img_key])[: fx[case + "_pred"].shape[0]]
    y = torch.from_numpy(fx["y"])
    preds, stage1 = oglue.process(imgs, lambda x: oswin.swinir_forward(sws, x, dict(embed_dim=60, depths=[2, 2], num_heads=[6, 6])),
                                  lambda x: ovae.vae_encode_mean(svae, x, dict(ch=32)),
                                  lambda lat, t, yy, mm: odit.dit_forward(dsd, lat, t, yy, mm, PROCESS_DIT_CFG),
                                  lambda z: ovae.vae_decode(svae, z, dict(ch=32)), oglue.alphas_cumprod_diffusers(), y, None, **kw)
    for got, want, name in ((np.stack(preds), fx[case + "_pred"], "pred"), (np.stack(stage1), fx[case + "_stage1"], "stage1")):
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        frac = float((d != 0).mean())
        print(f"{case} {name}: {100 * frac:.4f} % of the uint8 values differ, max {int(d.max())}")
        assert d.max() <= 1 and frac <= 5e-4, (case, name, int(d.max()), frac)


def test_center_crop_and_assets_match_reference():
    from PIL import Image
    from instarevive_amd import utils  # host plumbing of the product, pinned by the same fixture as the oracle's
    fx = np.load(os.path.join(G, "glue.npz"))
    for k, size in (("cc_a", 64), ("cc_b", 32), ("cc_c", 64)):
        got = utils.center_crop_arr(Image.fromarray(fx[k + "_in"]), size)
        assert got.shape == (size, size, 3) and np.array_equal(got, fx[k + "_out"]), k
    rs = utils.auto_resize(Image.fromarray(fx["resize_in"]), 64)
    assert np.array_equal(np.array(rs), fx["resize_out"]) and np.array_equal(utils.pad(np.array(rs), 64), fx["pad_out"])
    ref_inputs = "/root/reference/assets/inputs"   # present in the build container only; the checksums travel, the images do not
    if os.path.isdir(ref_inputs):
        for name in sorted(os.listdir(ref_inputs)):
            im = Image.open(os.path.join(ref_inputs, name)).convert("RGB")
            ar = utils.pad(np.array(utils.auto_resize(im, 512)), 64)
            want = fx["asset_" + name.split(".")[0]]
            assert [im.size[0], im.size[1], ar.shape[0], ar.shape[1], int(ar.astype(np.int64).sum() % (1 << 31))] == want.tolist(), name


CLDM_SMALL = dict(model_channels=64, channel_mult=(1, 2, 4, 4), num_res_blocks=2, attention_resolutions=(4, 2, 1), num_head_channels=32,
                  context_dim=64, in_channels=4, hint_channels=4, out_channels=4)


def cldm_small_weights():
    from oracle import cldm as ocldm
    sd_u = det_state_dict(ocldm.state_dict_shapes(CLDM_SMALL), seed=707)
    sd_c = det_state_dict(ocldm.state_dict_shapes(CLDM_SMALL, control=True), seed=708)
    return sd_u, sd_c


def test_cldm_matches_reference():
    """N4: ControlledUnetModel / ControlNet forward and Reflow_ControlLDM.sample_log / apply_condition_encoder of the reference
    (diffusion/cldm.py:32-292,486-490,568-588) against oracle/cldm.py on the same deterministic weights."""
    from oracle import cldm as ocldm
    fx = load("cldm_small.npz")
    sd_u, sd_c = cldm_small_weights()
    assert abs(checksum(sd_u) - float(fx["wsum_unet"])) < 1e-6 * float(fx["wsum_unet"])
    assert abs(checksum(sd_c) - float(fx["wsum_cnet"])) < 1e-6 * float(fx["wsum_cnet"])
    B = fx["x"].shape[0]
    ctx = fx["context"].expand(B, -1, -1)
    t = torch.full((B,), 999.0)
    torch.testing.assert_close(ocldm.unet_forward(sd_u, fx["x"], t, ctx, None, CLDM_SMALL), fx["unet_alone"], rtol=2e-4, atol=2e-4)
    ctrl = ocldm.controlnet_forward(sd_c, fx["x"], fx["c_latent"], t, ctx, CLDM_SMALL)
    assert len(ctrl) == 13
    stats = torch.stack([torch.stack([c.mean(), c.abs().mean(), c.std()]) for c in ctrl])
    torch.testing.assert_close(stats, fx["control_stats"], rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(ctrl[0], fx["control_0"], rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(ctrl[12], fx["control_12"], rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(ocldm.unet_forward(sd_u, fx["x"], t, ctx, ctrl, CLDM_SMALL), fx["unet_controlled"], rtol=2e-4, atol=2e-4)
    sd = {**{"model.diffusion_model." + k: v for k, v in sd_u.items()}, **{"control_model." + k: v for k, v in sd_c.items()}}
    torch.testing.assert_close(ocldm.reflow_sample(sd, fx["zT"], fx["c_latent"], ctx, CLDM_SMALL), fx["sample"], rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(ocldm.reflow_sample(sd, fx["zT"], None, ctx, CLDM_SMALL), fx["sample_no_control"], rtol=2e-4, atol=2e-4)
    # apply_condition_encoder: mode of the posterior of the encoder copy on control * 2 - 1, times scale_factor
    sd_v = det_state_dict(ovae.state_dict_shapes(dict(ch=32)), seed=202)
    assert abs(checksum(sd_v) - float(fx["wsum_vae"])) < 1e-6 * float(fx["wsum_vae"])
    lat = ovae.vae_encode_mean(sd_v, fx["cond_control"] * 2 - 1, dict(ch=32)) * 0.18215
    torch.testing.assert_close(lat, fx["cond_latent"], rtol=1e-4, atol=2e-5)


def test_diffusers_pins():
    """tests/golden/diffusers_pins.npz does not exist until tools/repin_with_diffusers.py has run on a box with diffusers (/ open_clip): it
    then holds the THIRD PARTY's outputs for the behaviours DESIGN.md section 4 lists as unpinned, and from then on this test pins the
    oracle (and the BPE restatement) against them on every box. Skipped while the file is absent."""
    import os
    import pytest
    path = os.path.join(os.path.dirname(__file__), "golden", "diffusers_pins.npz")
    if not os.path.exists(path):
        pytest.skip("diffusers_pins.npz not generated yet (needs diffusers; see tools/repin_with_diffusers.py)")
    from oracle import dit as odit, vae as ovae
    z = np.load(path, allow_pickle=False)
    if "dit_cfg" in z:
        keys = ("num_layers", "num_attention_heads", "attention_head_dim", "patch_size", "sample_size", "caption_channels")
        cfg = dict(zip(keys, (int(v) for v in z["dit_cfg"])), in_channels=4, out_channels=8, interpolation_scale=1.0)
        sd = {k[len("dit_sd/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("dit_sd/")}
        for name in ("native", "nonnative"):
            for mname in ("none", "2d", "3d"):
                tag = f"dit_{name}_{mname}"
                mask = torch.from_numpy(z[tag + "_mask"]) if tag + "_mask" in z else None
                got = odit.dit_forward(sd, torch.from_numpy(z[tag + "_lat"]), torch.tensor([400.0, 400.0]), torch.from_numpy(z[tag + "_y"]), mask, cfg)
                ref = torch.from_numpy(z[tag + "_out"])
                assert float((got - ref).norm() / ref.norm()) <= 1e-4, tag
    if "vae_x" in z:
        sd = {k[len("vae_sd/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("vae_sd/")}
        cfg = dict(ch=32, ch_mult=(1, 2), num_res_blocks=1)
        zz = torch.from_numpy(z["vae_z"])
        assert float((ovae.vae_encode_mean(sd, torch.from_numpy(z["vae_x"]), cfg) - zz).norm() / zz.norm()) <= 1e-4
        dd = torch.from_numpy(z["vae_dec"])
        assert float((ovae.vae_decode(sd, zz, cfg) - dd).norm() / dd.norm()) <= 1e-4
