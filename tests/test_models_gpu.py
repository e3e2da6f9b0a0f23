"""Stage-level and whole-path parity on the GPU: instarevive_amd (HIP, bf16 storage / fp32 accumulation) against the
oracle (CPU fp32, pinned to the reference by tests/test_oracle_golden.py) on the same seeded inputs and weights.

Tolerances (stated once; round 1 measured the values in brackets, the gates are those plus a margin): activations are stored in bf16
(relative rounding 2^-9) across 10-60 chained kernels, so against the fp32 oracle / the reference fixtures a stage must reach
    SwinIR, DiT (+ ControlNet-Half):  relative L2 <= 0.8 %  [0.23-0.42 %], worst element <= 1.5 % of the output range [0.15-0.78 %]
    VAE encode / decode:              relative L2 <= 2.0 %  [1.0-1.45 %],  worst element <= 1.5 % of the range        [0.5-0.8 %]
    T5 encoder:                       relative L2 <= 1.5 %  [0.6-1.0 %],   worst element <= 1.2 % of the range        [0.3-0.6 %]
and the uint8 end result >= 45 dB PSNR against the oracle's uint8 result [47.7-52.5 dB], the stage-1 image >= 50 dB [53.2-53.7 dB]."""
import os

import numpy as np
import pytest
import torch

from oracle import dit as odit
from oracle import glue as oglue
from oracle import swinir as oswin
from oracle import vae as ovae
from instarevive_amd import _lib as L
from tests.golden._det import det_input, det_state_dict

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def rel_l2(a, b):
    return float((a - b).norm() / b.norm())


TOL_XFMR = dict(l2=0.008, worst=0.015)   # SwinIR, DiT
TOL_VAE = dict(l2=0.02, worst=0.015)
TOL_T5 = dict(l2=0.015, worst=0.012)
PSNR_MIN, PSNR_STAGE1_MIN = 45.0, 50.0


def check(got, ref, what, l2, worst):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    r = rel_l2(got, ref)
    w = float((got - ref).abs().max() / (ref.max() - ref.min()))
    print(f"{what}: rel-L2 {r:.4f}, worst/range {w:.4f}")
    assert r <= l2 and w <= worst, f"{what}: rel-L2 {r:.4f} (<= {l2}), worst/range {w:.4f} (<= {worst})"


SWIN_SMALL = dict(embed_dim=60, depths=[2, 2], num_heads=[6, 6])
VAE_SMALL = dict(ch=32)
DIT_SMALL = dict(num_layers=2, num_attention_heads=4, attention_head_dim=72, sample_size=16, caption_channels=64)


def make_swin(cfg, seed=101):
    from instarevive_amd.models import SwinIR
    sd = det_state_dict(oswin.state_dict_shapes(cfg), seed=seed)
    full = dict(oswin.DEFAULT_CFG, **cfg)
    m = SwinIR(img_size=64, patch_size=1, in_chans=3, embed_dim=full["embed_dim"], depths=full["depths"], num_heads=full["num_heads"], window_size=8,
               mlp_ratio=full["mlp_ratio"], sf=8, img_range=1.0, upsampler="nearest+conv", resi_connection="1conv", unshuffle=True, unshuffle_scale=8)
    m.load_state_dict(sd, strict=False)
    return m.to("cuda"), sd


def make_vae(cfg, seed=202):
    from instarevive_amd.models import AutoencoderKL
    sd = det_state_dict(ovae.state_dict_shapes(cfg), seed=seed)
    ch = cfg["ch"]
    m = AutoencoderKL(block_out_channels=(ch, 2 * ch, 4 * ch, 4 * ch))
    m.load_state_dict(sd, strict=True)
    return m.to("cuda"), sd


def make_dit(cfg, seed=404):
    from instarevive_amd.models import Transformer2DModel
    sd = det_state_dict(odit.state_dict_shapes(cfg), seed=seed)
    full = dict(odit.DEFAULT_CFG, **cfg)
    m = Transformer2DModel(num_attention_heads=full["num_attention_heads"], attention_head_dim=full["attention_head_dim"], num_layers=full["num_layers"],
                           sample_size=full["sample_size"], caption_channels=full["caption_channels"],
                           cross_attention_dim=full["num_attention_heads"] * full["attention_head_dim"])
    m.load_state_dict(sd, strict=True)
    return m.to("cuda"), sd


def test_swinir_small_vs_golden_and_oracle():
    fx = np.load(os.path.join(G, "swinir_small.npz"))
    m, sd = make_swin(SWIN_SMALL)
    for k in ("x64", "x128x192"):
        x = torch.from_numpy(fx[k])
        out = m(x.cuda())
        check(out, torch.from_numpy(fx[k + "_out"]), f"swinir small {k} vs reference fixture", **TOL_XFMR)


def test_swinir_full_arch_64():
    m, sd = make_swin({}, seed=111)
    x = det_input(21, (1, 3, 64, 64))
    check(m(x.cuda()), oswin.swinir_forward(sd, x), "swinir full arch 64x64", **TOL_XFMR)


def test_swinir_qkv_inside_the_mlp_launch_ragged_tokens():
    """A Swin block behind its qkv projection is one launch (swin_block_kernel); blocks 1.. of an RSTB get norm1 and their qkv rows from the previous
    block's launch (weights.pack_swin_qkv_tiles), shifted blocks read the attention mask out of per-window-class bias tables (weights.swin_masked_bias).
    24 x 40 token grid = 960 tokens: three full workgroups and a ragged one of 192; two RSTBs of three blocks so that a fused and an
    unfused first block alternate. Against the fp32 oracle and against the plain kernel set (separate LayerNorm / qkv GEMM launches)."""
    cfg = dict(depths=[3, 3], num_heads=[6, 6])
    m, sd = make_swin(cfg, seed=141)
    x = det_input(25, (1, 3, 192, 320))
    fast = m(x.cuda())
    check(fast, oswin.swinir_forward(sd, x, cfg), "swinir 192x320, qkv inside the MLP launch", **TOL_XFMR)
    ctx = m.ctx
    assert ctx.has("swin.l0.b1.qkv_t") and ctx.has("swin.l1.b2.qkv_t") and not ctx.has("swin.l0.b0.qkv_t")
    assert ctx.has("swin.l0.b1.biasM") and not ctx.has("swin.l0.b0.biasM")   # shifted blocks: the mask folded into four bias tables (3 x 5 windows: all classes occur)
    ctx.check(ctx.lib.ir_set_plain_kernels(ctx.h, 1), "ir_set_plain_kernels")
    try:
        plain = m(x.cuda())
    finally:
        ctx.check(ctx.lib.ir_set_plain_kernels(ctx.h, 0), "ir_set_plain_kernels")
    check(fast, plain.float().cpu(), "swinir 192x320 fused vs plain kernels", **TOL_XFMR)


@pytest.mark.parametrize("cfg", [dict(embed_dim=64, depths=[2], num_heads=[4]),                 # Cp = 128: no fused MLP / attention+proj form
                                 dict(embed_dim=128, depths=[2], num_heads=[8]),                # Cp = 256
                                 dict(embed_dim=180, depths=[2], num_heads=[6], mlp_ratio=4)])  # Cp = 192 but 736 hidden units > 512: unfused MLP
def test_swinir_other_widths_take_the_unfused_kernels(cfg):
    """SwinIR configurations the fused Swin kernels do not cover (other head counts, mlp_ratio 4) must load and run through the
    separate LayerNorm / linear / window-attention launches (weights.pack_swinir emits the fused forms for 6 x 32 channels, <= 512 hidden only)."""
    m, sd = make_swin(cfg, seed=131)
    x = det_input(24, (1, 3, 64, 128))
    check(m(x.cuda()), oswin.swinir_forward(sd, x, cfg), f"swinir {cfg}", **TOL_XFMR)


def test_vae_small_vs_golden():
    fx = np.load(os.path.join(G, "vae_small.npz"))
    m, sd = make_vae(VAE_SMALL)
    for k in ("x64", "x64x128"):
        check(m.encode(torch.from_numpy(fx[k]).cuda()).latent_dist.mode(), torch.from_numpy(fx[k + "_mean"]), f"vae encode {k} vs reference fixture", **TOL_VAE)
    for k in ("z8", "z8x16"):
        check(m.decode(torch.from_numpy(fx[k]).cuda()).sample, torch.from_numpy(fx[k + "_dec"]), f"vae decode {k} vs reference fixture", **TOL_VAE)


def test_optional_weight_forms_do_not_leak_between_models():
    """The context's tensor table is keyed by name: a 64-channel VAE leaves "vae.dec.up1.us.wup" (the phase form of its 128 -> 128 upsampling conv)
    behind, and a 32-channel VAE loaded next has a 64 -> 64 conv under that name whose packer makes no phase form. Before round 4's
    ir_drop_optional / exact-size binding the second model bound the first one's matrices (caught by the full suite's test order only)."""
    big, _ = make_vae(dict(ch=64), seed=77)
    z = det_input(31, (1, 4, 8, 8), -3, 3)
    big.decode(z.cuda())
    small, sd = make_vae(VAE_SMALL, seed=78)
    check(small.decode(z.cuda()).sample, ovae.vae_decode(sd, z, VAE_SMALL), "32-channel VAE decoded after a 64-channel one", **TOL_VAE)


def test_vae_full_arch_128():
    m, sd = make_vae(dict(ch=128), seed=222)
    x = det_input(22, (1, 3, 128, 128), -1, 1)
    check(m.encode(x.cuda()).latent_dist.mode(), ovae.vae_encode_mean(sd, x), "vae full encode 128", **TOL_VAE)
    z = det_input(23, (1, 4, 16, 16), -3, 3)
    check(m.decode(z.cuda()).sample, ovae.vae_decode(sd, z), "vae full decode 16->128", **TOL_VAE)


def _prompt(cfg, ntok=20, valid=13, seed=9):
    y = det_input(seed, (1, ntok, cfg["caption_channels"]), -1, 1)
    mask = torch.zeros(1, 1, ntok)
    mask[..., :valid] = 1
    return y, mask


def test_dit_small_all_mask_forms():
    m, sd = make_dit(DIT_SMALL)
    y, mask3 = _prompt(DIT_SMALL)
    for shape in ((1, 4, 16, 16), (2, 4, 16, 24)):
        lat = det_input(sum(shape), shape, -2, 2)
        for mask in (mask3, mask3[:, 0], None):  # [B,1,L] additive (CLI), [B,L] -> -10000, none
            ref = odit.dit_forward(sd, lat, 400.0, y, mask, DIT_SMALL)
            out = m(lat.cuda(), timestep=torch.full((lat.shape[0],), 400), encoder_hidden_states=y.cuda(),
                    encoder_attention_mask=None if mask is None else mask.cuda(), added_cond_kwargs={"resolution": None, "aspect_ratio": None}).sample
            check(out, ref, f"dit small {shape} mask={'none' if mask is None else mask.ndim}", **TOL_XFMR)


def test_dit_micro_conditioning_vs_oracle():
    """The branch of forward_model the repo used to refuse (generate.py:56-62; VERDICT r05 missing 2): a sample_size-128 model adds the embedded
    latent height / width / aspect ratio to the timestep embedding. HIP path (size embedders through gemv launches when the timestep tables are
    rebuilt, cached per (t, h, w)) against the oracle (pinned to the reference's SizeEmbedder by size_embedder.npz), through __call__ with the
    added_cond_kwargs generate.py builds, through the fused step, and after a change of the latent's shape (the tables must follow)."""
    from instarevive_amd import pipeline as P
    from instarevive_amd.models import DDPMScheduler
    cfg = dict(DIT_SMALL, sample_size=128, interpolation_scale=2.0)   # (diffusers: interpolation_scale = sample_size // 64)
    m, sd = make_dit(cfg)
    y, mask3 = _prompt(cfg)
    off = {k: (torch.zeros_like(v) if ("resolution_embedder.linear_2" in k or "aspect_ratio_embedder.linear_2" in k) else v) for k, v in sd.items()}
    for shape in ((1, 4, 16, 16), (2, 4, 16, 24), (1, 4, 16, 16)):
        lat = det_input(sum(shape) + 1, shape, -2, 2)
        ref = odit.dit_forward(sd, lat, 400.0, y, mask3, cfg)
        base = odit.dit_forward(off, lat, 400.0, y, mask3, cfg)   # the same model with the conditioning switched off
        out = P.forward_model(m, lat.cuda(), torch.tensor([400]), y.cuda(), mask3.cuda())
        check(out, ref[:, :4], f"dit micro-conditioning {shape}", **TOL_XFMR)
        assert rel_l2(ref[:, :4], base[:, :4]) > 0.02, "the fixture weights must make the conditioning visible"
    acp = float(DDPMScheduler().alphas_cumprod[400])
    lat = det_input(77, (1, 4, 24, 16), -2, 2)
    x0 = m.step(lat.cuda(), 400.0, acp, y.cuda(), mask3.cuda())
    eps = odit.dit_forward(sd, lat, 400.0, y, mask3, cfg)[:, :4]
    check(x0, (lat - (1 - acp) ** 0.5 * eps) / acp ** 0.5, "dit micro-conditioning, fused step 24 x 16", **TOL_XFMR)
    with pytest.raises(NotImplementedError):
        m(lat.cuda(), timestep=torch.tensor([400]), encoder_hidden_states=y.cuda(), encoder_attention_mask=mask3.cuda(),
          added_cond_kwargs={"resolution": torch.tensor([[512.0, 512.0]]), "aspect_ratio": torch.tensor([[1.0]])})
    # a model without the branch after one with it on the same context: the size embedders must be gone
    m2, sd2 = make_dit(DIT_SMALL)
    lat = det_input(5, (1, 4, 16, 16), -2, 2)
    check(m2(lat.cuda(), timestep=torch.tensor([400]), encoder_hidden_states=y.cuda(), encoder_attention_mask=mask3.cuda()).sample,
          odit.dit_forward(sd2, lat, 400.0, y, mask3, DIT_SMALL), "dit without micro-conditioning after one with", **TOL_XFMR)


@pytest.mark.parametrize("variant", ["conv", "uniform_qknorm", "ave"])
def test_dit_kv_compression_and_qk_norm(variant):
    """The self-attention branches the repo used to refuse (AttentionKVCompress, PixArt_blocks.py:60-158; VERDICT r05 missing 3): KV token compression
    ('conv': depthwise 2 x 2 / stride 2 + LayerNorm; 'uniform' / 'ave': every second token row / column) and LayerNorm on q and k. (a) The reference's
    own PixArtMS outputs (dit_kvc_small.npz, make_golden_r6.py) through the HIP path with the converted weights; (b) a 4 x 72 model on a 32 x 48 latent
    (384 queries against 96 keys per layer that compresses) against the oracle, through __call__ and the fused step."""
    from instarevive_amd.models import DDPMScheduler, Transformer2DModel
    from tests.test_oracle_golden import KVC_VARIANTS, _dit_kvc_small
    fx = np.load(os.path.join(G, "dit_kvc_small.npz"))
    kvc, qkn = KVC_VARIANTS[variant]
    sd, dsd, cfg = _dit_kvc_small(variant)
    m = Transformer2DModel(num_attention_heads=cfg["num_attention_heads"], attention_head_dim=cfg["attention_head_dim"], num_layers=cfg["num_layers"],
                           sample_size=cfg["sample_size"], caption_channels=cfg["caption_channels"], cross_attention_dim=288, kv_compress_config=kvc, qk_norm=qkn)
    m.load_state_dict(dsd, strict=True)
    m.to("cuda")
    lat, y = torch.from_numpy(fx["lat"]), torch.from_numpy(fx["y"])[None]
    out = m(lat.cuda(), timestep=torch.full((2,), 400), encoder_hidden_states=y.cuda()).sample
    check(out, torch.from_numpy(fx["out_" + variant]), f"dit {variant} vs the reference's PixArtMS", **TOL_XFMR)
    # (b) the bench's head width
    cfg2 = dict(DIT_SMALL, qk_norm=qkn, kv_compress=dict(sampling=kvc["sampling"], scale_factor=2, layers=tuple(kvc["kv_compress_layer"])))
    sd2 = det_state_dict(odit.state_dict_shapes(cfg2), seed=414)
    for k in sd2:
        if k.endswith(("attn1.norm.weight", "attn1.q_norm.weight", "attn1.k_norm.weight")):
            sd2[k] = sd2[k] + 1.0
    m2 = Transformer2DModel(num_attention_heads=4, attention_head_dim=72, num_layers=2, sample_size=16, caption_channels=64, cross_attention_dim=288,
                            kv_compress_config=kvc, qk_norm=qkn)
    m2.load_state_dict(sd2, strict=True)
    m2.to("cuda")
    y2, mask3 = _prompt(cfg2)
    lat2 = det_input(88, (1, 4, 32, 48), -2, 2)
    ref = odit.dit_forward(sd2, lat2, 400.0, y2, mask3, cfg2)
    plain = odit.dit_forward(sd2, lat2, 400.0, y2, mask3, dict(cfg2, kv_compress=None, qk_norm=False))
    out2 = m2(lat2.cuda(), timestep=torch.tensor([400]), encoder_hidden_states=y2.cuda(), encoder_attention_mask=mask3.cuda()).sample
    check(out2, ref, f"dit {variant} 32 x 48 vs oracle", **TOL_XFMR)
    assert rel_l2(ref, plain) > 0.02, "the branch must be visible in the comparison"
    acp = float(DDPMScheduler().alphas_cumprod[400])
    x0 = m2.step(lat2.cuda(), 400.0, acp, y2.cuda(), mask3.cuda())
    check(x0, (lat2 - (1 - acp) ** 0.5 * ref[:, :4]) / acp ** 0.5, f"dit {variant}, fused step", **TOL_XFMR)
    # and a model without the branches on the same context afterwards: the optional tensors must be gone
    m3, sd3 = make_dit(DIT_SMALL)
    lat3 = det_input(5, (1, 4, 16, 16), -2, 2)
    check(m3(lat3.cuda(), timestep=torch.tensor([400]), encoder_hidden_states=y2.cuda(), encoder_attention_mask=mask3.cuda()).sample,
          odit.dit_forward(sd3, lat3, 400.0, y2, mask3, DIT_SMALL), "dit without the branches after one with", **TOL_XFMR)


def test_dit_step_matches_eps_to_mu():
    from instarevive_amd.models import DDPMScheduler
    from instarevive_amd.pipeline import eps_to_mu, forward_model
    m, sd = make_dit(DIT_SMALL)
    y, mask3 = _prompt(DIT_SMALL)
    lat = det_input(31, (1, 4, 16, 16), -2, 2).cuda()
    sch = DDPMScheduler()
    t = torch.full((1,), 400).long()
    eps = forward_model(m, lat, t, y.cuda(), mask3.cuda())
    want = eps_to_mu(sch, eps, lat, t.cuda())
    got = m.step(lat, 400.0, float(sch.alphas_cumprod[400]), y.cuda(), mask3.cuda())
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def make_dit_control(cfg, ncopy, seed=505):
    """Base DiT + ControlTransformerHalf, both loaded through the wrapper's state-dict layout (base_model.* / controlnet.*)."""
    from instarevive_amd.models import ControlTransformerHalf
    base, _ = make_dit(cfg)
    sd = det_state_dict(dict({"base_model." + k: v for k, v in odit.state_dict_shapes(cfg).items()},
                             **odit.control_state_dict_shapes(cfg, copy_blocks_num=ncopy)), seed=seed)
    m = ControlTransformerHalf(base, copy_blocks_num=ncopy)
    m.load_state_dict(sd, strict=True)
    flat = {(k[len("base_model."):] if k.startswith("base_model.") else k): v for k, v in sd.items()}
    return m, flat


def test_dit_control_vs_reference_fixture():
    """the HIP ControlNet-Half step against the output of the reference's ControlPixArtHalf."""
    from instarevive_amd.models import ControlTransformerHalf, Transformer2DModel
    from tests.test_oracle_golden import _dit_control_small
    fx = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "dit_control_small.npz")).items() if v.dtype.kind == "f"}
    _, dsd, cfg = _dit_control_small()
    base = Transformer2DModel(num_attention_heads=cfg["num_attention_heads"], attention_head_dim=cfg["attention_head_dim"],
                              num_layers=cfg["num_layers"], sample_size=16, caption_channels=cfg["caption_channels"], cross_attention_dim=288)
    m = ControlTransformerHalf(base, copy_blocks_num=cfg["copy_blocks_num"])
    m.load_state_dict({("base_model." + k if not k.startswith("controlnet.") else k): v for k, v in dsd.items()}, strict=True)
    m.to("cuda")
    y = fx["y"][None]
    kw = dict(timestep=torch.full((1,), 400), encoder_hidden_states=y.cuda(), added_cond_kwargs={"resolution": None, "aspect_ratio": None})
    check(m(fx["lat"].cuda(), c=fx["c"].cuda(), **kw), fx["out_c"], "control dit vs reference fixture", **TOL_XFMR)
    check(base(fx["lat"].cuda(), **kw).sample, fx["out_0"], "base dit of the control fixture, c=None", **TOL_XFMR)


def test_dit_control_small_vs_oracle():
    from instarevive_amd.models import DDPMScheduler
    from instarevive_amd.pipeline import eps_to_mu, forward_model, generate_sample_1step
    cfg = dict(DIT_SMALL, num_layers=4)
    m, sd = make_dit_control(cfg, 2)
    y, mask3 = _prompt(cfg)
    for shape in ((1, 4, 16, 16), (2, 4, 16, 24)):
        lat, c = det_input(sum(shape), shape, -2, 2), det_input(sum(shape) + 1, shape, -2, 2)
        ref = odit.dit_forward(sd, lat, 400.0, y, mask3, dict(cfg, copy_blocks_num=2), c=c)
        out = m(lat.cuda(), timestep=torch.full((lat.shape[0],), 400), encoder_hidden_states=y.cuda(), encoder_attention_mask=mask3.cuda(),
                added_cond_kwargs={"resolution": None, "aspect_ratio": None}, c=c.cuda())
        check(out, ref, f"control dit small {shape}", **TOL_XFMR)
        assert rel_l2(out.cpu(), odit.dit_forward(sd, lat, 400.0, y, mask3, cfg)) > 0.05  # the branch is not a no-op here
    # fused step == forward_model + eps_to_mu through the reference's hook (generate.py:22-51 with c)
    sch, t = DDPMScheduler(), torch.full((1,), 400).long()
    lat, c = det_input(41, (1, 4, 16, 16), -2, 2).cuda(), det_input(42, (1, 4, 16, 16), -2, 2).cuda()
    want = eps_to_mu(sch, forward_model(m, lat, t, y.cuda(), mask3.cuda(), c=c), lat, t.cuda())
    got = generate_sample_1step(m, sch, lat, 400, y.cuda(), mask3.cuda(), c=c)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        m(lat, timestep=t, encoder_hidden_states=y.cuda(), c=None)


def test_dit_control_zero_init_is_the_base_model():
    """A freshly wrapped model (zero before/after projections, transformer_controlnet.py:33-39) must reproduce its base model."""
    from instarevive_amd.models import ControlTransformerHalf
    cfg = dict(DIT_SMALL, num_layers=3)
    base, _ = make_dit(cfg)
    m = ControlTransformerHalf(base, copy_blocks_num=2)
    assert set(m.state_dict()) == {"base_model." + k for k in odit.state_dict_shapes(cfg)} | set(odit.control_state_dict_shapes(cfg, copy_blocks_num=2))
    y, mask3 = _prompt(cfg)
    lat, c = det_input(51, (1, 4, 16, 16), -2, 2).cuda(), det_input(52, (1, 4, 16, 16), -2, 2).cuda()
    kw = dict(timestep=torch.full((1,), 400), encoder_hidden_states=y.cuda(), encoder_attention_mask=mask3.cuda())
    torch.testing.assert_close(m(lat, c=c, **kw), base(lat, **kw).sample, rtol=0, atol=0)
    with pytest.raises(ValueError):
        ControlTransformerHalf(base, copy_blocks_num=3)


def test_dit_control_full_arch_256_tokens():
    """28 blocks + 13 copies at the released width (1152, 16 heads of 72), 32x32 latent: the kernels the 512 px path runs."""
    m, sd = make_dit_control(dict(caption_channels=64), 13, seed=606)
    y, mask3 = _prompt(dict(caption_channels=64), ntok=40, valid=29)
    lat, c = det_input(61, (1, 4, 32, 32), -2, 2), det_input(62, (1, 4, 32, 32), -2, 2)
    ref = odit.dit_forward(sd, lat, 400.0, y, mask3, dict(caption_channels=64), c=c)
    out = m(lat.cuda(), timestep=torch.full((1,), 400), encoder_hidden_states=y.cuda(), encoder_attention_mask=mask3.cuda(), c=c.cuda())
    check(out, ref, "control dit full arch 32x32", **TOL_XFMR)


def make_t5(cfg, seed=707, embed_gain=8.0):
    from instarevive_amd.models import T5EncoderModel
    from oracle import t5 as ot5
    sd = det_state_dict(ot5.state_dict_shapes(cfg), seed=seed)
    sd["shared.weight"] = sd["shared.weight"] * embed_gain
    for k in sd:  # q carries the 1/sqrt(d_kv) that T5 folds into its initialisation: logits of a realistic size
        if k.endswith("SelfAttention.q.weight"):
            sd[k] = sd[k] * cfg["d_kv"] ** -0.5
    m = T5EncoderModel(**cfg)
    m.load_state_dict(sd, strict=True)
    return m.to("cuda"), sd


def test_t5_encoder_vs_transformers_fixture_and_oracle():
    """SURVEY.md section 8(f) N3, the prompt producer's text encoder: the HIP path against the output of transformers.T5EncoderModel
    (fixture) and against the oracle at widths nearer the real model (64 heads of 64 would be 4096 wide; here 8 x 64 and 300 tokens)."""
    from oracle import t5 as ot5
    from tests.test_oracle_golden import T5_SMALL
    fx = np.load(os.path.join(G, "t5_small.npz"))
    m, sd = make_t5(T5_SMALL)
    ids, mask = torch.from_numpy(fx["ids"]), torch.from_numpy(fx["mask"])
    out = m(input_ids=ids.cuda(), attention_mask=mask.cuda())["last_hidden_state"]
    check(out, torch.from_numpy(fx["out"]), "t5 small vs transformers fixture (padding mask)", **TOL_T5)
    check(m(input_ids=ids.cuda())["last_hidden_state"], torch.from_numpy(fx["out_nomask"]), "t5 small vs transformers fixture (no mask)", **TOL_T5)
    cfg = dict(d_model=512, d_kv=64, num_heads=8, d_ff=1024, num_layers=3, vocab_size=500)
    m, sd = make_t5(cfg, seed=808)
    g = np.random.Generator(np.random.PCG64(5))
    for b, t, valid in ((1, 300, 41), (2, 120, 120)):   # the prompt file's 300 tokens; the reference default max_length 120, batch 2
        ids = torch.from_numpy(g.integers(1, 500, size=(b, t)))
        mask = torch.zeros(b, t, dtype=torch.long)
        mask[:, :valid] = 1
        ref = ot5.t5_encode(sd, ids, mask, cfg)
        out = m(input_ids=ids.cuda(), attention_mask=mask.cuda())["last_hidden_state"]
        check(out[:, :valid], ref[:, :valid], f"t5 512-wide b{b} t{t} (valid tokens)", **TOL_T5)
        check(out, ref, f"t5 512-wide b{b} t{t} (all positions)", **TOL_T5)
    with pytest.raises(RuntimeError):
        m(input_ids=torch.full((1, 8), 500).cuda())  # id outside the vocabulary


def test_t5_one_block_at_xxl_width():
    """One block at the dimensions of DeepFloyd/t5-v1_1-xxl (4096 wide, 64 heads of 64, d_ff 10240) on the prompt file's 300 tokens:
    the real model's GEMM shapes and attention LDS footprint (24 such blocks make the 4.7 B parameter encoder)."""
    from oracle import t5 as ot5
    cfg = dict(d_model=4096, d_kv=64, num_heads=64, d_ff=10240, num_layers=1, vocab_size=256)
    m, sd = make_t5(cfg, seed=909)
    g = np.random.Generator(np.random.PCG64(6))
    ids = torch.from_numpy(g.integers(1, 256, size=(1, 300)))
    mask = torch.zeros(1, 300, dtype=torch.long)
    mask[:, :25] = 1
    ref = ot5.t5_encode(sd, ids, mask, cfg)
    out = m(input_ids=ids.cuda(), attention_mask=mask.cuda())["last_hidden_state"]
    check(out, ref, "t5 one block at XXL width, 300 tokens", **TOL_T5)


def test_t5_embedder_interface():
    """T5Embedder.get_text_embeddings (t5.py:82-101) with a stand-in tokenizer: max_length padding, mask, embeddings of the padded batch."""
    from instarevive_amd.models import T5Embedder
    from oracle import t5 as ot5
    from tests.test_oracle_golden import T5_SMALL

    class Tok:  # whitespace "tokenizer" with the call signature the reference uses (t5.py:85-93)
        def __call__(self, texts, max_length, padding, truncation, return_attention_mask, add_special_tokens, return_tensors):
            ids = torch.zeros(len(texts), max_length, dtype=torch.long)
            mask = torch.zeros_like(ids)
            for i, t in enumerate(texts):
                w = [2 + (sum(map(ord, x)) % 90) for x in t.split()][: max_length - 1] + [1]  # </s> = 1
                ids[i, : len(w)] = torch.tensor(w)
                mask[i, : len(w)] = 1
            return {"input_ids": ids, "attention_mask": mask}

    m, sd = make_t5(T5_SMALL)
    emb = T5Embedder("cuda", tokenizer=Tok(), model=m, model_max_length=24)
    e, msk = emb.get_text_embeddings(["  A portrait photo of a human FACE ", "4k, highly detailed"])
    assert tuple(e.shape) == (2, 24, 128) and tuple(msk.shape) == (2, 24) and int(msk.sum()) == 8 + 4
    tok = Tok()(["a portrait photo of a human face", "4k, highly detailed"], 24, "max_length", True, True, True, "pt")
    check(e, ot5.t5_encode(sd, tok["input_ids"], tok["attention_mask"], T5_SMALL), "T5Embedder.get_text_embeddings", **TOL_T5)


def _small_models():
    return make_swin(SWIN_SMALL), make_vae(VAE_SMALL), make_dit(DIT_SMALL)


def _oracle_process(imgs, sws, svae, sdit, y, mask3, **kw):
    return oglue.process(imgs, lambda x: oswin.swinir_forward(sws, x, SWIN_SMALL), lambda x: ovae.vae_encode_mean(svae, x, VAE_SMALL),
                         lambda lat, t, yy, mm: odit.dit_forward(sdit, lat, t, yy, mm, DIT_SMALL), lambda z: ovae.vae_decode(svae, z, VAE_SMALL),
                         oglue.alphas_cumprod_diffusers(), y, mask3, **kw)


def _psnr_u8(a, b):
    return float(oglue.psnr(torch.from_numpy(np.stack(a)).permute(0, 3, 1, 2) / 255.0, torch.from_numpy(np.stack(b)).permute(0, 3, 1, 2) / 255.0).min())


@pytest.mark.parametrize("tiled,fix", [(False, "wavelet"), (True, "wavelet"), (True, "adain"), (True, "none")])
def test_process_small_vs_oracle(tiled, fix):
    from instarevive_amd.pipeline import process
    (sw, sws), (vae, svae), (dit, sdit) = _small_models()
    y, mask3 = _prompt(DIT_SMALL)
    h, w = (128, 192) if tiled else (64, 128)
    imgs = [(det_input(40 + i, (h, w, 3)) * 255).numpy().astype(np.uint8) for i in range(2)]
    kw = dict(color_fix_type=fix, tiled=tiled, tile_size=64, tile_stride=32)
    ref, ref1 = _oracle_process(imgs, sws, svae, sdit, y, mask3, **kw)
    for fused in (True, False):
        got, got1 = process(dit, imgs, 1, fix, False, tiled, 64, 32, preprocess_model=sw, vae=vae, y=y.cuda(), y_mask=mask3.cuda(), fused=fused)
        p, p1 = _psnr_u8(got, ref), _psnr_u8(got1, ref1)
        print(f"process tiled={tiled} fix={fix} fused={fused}: PSNR vs oracle {p:.2f} dB (stage-1 {p1:.2f} dB)")
        assert p >= PSNR_MIN and p1 >= PSNR_STAGE1_MIN


@pytest.mark.parametrize("tiled", [False, True])
def test_process_with_control_branch(tiled):
    """process() with a ControlTransformerHalf (c = the scaled LQ latent, per tile when tiled): fused ir_pipeline with
    IR_FLAG_CONTROL_LQ and the staged generate_sample_1step(..., c=) form against the oracle given the same c."""
    from instarevive_amd.pipeline import process
    cfg = dict(DIT_SMALL, num_layers=3)
    (sw, sws), (vae, svae) = make_swin(SWIN_SMALL), make_vae(VAE_SMALL)
    ctl, sd = make_dit_control(cfg, 2)
    y, mask3 = _prompt(cfg)
    h, w = (128, 192) if tiled else (64, 128)
    imgs = [(det_input(70 + i, (h, w, 3)) * 255).numpy().astype(np.uint8) for i in range(2)]
    kw = dict(color_fix_type="wavelet", tiled=tiled, tile_size=64, tile_stride=32)
    ocfg = dict(cfg, copy_blocks_num=2)
    ref, _ = oglue.process(imgs, lambda x: oswin.swinir_forward(sws, x, SWIN_SMALL), lambda x: ovae.vae_encode_mean(svae, x, VAE_SMALL),
                           lambda lat, t, yy, mm: odit.dit_forward(sd, lat, t, yy, mm, ocfg, c=lat), lambda z: ovae.vae_decode(svae, z, VAE_SMALL),
                           oglue.alphas_cumprod_diffusers(), y, mask3, **kw)
    plain, _ = oglue.process(imgs, lambda x: oswin.swinir_forward(sws, x, SWIN_SMALL), lambda x: ovae.vae_encode_mean(svae, x, VAE_SMALL),
                             lambda lat, t, yy, mm: odit.dit_forward(sd, lat, t, yy, mm, cfg), lambda z: ovae.vae_decode(svae, z, VAE_SMALL),
                             oglue.alphas_cumprod_diffusers(), y, mask3, **kw)
    for fused in (True, False):
        got, _ = process(ctl, imgs, 1, "wavelet", False, tiled, 64, 32, preprocess_model=sw, vae=vae, y=y.cuda(), y_mask=mask3.cuda(), fused=fused)
        p, q = _psnr_u8(got, ref), _psnr_u8(got, plain)
        print(f"process+control tiled={tiled} fused={fused}: PSNR vs oracle with c {p:.2f} dB, vs oracle without c {q:.2f} dB")
        assert p >= PSNR_MIN and q < p - 3.0  # matches the conditioned oracle, and the branch visibly changes the image


@pytest.mark.parametrize("tiled", [False, True])
def test_process_hipgraph_replay_is_identical(tiled):
    """IR_FLAG_GRAPH (BASELINE configs[2], "hipGraph-captured per-tile step"): recording, replaying on new pixels through the same
    buffers, and re-recording after the prompt changed must all be bit-identical to the plain launch sequence."""
    from instarevive_amd.pipeline import process
    (sw, _), (vae, _), (dit, _) = _small_models()
    y, mask3 = _prompt(DIT_SMALL)
    h, w = (128, 192) if tiled else (64, 128)
    kw = dict(preprocess_model=sw, vae=vae, y_mask=mask3.cuda())
    args = (1, "wavelet", False, tiled, 64, 32)
    for rnd, yy in enumerate((y.cuda(), (y * 0.5).cuda())):   # the second prompt invalidates the recorded graphs (ir_dit_set_prompt)
        for i in range(3):                    # i = 0 records (first round) / re-records (second round), i > 0 replays
            imgs = [(det_input(90 + 10 * rnd + i, (h, w, 3)) * 255).numpy().astype(np.uint8)]
            want, want1 = process(dit, imgs, *args, y=yy, **kw)
            got, got1 = process(dit, imgs, *args, y=yy, graph=True, **kw)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got1[0], want1[0]), (rnd, i)
    assert len({a.tobytes() for a in (want[0],)}) == 1 and want[0].std() > 1  # a real image, not zeros


def test_process_disable_preprocess():
    from instarevive_amd.pipeline import process
    (sw, sws), (vae, svae), (dit, sdit) = _small_models()
    y, mask3 = _prompt(DIT_SMALL)
    imgs = [(det_input(50, (64, 64, 3)) * 255).numpy().astype(np.uint8)]
    ref, ref1 = _oracle_process(imgs, sws, svae, sdit, y, mask3, disable_preprocess_model=True)
    got, got1 = process(dit, imgs, 1, "wavelet", True, False, 512, 448, preprocess_model=None, vae=vae, y=y.cuda(), y_mask=mask3.cuda())
    assert np.array_equal(got1[0], imgs[0]) and np.array_equal(ref1[0], imgs[0])  # stage-1 == LQ input, bit exact
    assert _psnr_u8(got, ref) >= PSNR_MIN


# ---------------------------------------------------------------------------------------------------------------------
# Full-size architectures (SwinIR 15.8 M, VAE 83.7 M, DiT 611 M parameters, 300 x 4096 prompt), seeded random weights.
def test_full_arch_process_256_vs_oracle(full_models):
    """Whole path at full depth (8x6 Swin blocks, 28 DiT layers, full VAE) on a 256x256 image against the fp32 oracle."""
    import bench
    from instarevive_amd.pipeline import process
    swin, vae, dit, sds, y, mask = full_models
    imgs = [bench.synthetic_lq(1, 256, 256, 5)[0].numpy()]
    ref, ref1, inter = oglue.process(imgs, lambda x: oswin.swinir_forward(sds["swin"], x), lambda x: ovae.vae_encode_mean(sds["vae"], x),
                                     lambda lat, t, yy, mm: odit.dit_forward(sds["dit"], lat, t, yy, mm), lambda z: ovae.vae_decode(sds["vae"], z),
                                     oglue.alphas_cumprod_diffusers(), y, mask, return_intermediates=True)
    got, got1 = process(dit, imgs, 1, "wavelet", False, False, 512, 448, preprocess_model=swin, vae=vae, y=y.cuda(), y_mask=mask.cuda())
    p, p1 = _psnr_u8(got, ref), _psnr_u8(got1, ref1)
    print(f"full-arch 256x256: PSNR vs fp32 oracle {p:.2f} dB (stage-1 {p1:.2f} dB)")
    assert p >= PSNR_MIN and p1 >= PSNR_STAGE1_MIN
    # north_star's acceptance form: PSNR against a ground truth must be within 0.1 dB of the reference path's PSNR against it.
    # Any third image serves as "ground truth" for that comparison; here the LQ input itself.
    gt = [np.asarray(i) for i in imgs]
    d = abs(_psnr_u8(got, gt) - _psnr_u8(ref, gt))
    print(f"full-arch 256x256: |PSNR(ours, GT) - PSNR(oracle, GT)| = {d:.4f} dB")
    assert d <= 0.1
    # stage-level: x0 latent of the fused DiT step against the oracle's
    lat = inter["init_noise"].cuda()
    x0 = dit.step(lat, 400.0, float(oglue.alphas_cumprod_diffusers()[400]), y.cuda(), mask.cuda())
    check(x0, inter["x0"], "full-arch DiT x0 (28 layers)", l2=0.012, worst=0.03)  # measured 0.64 % over 28 layers


def test_full_arch_process_512_vs_oracle(full_models):
    """BASELINE.json configs[0]/[1] network size before the sr_scale: the whole path at full depth on a 512 x 512 image (4096 VAE
    mid-block tokens, 1024 DiT tokens, 64 x 64 Swin windows) against the fp32 oracle - the same pass bench.py times as its CPU
    baseline (about 6 s of host time)."""
    import bench
    from instarevive_amd.pipeline import process
    swin, vae, dit, sds, y, mask = full_models
    imgs = [bench.synthetic_lq(1, 512, 512, 15)[0].numpy()]
    ref, ref1 = oglue.process(imgs, lambda x: oswin.swinir_forward(sds["swin"], x), lambda x: ovae.vae_encode_mean(sds["vae"], x),
                              lambda lat, t, yy, mm: odit.dit_forward(sds["dit"], lat, t, yy, mm), lambda z: ovae.vae_decode(sds["vae"], z),
                              oglue.alphas_cumprod_diffusers(), y, mask)
    got, got1 = process(dit, imgs, 1, "wavelet", False, False, 512, 448, preprocess_model=swin, vae=vae, y=y.cuda(), y_mask=mask.cuda())
    p, p1 = _psnr_u8(got, ref), _psnr_u8(got1, ref1)
    gt = [np.asarray(i) for i in imgs]
    d = abs(_psnr_u8(got, gt) - _psnr_u8(ref, gt))
    print(f"full-arch 512x512: PSNR vs fp32 oracle {p:.2f} dB (stage-1 {p1:.2f} dB); |PSNR(ours, GT) - PSNR(oracle, GT)| = {d:.4f} dB")
    assert p >= PSNR_MIN and p1 >= PSNR_STAGE1_MIN and d <= 0.1


def test_full_arch_size_independent_properties(full_models):
    """At a size the oracle cannot finish quickly (1024x1024): determinism, batch independence, single-tile == untiled."""
    import bench
    from instarevive_amd.pipeline import process
    swin, vae, dit, sds, y, mask = full_models
    a, b = [bench.synthetic_lq(1, 1024, 1024, s)[0].numpy() for s in (11, 12)]
    kw = dict(preprocess_model=swin, vae=vae, y=y.cuda(), y_mask=mask.cuda())
    pa, sa = process(dit, [a], 1, "none", False, False, 512, 448, **kw)
    pa2, _ = process(dit, [a], 1, "none", False, False, 512, 448, **kw)
    assert np.array_equal(pa[0], pa2[0]), "the path must be deterministic run to run"
    pab, _ = process(dit, [a, b], 1, "none", False, False, 512, 448, **kw)
    assert np.array_equal(pab[0], pa[0]), "an image's result must not depend on its batch neighbours"
    # one 1024-px tile with stride 1024 covers the image exactly once: the tiled code path must reproduce the untiled result
    pt, _ = process(dit, [a], 1, "none", False, True, 1024, 1024, **kw)
    d = np.abs(pt[0].astype(int) - pa[0].astype(int))
    assert d.max() <= 1, f"single-tile tiled vs untiled differ by {d.max()} grey levels"
    assert pa[0].std() > 1.0 and sa[0].std() > 1.0  # not a constant image


def test_full_arch_tiled_2048(full_models):
    """2048x2048 --tiled (25 tiles, wavelet fix): runs, is finite/non-trivial and agrees with the stage-by-stage form."""
    import bench
    from instarevive_amd.pipeline import process
    swin, vae, dit, sds, y, mask = full_models
    a = bench.synthetic_lq(1, 2048, 2048, 21)[0].numpy()
    kw = dict(preprocess_model=swin, vae=vae, y=y.cuda(), y_mask=mask.cuda())
    fused, _ = process(dit, [a], 1, "wavelet", False, True, 512, 448, fused=True, **kw)
    staged, _ = process(dit, [a], 1, "wavelet", False, True, 512, 448, fused=False, **kw)
    p = _psnr_u8(fused, staged)
    print(f"2048 tiled fused vs staged: {p:.2f} dB")
    assert p >= 45.0 and fused[0].std() > 1.0


def test_fast_vs_plain_kernels_at_awkward_sizes(full_models):
    """A padded 1080p frame (1088 x 1920: 8160 DiT tokens, not a multiple of 64 or 256; 16320-row linears) and a 832 x 1216 one through
    process() at full architecture, once with the default kernels and once with the ping-pong / big-tile kernels switched off
    (ir_set_plain_kernels): the same arithmetic through two independent kernel sets must agree (>= 45 dB on the uint8 result)."""
    import bench
    from instarevive_amd.pipeline import process
    swin, vae, dit, sds, y, mask = full_models
    ctx = dit.ctx
    kw = dict(preprocess_model=swin, vae=vae, y=full_models.y_cuda, y_mask=full_models.mask_cuda)
    for h, w in ((1088, 1920), (832, 1216)):
        img = bench.synthetic_lq(1, h, w, 7)[0].numpy()
        fast, _ = process(dit, [img], 1, "wavelet", False, False, 512, 448, **kw)
        ctx.check(ctx.lib.ir_set_plain_kernels(ctx.h, 1), "ir_set_plain_kernels")
        try:
            plain, _ = process(dit, [img], 1, "wavelet", False, False, 512, 448, **kw)
        finally:
            ctx.check(ctx.lib.ir_set_plain_kernels(ctx.h, 0), "ir_set_plain_kernels")
        p = _psnr_u8(fast, plain)
        print(f"{h}x{w}: fast vs plain kernels {p:.2f} dB, output std {fast[0].std():.1f}")
        assert p >= 45.0 and fast[0].std() > 1.0


def test_vae_fp8_resnet_convs(full_models):
    """BASELINE.json configs[4] (fp8 VAE conv weights): the full-size VAE with the ResnetBlock 3x3 convs on fp8 (OCP e4m3) operands —
    weights quantised per output channel, GroupNorm+SiLU outputs written as e4m3 — against the fp32 oracle and against the bf16 path.
    e4m3 keeps 3 mantissa bits (2.6 % relative error per weight tensor): measured rel-L2 0.10 (encoder mean) / 0.06 (decoder), gate 0.12."""
    swin, vae, dit, sds, y, mask = full_models
    x = det_input(22, (1, 3, 128, 128), -1, 1)
    z = det_input(23, (1, 4, 16, 16), -3, 3)
    ref_e, ref_d = ovae.vae_encode_mean(sds["vae"], x), ovae.vae_decode(sds["vae"], z)
    b_e, b_d = vae.encode(x.cuda()).latent_dist.mode().cpu(), vae.decode(z.cuda()).sample.cpu()
    ctx = vae.ctx
    vae.enable_fp8(True)
    ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, 0xffffffff), "ir_set_fp8_mask")   # every conv level of both halves (the default set holds two decoder levels)
    try:
        f_e, f_d = vae.encode(x.cuda()).latent_dist.mode().cpu(), vae.decode(z.cuda()).sample.cpu()
    finally:
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, L.FP8_MASK_DEFAULT), "ir_set_fp8_mask")
        vae.enable_fp8(False)
    again = vae.decode(z.cuda()).sample.cpu()
    assert torch.equal(again, b_d), "switching fp8 off must restore the bf16 path bit for bit"
    for name, f, b, ref in (("encode", f_e, b_e, ref_e), ("decode", f_d, b_d, ref_d)):
        print(f"vae {name}: fp8 vs oracle rel-L2 {rel_l2(f, ref):.4f} (bf16 path {rel_l2(b, ref):.4f}); fp8 vs bf16 {rel_l2(f, b):.4f}")
        assert not torch.equal(f, b), "the fp8 path must actually run"
        assert rel_l2(f, ref) <= 0.12


def test_psnr_guard_bites_at_realistic_reference_quality(full_models):
    """north_star's 0.1 dB criterion where it binds: the reference scoring 25 / 30 / 35 dB against the ground truth.
    No ground truth exists offline, so synthetic ones are built around the oracle's 512 x 512 full-architecture output
    (tests/support/psnr_guard.py): independent white noise at each level, and the worst case (ground truth on the far side of the
    oracle, anti-parallel to our error). Asserted: bf16 within 0.1 dB at 25 and 30 dB (independent). Reported: 35 dB, the worst case, the
    crossing level P_err - 16.33 dB for bf16 and for fp8 (cfg-5), which DESIGN.md section 4 and the fp8 bench line quote. Round 5: the DEFAULT
    fp8 operand set (IR_FP8_MASK_DEFAULT, chosen by this criterion) is asserted like bf16; the set of every part is reported."""
    import bench
    from instarevive_amd.pipeline import process
    from tests.support.psnr_guard import crossing_level, guard_table
    swin, vae, dit, sds, y, mask = full_models
    imgs = [bench.synthetic_lq(1, 512, 512, 15)[0].numpy()]
    ref, _ = oglue.process(imgs, lambda x: oswin.swinir_forward(sds["swin"], x), lambda x: ovae.vae_encode_mean(sds["vae"], x),
                           lambda lat, t, yy, mm: odit.dit_forward(sds["dit"], lat, t, yy, mm), lambda z: ovae.vae_decode(sds["vae"], z),
                           oglue.alphas_cumprod_diffusers(), y, mask)
    kw = dict(preprocess_model=swin, vae=vae, y=full_models.y_cuda, y_mask=full_models.mask_cuda)
    bf, _ = process(dit, imgs, 1, "wavelet", False, False, 512, 448, **kw)
    from instarevive_amd import _lib as L
    ctx = dit.ctx
    vae.enable_fp8(True)
    try:
        f8, _ = process(dit, imgs, 1, "wavelet", False, False, 512, 448, fp8=True, **kw)            # IR_FP8_MASK_DEFAULT
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, L.FP8_MASK_ALL), "ir_set_fp8_mask")
        f8_all, _ = process(dit, imgs, 1, "wavelet", False, False, 512, 448, fp8=True, **kw)
    finally:
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, L.FP8_MASK_DEFAULT), "ir_set_fp8_mask")
        vae.enable_fp8(False)
    pb, rb = guard_table(bf[0], ref[0])
    p8, r8 = guard_table(f8[0], ref[0])
    pa, ra = guard_table(f8_all[0], ref[0])
    for name, pe, rows in (("bf16", pb, rb), ("fp8 default operand set", p8, r8), ("fp8 every part (opt-in, out of tolerance)", pa, ra)):
        txt = ", ".join(f"{lv:.0f} dB: {d[0]:.3f} (worst case {d[1]:.2f})" for lv, d in rows.items())
        print(f"PSNR guard {name}: {pe:.2f} dB vs oracle; |dPSNR| with the reference at {txt}; within 0.1 dB up to a reference quality of {crossing_level(pe):.1f} dB")
    assert rb[25.0][0] <= 0.1 and rb[30.0][0] <= 0.1, "bf16 must stay within 0.1 dB of the reference where the reference scores <= 30 dB"
    # cfg-5: the DEFAULT operand set is held to the same criterion as bf16; the set of every part is reported, not gated on it
    assert r8[25.0][0] <= 0.1 and r8[30.0][0] <= 0.1, "the default fp8 operand set must stay within 0.1 dB at 25 and 30 dB"
    assert p8 >= 46.3 and pa < p8
    # the estimate the bench line quotes must describe what was measured
    for pe, rows in ((pb, rb), (p8, r8), (pa, ra)):
        for lv, d in rows.items():
            assert abs(d[0] - 10 * np.log10(1 + 10 ** ((lv - pe) / 10))) <= 0.03


def test_fp8_mask_selects_the_operand_set(full_models):
    """ir_set_fp8_mask: with every part switched off the fp8 call IS the bf16 path (bit for bit); the unmasked call runs the default set
    (IR_FP8_MASK_DEFAULT = 0x5006: the VAE attention parts + decoder level-0 / level-2 convs; IR_FP8_MASK_QUALIFIED adds the DiT self-attention); the attention parts alone (DiT self-attention +
    both VAE mid blocks) stay within 0.3 dB of the bf16 path's PSNR against the oracle - tools/fp8_attribution.py: the e4m3 conv activations
    carry cfg-5's error, the attention products do not - and the full set is what the unmasked call runs."""
    import bench
    from instarevive_amd.pipeline import process
    swin, vae, dit, sds, y, mask = full_models
    ctx = dit.ctx
    imgs = [bench.synthetic_lq(1, 512, 512, 15)[0].numpy()]
    ref, _ = oglue.process(imgs, lambda x: oswin.swinir_forward(sds["swin"], x), lambda x: ovae.vae_encode_mean(sds["vae"], x),
                           lambda lat, t, yy, mm: odit.dit_forward(sds["dit"], lat, t, yy, mm), lambda z: ovae.vae_decode(sds["vae"], z),
                           oglue.alphas_cumprod_diffusers(), y, mask)
    kw = dict(preprocess_model=swin, vae=vae, y=full_models.y_cuda, y_mask=full_models.mask_cuda)
    bf, _ = process(dit, imgs, 1, "wavelet", False, False, 512, 448, **kw)
    vae.enable_fp8(True)
    try:
        full, _ = process(dit, imgs, 1, "wavelet", False, False, 512, 448, fp8=True, **kw)          # the context's default: IR_FP8_MASK_DEFAULT
        out = {}
        for name, m in (("none", 0), ("attention", 0b111), ("default", L.FP8_MASK_DEFAULT), ("qualified", L.FP8_MASK_QUALIFIED), ("all", 0xffffffff)):
            ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, m), "ir_set_fp8_mask")
            out[name], _ = process(dit, imgs, 1, "wavelet", False, False, 512, 448, fp8=True, **kw)
    finally:
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, L.FP8_MASK_DEFAULT), "ir_set_fp8_mask")
        vae.enable_fp8(False)
    assert L.FP8_MASK_DEFAULT == 0x5006 and L.FP8_MASK_QUALIFIED == 0x5007 and not np.array_equal(out["qualified"][0], full[0])
    assert np.array_equal(out["none"][0], bf[0]), "an empty operand set must be the bf16 path"
    assert np.array_equal(out["default"][0], full[0]) and not np.array_equal(full[0], bf[0]), "the unmasked call runs IR_FP8_MASK_DEFAULT"
    assert not np.array_equal(out["all"][0], full[0])
    pb, pa, pd, pf = _psnr_u8(bf, ref), _psnr_u8(out["attention"], ref), _psnr_u8(full, ref), _psnr_u8(out["all"], ref)
    print(f"fp8 operand sets vs fp32 oracle: bf16 {pb:.2f} dB, attention parts only {pa:.2f} dB, default set {pd:.2f} dB, all parts {pf:.2f} dB")
    assert not np.array_equal(out["attention"][0], bf[0]) and pa >= pb - 0.3 and pf < pd <= pa + 0.05


def test_fp8_tiled_and_hipgraph(full_models):
    """cfg-5 under --tiled: the fp8 attention then runs per batch of 1024-token tiles and the fp8 convs on 512 x 512 tiles (the small levels
    stay with the 4-wave fp8 kernel); tiled fp8 against tiled bf16 on a 1024 x 1024 image, and the hipGraph replay of the fp8 path against
    its plain launches (the prep kernel, the flagged fallback launches and the fp8 convs must all be capturable and re-playable)."""
    import bench
    from instarevive_amd.pipeline import process
    swin, vae, dit, sds, y, mask = full_models
    img = bench.synthetic_lq(1, 1024, 1024, 61)[0].numpy()
    kw = dict(preprocess_model=swin, vae=vae, y=full_models.y_cuda, y_mask=full_models.mask_cuda)
    bf, _ = process(dit, [img], 1, "wavelet", False, True, 512, 448, **kw)
    vae.enable_fp8(True)
    try:
        f8, _ = process(dit, [img], 1, "wavelet", False, True, 512, 448, fp8=True, **kw)
        g1, _ = process(dit, [img], 1, "wavelet", False, True, 512, 448, fp8=True, graph=True, **kw)   # records
        g2, _ = process(dit, [img], 1, "wavelet", False, True, 512, 448, fp8=True, graph=True, **kw)   # replays
    finally:
        vae.enable_fp8(False)
    p = _psnr_u8(f8, bf)
    print(f"fp8 tiled 1024x1024 vs bf16 tiled: {p:.2f} dB")
    assert not np.array_equal(f8[0], bf[0]) and p >= 38.0
    assert np.array_equal(g1[0], f8[0]) and np.array_equal(g2[0], f8[0]), "hipGraph record / replay of the fp8 path must equal its plain launches"
