#!/usr/bin/env python3
"""Close the "parity unpinned" items the first time a box has the third-party packages this image lacks. NOT runnable in the build
container (no diffusers / open_clip / pyiqa, no network); documented so that one command does it where they exist:

    pip install diffusers==0.30.0 open_clip_torch pyiqa          # whichever are available; every section is skipped when its package is not
    python tools/repin_with_diffusers.py [--dit DIR] [--vae DIR] [--clip_tokens]   # writes tests/golden/diffusers_pins.npz, prints a verdict per item

What is unpinned today and how each item is pinned here. All of it is BEHAVIOUR OF CODE, not of weights, so seeded
random-initialised models of reduced width pin it exactly as the 4 GB checkpoints would; --dit / --vae additionally load real folders
(diffusers `from_pretrained` layouts) and run the same comparison at full size.

 1. diffusers Transformer2DModel (norm_type="ada_norm_single"), a 3-D `encoder_attention_mask` is ADDED to the cross-attention logits, a 2-D
    one becomes (1 - m) * -10000 (test_scripts/inference.py:274-277 passes 3-D): outputs for mask = None / 2-D / 3-D -> oracle.dit.dit_forward.
 2. PatchEmbed regenerates the sin-cos table for a latent that is not sample_size x sample_size with base_size = sample_size // patch_size
    (and `interpolation_scale`): a non-native, non-square latent -> oracle.dit.dit_forward / sincos_pos_embed.
 3. AutoencoderKL state-dict key names (to_q / to_k / to_v / to_out.0 against the legacy query / key / value / proj_attn) and the
    encode().latent_dist.mode() / decode().sample contract: a random-initialised AutoencoderKL's state dict -> oracle.vae + weights.pack_vae's
    expected keys.
 4. open_clip's text tower blocks and tokenizer (FrozenOpenCLIPEmbedder, ldm/modules/encoders/modules.py:171-193): a reduced CLIP text tower ->
    oracle.clip_text; open_clip.tokenize on sample prompts -> instarevive_amd.clip_bpe (needs open_clip's vocabulary file, which ships inside the
    open_clip package).
 5. pyiqa's PSNR-Y / SSIM-Y (evaluate_img.py:30-33) -> tools/evaluate_pairs.py.
 7. the `lpips` package's LPIPS(net="alex") (utils/metrics.py:41-66, evaluate_img.py:32) -> tools/evaluate_pairs.py::LPIPS on the package's own weights.
 6. ftfy.fix_text (diffusion/model/t5.py:118-124) -> instarevive_amd.captions.fix_text (deterministic steps + the restricted mojibake repair).

The fixture holds inputs, state-dict checksums and the third party's outputs (data, not source); tests/test_oracle_golden.py picks
tests/golden/diffusers_pins.npz up when it exists (test_diffusers_pins) and compares the oracle against it on every later run, on any box."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "diffusers_pins.npz")


def seeded_(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in sorted(model.named_parameters()):
            p.copy_((torch.rand(p.shape, generator=g) - 0.5) * (0.2 if p.ndim == 1 else 2 * (3.0 / max(1, int(np.prod(p.shape[1:])))) ** 0.5))
    return model


def rel(a, b):
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))


def pin_dit(out, real_dir=None):
    from diffusers import Transformer2DModel
    from oracle import dit as odit
    cfg = dict(num_layers=2, num_attention_heads=2, attention_head_dim=8, in_channels=4, out_channels=8, patch_size=2, sample_size=8,
               caption_channels=32, interpolation_scale=1.0)
    if real_dir:
        m = Transformer2DModel.from_pretrained(real_dir).eval()
        c = m.config
        cfg = dict(num_layers=c.num_layers, num_attention_heads=c.num_attention_heads, attention_head_dim=c.attention_head_dim, in_channels=c.in_channels,
                   out_channels=c.out_channels, patch_size=c.patch_size, sample_size=c.sample_size, caption_channels=c.caption_channels,
                   interpolation_scale=getattr(c, "interpolation_scale", None) or max(c.sample_size // 64, 1))
    else:
        m = seeded_(Transformer2DModel(num_attention_heads=2, attention_head_dim=8, in_channels=4, out_channels=8, num_layers=2, cross_attention_dim=16,
                                       norm_type="ada_norm_single", sample_size=8, patch_size=2, caption_channels=32, norm_elementwise_affine=False,
                                       norm_eps=1e-6, attention_bias=True, activation_fn="gelu-approximate", num_embeds_ada_norm=1000), 11).eval()
    sd = {k: v.detach().float() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    ok = True
    for name, (h, w) in (("native", (cfg["sample_size"],) * 2), ("nonnative", (cfg["sample_size"] + 4, cfg["sample_size"] * 2 + 4))):
        lat = torch.randn(2, 4, h, w, generator=g)
        y = torch.randn(2, 6, cfg["caption_channels"], generator=g) * 0.3
        m2 = torch.tensor([[1, 1, 1, 1, 0, 0], [1, 1, 0, 0, 0, 0]], dtype=torch.float32)
        for mname, mask in (("none", None), ("2d", m2), ("3d", m2[:, None, :])):
            with torch.no_grad():
                ref = m(lat, encoder_hidden_states=y, timestep=torch.tensor([400, 400]), encoder_attention_mask=mask,
                        added_cond_kwargs={"resolution": None, "aspect_ratio": None}).sample
            got = odit.dit_forward(sd, lat, torch.tensor([400.0, 400.0]), y, mask, cfg)
            e = rel(got, ref)
            ok &= e <= 1e-4
            print(f"  [1/2] Transformer2DModel {name} latent {h}x{w}, mask {mname}: oracle vs diffusers rel. L2 {e:.2e}")
            tag = f"dit_{'real_' if real_dir else ''}{name}_{mname}"
            out[tag + "_lat"], out[tag + "_y"], out[tag + "_out"] = lat.numpy(), y.numpy(), ref.numpy()
            if mask is not None:
                out[tag + "_mask"] = mask.numpy()
    if not real_dir:
        out["dit_cfg"] = np.array([cfg[k] for k in ("num_layers", "num_attention_heads", "attention_head_dim", "patch_size", "sample_size", "caption_channels")])
        for k, v in sd.items():
            out["dit_sd/" + k] = v.numpy()
    return ok


def pin_vae(out, real_dir=None):
    from diffusers import AutoencoderKL
    from instarevive_amd import weights as W
    from oracle import vae as ovae
    cfg = dict(ch=32, ch_mult=(1, 2), num_res_blocks=1)
    if real_dir:
        m = AutoencoderKL.from_pretrained(real_dir).eval()
        cfg = None
    else:
        m = seeded_(AutoencoderKL(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 2, up_block_types=("UpDecoderBlock2D",) * 2,
                                  block_out_channels=(32, 64), layers_per_block=1, latent_channels=4, norm_num_groups=32, sample_size=32), 12).eval()
    sd = {k: v.detach().float() for k, v in m.state_dict().items()}
    keys = set(sd)
    want = set(W.vae_shapes(cfg) if cfg else W.vae_shapes(dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2)))
    legacy = {k.replace("to_q", "query").replace("to_k", "key").replace("to_v", "value").replace("to_out.0", "proj_attn") for k in want}
    names_ok = keys == want or keys == legacy
    print(f"  [3] AutoencoderKL state-dict keys: {'match' if names_ok else 'DIFFER'} ({len(keys)} keys; "
          f"{'current to_q/... form' if keys == want else 'legacy query/... form' if keys == legacy else sorted(keys ^ want)[:6]})")
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(6)) * 2 - 1
    with torch.no_grad():
        z_ref = m.encode(x).latent_dist.mode()
        d_ref = m.decode(z_ref).sample
    e1, e2 = rel(ovae.vae_encode_mean(sd, x, cfg), z_ref), rel(ovae.vae_decode(sd, z_ref, cfg), d_ref)
    print(f"  [3] AutoencoderKL encode().latent_dist.mode(): oracle rel. L2 {e1:.2e}; decode().sample: {e2:.2e}")
    tag = "vae_real" if real_dir else "vae"
    out[tag + "_x"], out[tag + "_z"], out[tag + "_dec"] = x.numpy(), z_ref.numpy(), d_ref.numpy()
    if not real_dir:
        out["vae_keys"] = np.array(sorted(keys))
        for k, v in sd.items():
            out["vae_sd/" + k] = v.numpy()
    return names_ok and e1 <= 1e-4 and e2 <= 1e-4


def pin_clip(out):
    import open_clip
    from instarevive_amd.clip_bpe import ClipBPETokenizer
    from oracle import clip_text as oclip
    texts = ["", "a photo of a cat", "High-quality restoration of an OLD photograph, 4k & sharp!", "naïve café — 42 dogs"]
    want = open_clip.tokenize(texts)
    folder = os.path.dirname(open_clip.tokenizer.default_bpe())
    got = ClipBPETokenizer.from_folder(folder)(texts)
    tok_ok = bool(torch.equal(got, want))
    print(f"  [4] open_clip.tokenize vs instarevive_amd.clip_bpe on {len(texts)} prompts: {'identical' if tok_ok else 'DIFFER'}")
    out["clip_tokens"], out["clip_texts"] = want.numpy(), np.array(texts)
    tower_ok = True
    try:
        model = open_clip.model.CLIP(embed_dim=32, vision_cfg=dict(image_size=32, layers=1, width=32, patch_size=16),
                                     text_cfg=dict(context_length=77, vocab_size=49408, width=64, heads=2, layers=3)).eval()
        seeded_(model, 13)
        sd = {k: v.detach().float() for k, v in model.state_dict().items() if not k.startswith("visual.")}
        x = model.token_embedding(want) + model.positional_embedding
        x = x.permute(1, 0, 2)
        for i, r in enumerate(model.transformer.resblocks):
            if i == len(model.transformer.resblocks) - 1:   # layer "penultimate" (modules.py:185-186)
                break
            x = r(x, attn_mask=model.attn_mask)
        ref = model.ln_final(x.permute(1, 0, 2))
        got = oclip.encode_with_transformer(sd, want, dict(width=64, heads=2, layers=3, context_length=77, mlp_ratio=4.0, layer="penultimate"))
        e = rel(got, ref)
        tower_ok = e <= 1e-4
        print(f"  [4] open_clip text tower (3 blocks, penultimate): oracle rel. L2 {e:.2e}")
        out["clip_tower_out"] = ref.detach().numpy()
        for k, v in sd.items():
            out["clip_sd/" + k] = v.numpy()
    except Exception as ex:  # open_clip's constructor surface moves between versions: report, keep the tokenizer pin
        print(f"  [4] open_clip text tower: could not build the reduced model on this version ({ex})")
    return tok_ok and tower_ok


def pin_iqa(out):
    import importlib.util
    import pyiqa
    spec = importlib.util.spec_from_file_location("evaluate_pairs", os.path.join(ROOT, "tools", "evaluate_pairs.py"))
    ep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ep)
    g = torch.Generator().manual_seed(8)
    a = torch.rand(1, 3, 96, 128, generator=g)
    b = (a + 0.05 * torch.randn(a.shape, generator=g)).clamp(0, 1)
    a8, b8 = (a * 255).round() / 255, (b * 255).round() / 255
    p_ref = float(pyiqa.create_metric("psnr", test_y_channel=True, color_space="ycbcr", device="cpu")(a8, b8))
    s_ref = float(pyiqa.create_metric("ssim", test_y_channel=True, color_space="ycbcr", device="cpu")(a8, b8))
    an, bn = a8[0].permute(1, 2, 0).numpy(), b8[0].permute(1, 2, 0).numpy()
    p, s = ep.psnr_y(an, bn), ep.ssim_y(an, bn)
    print(f"  [5] pyiqa PSNR-Y {p_ref:.4f} vs evaluate_pairs {p:.4f}; SSIM-Y {s_ref:.5f} vs {s:.5f}")
    out["iqa_a"], out["iqa_b"], out["iqa_psnr"], out["iqa_ssim"] = an, bn, np.float64(p_ref), np.float64(s_ref)
    return abs(p - p_ref) <= 1e-3 and abs(s - s_ref) <= 1e-4


def pin_lpips(out):
    """LPIPS v0.1 / alex as tools/evaluate_pairs.py restates it against the `lpips` package itself (utils/metrics.py:41-66), on the package's own weights."""
    import importlib.util
    import lpips
    spec = importlib.util.spec_from_file_location("evaluate_pairs", os.path.join(ROOT, "tools", "evaluate_pairs.py"))
    ep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ep)
    ref = lpips.LPIPS(net="alex").eval()
    mine = ep.LPIPS(None, {k: v for k, v in ref.state_dict().items()})
    g = torch.Generator().manual_seed(9)
    a = torch.rand(2, 3, 128, 160, generator=g)
    b = (a + 0.1 * torch.randn(a.shape, generator=g)).clamp(0, 1)
    with torch.no_grad():
        want = ref(a, b, normalize=True).reshape(-1)
    got = mine(a, b, normalize=True)
    print(f"  [7] lpips package {want.tolist()} vs evaluate_pairs {got.tolist()}")
    out["lpips_a"], out["lpips_b"], out["lpips_ref"] = a.numpy(), b.numpy(), want.numpy()
    return bool(torch.allclose(got, want, rtol=1e-4, atol=1e-6))


def pin_ftfy(out):
    import ftfy
    from instarevive_amd.captions import fix_text
    samples = ["caf\u00c3\u00a9 au lait", "it\u00e2\u20ac\u2122s a dog\u00e2\u20ac\u00a6", "\u00c3\u00a2\u00e2\u201a\u00ac\u00e2\u201e\u00a2 twice",
               "na\u00efve caf\u00e9", "\ufb01ne \uff21\uff22\uff23 \u201cq\u201d", "plain ascii", "&lt;b&gt; &amp;amp; x", "line\r\nbreak\u2028here"]
    want = [ftfy.fix_text(t) for t in samples]
    got = [fix_text(t) for t in samples]
    bad = [(t, w, g) for t, w, g in zip(samples, want, got) if w != g]
    print(f"  [6] ftfy.fix_text vs captions.fix_text on {len(samples)} samples: {'identical' if not bad else bad}")
    out["ftfy_samples"], out["ftfy_fixed"] = np.array(samples), np.array(want)
    return not bad


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--dit", default=None, help="folder of a diffusers Transformer2DModel (the converted PixArt / InstaRevive transformer): also compare at full size")
    ap.add_argument("--vae", default=None, help="folder of the diffusers AutoencoderKL (sd-vae-ft-ema): also compare at full size")
    ap.add_argument("--out", default=OUT)
    a = ap.parse_args()
    out, verdict = {}, {}
    for name, fn, args in (("diffusers DiT (items 1, 2)", pin_dit, (None,)), ("diffusers VAE (item 3)", pin_vae, (None,)),
                           ("open_clip (item 4)", pin_clip, ()), ("pyiqa (item 5)", pin_iqa, ()), ("ftfy (item 6)", pin_ftfy, ()), ("lpips (item 7)", pin_lpips, ())):
        print(name)
        try:
            verdict[name] = fn(out, *args)
        except ImportError as ex:
            print(f"  skipped: {ex}")
    if a.dit:
        print("diffusers DiT at full size")
        verdict["DiT, real folder"] = pin_dit(out, a.dit)
    if a.vae:
        print("diffusers VAE at full size")
        verdict["VAE, real folder"] = pin_vae(out, a.vae)
    if out:
        np.savez_compressed(a.out, **out)
        print("wrote", a.out)
    for k, v in verdict.items():
        print(f"{'PINNED ' if v else 'MISMATCH'}  {k}")
    return 0 if verdict and all(verdict.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
