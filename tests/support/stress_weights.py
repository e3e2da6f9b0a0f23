"""Weights that look like RELEASED weights where it matters numerically: every timing and every PSNR of rounds 1-4 was on
seeded uniform weights, whose attention rows are near-uniform and whose channels all carry the same scale. Real SD-VAE / PixArt weights have
(a) a few channels that are 10-100x larger than the rest ("massive activations": fixed residual-stream channels in the DiT, outlier
channels of the VAE's ResnetBlocks, reference shapes ldm/modules/diffusionmodules/model.py:102-151, PixArt_blocks.py:123-158) and (b) attention
rows dominated by a few keys (model.py:181-205, PixArt_blocks.py:43-58). stress_state_dicts() turns bench.py's seeded state dicts into such a set:

  * channel outliers: `frac` of the channels (at least one) scaled by `gain` - in the DiT the SAME residual-stream channels in every block (rows of
    attn1.to_out.0 / attn2.to_out.0 / ff.net.2: the three writers of the stream) plus `frac` of the MLP hidden units (rows of ff.net.0.proj); in
    the VAE the same output channels of conv1 / conv2 of every ResnetBlock of a given width (so the outliers stack on the block's residual path);
  * peaky softmax: q and k projections (weight and bias) of the DiT self-attention and of the VAE mid-block attention scaled by sqrt(logit_gain),
    i.e. every logit times logit_gain (one factor, or one per DiT block / VAE half as tests/golden/make_stress_fixture.py calibrates them).

Test infrastructure (tests/, tests/golden/make_stress_fixture.py, bench.py --logit_gain): the product never imports it."""
import torch


def _scale_rows(sd, key, idx, gain):
    for leaf in (".weight", ".bias"):
        if key + leaf in sd:
            t = sd[key + leaf].clone()
            t[idx] = t[idx] * gain
            sd[key + leaf] = t


def _scale_all(sd, key, gain):
    for leaf in (".weight", ".bias"):
        if key + leaf in sd:
            sd[key + leaf] = sd[key + leaf] * gain


def stress_state_dicts(sds, frac=0.01, gain=30.0, logit_gain=4.0, seed=9):
    """sds: {'swin', 'vae', 'dit'} state dicts (bench.build_models / random_state_dict). Returns new dicts (tensors of untouched keys shared)."""
    g = torch.Generator().manual_seed(seed)
    out = {k: dict(v) for k, v in sds.items()}
    dit, vae = out["dit"], out["vae"]
    layers = sorted({int(k.split(".")[1]) for k in dit if k.startswith("transformer_blocks.")})
    if frac > 0 and gain != 1.0:
        C = dit["transformer_blocks.0.attn1.to_out.0.weight"].shape[0]
        hid = dit["transformer_blocks.0.ff.net.0.proj.weight"].shape[0]
        stream = torch.randperm(C, generator=g)[:max(1, int(round(frac * C)))]
        hidden = torch.randperm(hid, generator=g)[:max(1, int(round(frac * hid)))]
        for l in layers:
            p = f"transformer_blocks.{l}."
            for name in ("attn1.to_out.0", "attn2.to_out.0", "ff.net.2"):
                _scale_rows(dit, p + name, stream, gain)
            _scale_rows(dit, p + "ff.net.0.proj", hidden, gain)
        per_width = {}
        for k in sorted(vae):
            if k.endswith((".conv1.weight", ".conv2.weight")) and "resnets" in k:
                cout = vae[k].shape[0]
                if cout not in per_width:
                    per_width[cout] = torch.randperm(cout, generator=g)[:max(1, int(round(frac * cout)))]
                _scale_rows(vae, k[:-len(".weight")], per_width[cout], gain)
    # logit_gain: one factor for every attention, or {"dit": [one per block], "vae_encoder": f, "vae_decoder": f} - the calibrated form the fixture
    # stores: with the stream outliers in place LayerNorm shrinks the ordinary channels, so a single factor cannot make every block's rows peaky
    if isinstance(logit_gain, dict):
        per_block = {l: float(logit_gain["dit"][i]) for i, l in enumerate(layers)}
        per_half = {"encoder": float(logit_gain["vae_encoder"]), "decoder": float(logit_gain["vae_decoder"])}
    else:
        per_block = {l: float(logit_gain) for l in layers}
        per_half = {"encoder": float(logit_gain), "decoder": float(logit_gain)}
    for l in layers:
        if per_block[l] != 1.0:
            for name in ("attn1.to_q", "attn1.to_k"):
                _scale_all(dit, f"transformer_blocks.{l}.{name}", per_block[l] ** 0.5)
    for half in ("encoder", "decoder"):
        if per_half[half] != 1.0:
            for name in ("to_q", "to_k"):
                _scale_all(vae, f"{half}.mid_block.attentions.0.{name}", per_half[half] ** 0.5)
    return out
