"""CPU restatement of the SD VAE (sd-vae-ft-ema) encode/decode (test infrastructure; see oracle/__init__.py).

Arithmetic follows the in-tree LDM twin /root/reference/ldm/modules/diffusionmodules/model.py (Normalize :47-48,
nonlinearity :43-45, Upsample :63-67, Downsample :82-89, ResnetBlock.forward :131-151, AttnBlock.forward :181-205,
Encoder.forward :521-546, Decoder.forward :622-655), ldm/models/autoencoder.py:82-91 and
ldm/modules/distributions/distributions.py:24-27,59-60 (mode() == mean). Weights use the *diffusers* AutoencoderKL key
names because that is the checkpoint surface the CLI loads (test_scripts/inference.py:236); `ldm_to_diffusers` maps an
LDM state dict onto them (used to pin this file against the imported reference modules).
"""
import torch
import torch.nn.functional as F

DEFAULT_CFG = dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, scaling_factor=0.18215)


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _resnet(sd, p, x):
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x)))
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h)))
    if p + ".conv_shortcut.weight" in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def _attn_w(sd, p, new, old):
    w = sd[p + "." + new + ".weight"] if p + "." + new + ".weight" in sd else sd[p + "." + old + ".weight"]
    b = sd[p + "." + new + ".bias"] if p + "." + new + ".bias" in sd else sd[p + "." + old + ".bias"]
    return w.reshape(w.shape[0], -1), b


def _attn(sd, p, x):
    B, C, H, W = x.shape
    h = _gn(sd, p + ".group_norm", x).flatten(2).transpose(1, 2)  # B, HW, C
    q = F.linear(h, *_attn_w(sd, p, "to_q", "query"))
    k = F.linear(h, *_attn_w(sd, p, "to_k", "key"))
    v = F.linear(h, *_attn_w(sd, p, "to_v", "value"))
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None], scale=C ** -0.5)[:, 0]
    o = F.linear(o, *_attn_w(sd, p, "to_out.0", "proj_attn"))
    return x + o.transpose(1, 2).reshape(B, C, H, W)


@torch.no_grad()
def vae_encode_mean(sd, x, cfg=None):
    """AutoencoderKL.encode(x).latent_dist.mode(): x [B,3,H,W] in [-1,1] -> [B,4,H/8,W/8]."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    nl = len(cfg["ch_mult"])
    h = _conv(sd, "encoder.conv_in", x)
    for l in range(nl):
        for j in range(cfg["num_res_blocks"]):
            h = _resnet(sd, f"encoder.down_blocks.{l}.resnets.{j}", h)
        if l != nl - 1:
            h = _conv(sd, f"encoder.down_blocks.{l}.downsamplers.0.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = _resnet(sd, "encoder.mid_block.resnets.0", h)
    h = _attn(sd, "encoder.mid_block.attentions.0", h)
    h = _resnet(sd, "encoder.mid_block.resnets.1", h)
    h = _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.conv_norm_out", h)))
    moments = _conv(sd, "quant_conv", h, padding=0)
    return moments[:, :cfg["z_channels"]]


@torch.no_grad()
def vae_decode(sd, z, cfg=None):
    """AutoencoderKL.decode(z).sample: z [B,4,h,w] -> [B,3,8h,8w] in ~[-1,1]."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    nl = len(cfg["ch_mult"])
    h = _conv(sd, "decoder.conv_in", _conv(sd, "post_quant_conv", z, padding=0))
    h = _resnet(sd, "decoder.mid_block.resnets.0", h)
    h = _attn(sd, "decoder.mid_block.attentions.0", h)
    h = _resnet(sd, "decoder.mid_block.resnets.1", h)
    for i in range(nl):  # diffusers up_blocks.i == ldm up[nl-1-i]
        for j in range(cfg["num_res_blocks"] + 1):
            h = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", h)
        if i != nl - 1:
            h = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    return _conv(sd, "decoder.conv_out", F.silu(_gn(sd, "decoder.conv_norm_out", h)))


def ldm_to_diffusers(enc_sd=None, dec_sd=None, quant=None, post_quant=None, cfg=None):
    """Rename LDM Encoder/Decoder state dicts (model.py:455-655) to diffusers AutoencoderKL keys."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    nl = len(cfg["ch_mult"])
    out = {}

    def res(dst, src, sd):
        for a, b in (("norm1", "norm1"), ("conv1", "conv1"), ("norm2", "norm2"), ("conv2", "conv2"), ("conv_shortcut", "nin_shortcut")):
            for t in ("weight", "bias"):
                if f"{src}.{b}.{t}" in sd:
                    out[f"{dst}.{a}.{t}"] = sd[f"{src}.{b}.{t}"]

    def attn(dst, src, sd):
        for a, b in (("group_norm", "norm"), ("to_q", "q"), ("to_k", "k"), ("to_v", "v"), ("to_out.0", "proj_out")):
            for t in ("weight", "bias"):
                w = sd[f"{src}.{b}.{t}"]
                out[f"{dst}.{a}.{t}"] = w.reshape(w.shape[0], -1) if (t == "weight" and a != "group_norm") else w

    def half(prefix, sd, decoder):
        for n in ("conv_in", "conv_out"):
            for t in ("weight", "bias"):
                out[f"{prefix}.{n}.{t}"] = sd[f"{n}.{t}"]
        for t in ("weight", "bias"):
            out[f"{prefix}.conv_norm_out.{t}"] = sd[f"norm_out.{t}"]
        res(f"{prefix}.mid_block.resnets.0", "mid.block_1", sd)
        res(f"{prefix}.mid_block.resnets.1", "mid.block_2", sd)
        attn(f"{prefix}.mid_block.attentions.0", "mid.attn_1", sd)
        for l in range(nl):
            if not decoder:
                for j in range(cfg["num_res_blocks"]):
                    res(f"{prefix}.down_blocks.{l}.resnets.{j}", f"down.{l}.block.{j}", sd)
                if l != nl - 1:
                    for t in ("weight", "bias"):
                        out[f"{prefix}.down_blocks.{l}.downsamplers.0.conv.{t}"] = sd[f"down.{l}.downsample.conv.{t}"]
            else:
                i = nl - 1 - l
                for j in range(cfg["num_res_blocks"] + 1):
                    res(f"{prefix}.up_blocks.{i}.resnets.{j}", f"up.{l}.block.{j}", sd)
                if l != 0:
                    for t in ("weight", "bias"):
                        out[f"{prefix}.up_blocks.{i}.upsamplers.0.conv.{t}"] = sd[f"up.{l}.upsample.conv.{t}"]

    if enc_sd is not None:
        half("encoder", enc_sd, False)
    if dec_sd is not None:
        half("decoder", dec_sd, True)
    if quant is not None:
        out["quant_conv.weight"], out["quant_conv.bias"] = quant
    if post_quant is not None:
        out["post_quant_conv.weight"], out["post_quant_conv.bias"] = post_quant
    return out


def state_dict_shapes(cfg=None, encoder=True, decoder=True):
    """diffusers AutoencoderKL parameter names / shapes for the given architecture."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    ch, mult, nrb, z = cfg["ch"], cfg["ch_mult"], cfg["num_res_blocks"], cfg["z_channels"]
    nl = len(mult)
    s = {}

    def res(p, cin, cout):
        s.update({p + ".norm1.weight": (cin,), p + ".norm1.bias": (cin,), p + ".conv1.weight": (cout, cin, 3, 3), p + ".conv1.bias": (cout,),
                  p + ".norm2.weight": (cout,), p + ".norm2.bias": (cout,), p + ".conv2.weight": (cout, cout, 3, 3), p + ".conv2.bias": (cout,)})
        if cin != cout:
            s.update({p + ".conv_shortcut.weight": (cout, cin, 1, 1), p + ".conv_shortcut.bias": (cout,)})

    def attn(p, c):
        s.update({p + ".group_norm.weight": (c,), p + ".group_norm.bias": (c,)})
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            s.update({f"{p}.{n}.weight": (c, c), f"{p}.{n}.bias": (c,)})

    if encoder:
        s.update({"encoder.conv_in.weight": (ch, 3, 3, 3), "encoder.conv_in.bias": (ch,)})
        cin = ch
        for l in range(nl):
            cout = ch * mult[l]
            for j in range(nrb):
                res(f"encoder.down_blocks.{l}.resnets.{j}", cin, cout)
                cin = cout
            if l != nl - 1:
                s.update({f"encoder.down_blocks.{l}.downsamplers.0.conv.weight": (cin, cin, 3, 3), f"encoder.down_blocks.{l}.downsamplers.0.conv.bias": (cin,)})
        res("encoder.mid_block.resnets.0", cin, cin)
        attn("encoder.mid_block.attentions.0", cin)
        res("encoder.mid_block.resnets.1", cin, cin)
        s.update({"encoder.conv_norm_out.weight": (cin,), "encoder.conv_norm_out.bias": (cin,), "encoder.conv_out.weight": (2 * z, cin, 3, 3),
                  "encoder.conv_out.bias": (2 * z,), "quant_conv.weight": (2 * z, 2 * z, 1, 1), "quant_conv.bias": (2 * z,)})
    if decoder:
        cin = ch * mult[-1]
        s.update({"post_quant_conv.weight": (z, z, 1, 1), "post_quant_conv.bias": (z,), "decoder.conv_in.weight": (cin, z, 3, 3), "decoder.conv_in.bias": (cin,)})
        res("decoder.mid_block.resnets.0", cin, cin)
        attn("decoder.mid_block.attentions.0", cin)
        res("decoder.mid_block.resnets.1", cin, cin)
        for i in range(nl):
            cout = ch * mult[nl - 1 - i]
            for j in range(nrb + 1):
                res(f"decoder.up_blocks.{i}.resnets.{j}", cin, cout)
                cin = cout
            if i != nl - 1:
                s.update({f"decoder.up_blocks.{i}.upsamplers.0.conv.weight": (cin, cin, 3, 3), f"decoder.up_blocks.{i}.upsamplers.0.conv.bias": (cin,)})
        s.update({"decoder.conv_norm_out.weight": (cin,), "decoder.conv_norm_out.bias": (cin,), "decoder.conv_out.weight": (3, cin, 3, 3),
                  "decoder.conv_out.bias": (3,)})
    return s
