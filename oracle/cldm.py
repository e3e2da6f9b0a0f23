"""CPU restatement of the ControlLDM one-step path, SURVEY.md §8(f) N4 (test infrastructure; see oracle/__init__.py).

Follows, with the options of /root/reference/configs/cldm.yaml (use_spatial_transformer, transformer_depth 1, use_linear_in_transformer,
legacy False, num_head_channels, no scale-shift norm, conv_resample):
  timestep_embedding          ldm/modules/diffusionmodules/util.py:151-171        (cos | sin)
  ResBlock._forward           ldm/modules/diffusionmodules/openaimodel.py:252-272 (GroupNorm32 eps 1e-5, h + emb_out)
  Downsample / Upsample       openaimodel.py:90-160                                (3x3 stride 2 pad 1 / nearest x2 + 3x3)
  CrossAttention.forward      ldm/modules/attention.py:160-198                     (no bias on q / k / v, scale d_head^-0.5)
  GEGLU / FeedForward         attention.py:48-77
  BasicTransformerBlock       attention.py:289-293                                 (attn1(norm1 x) + x, attn2(norm2 x, ctx) + x, ff(norm3 x) + x)
  SpatialTransformer.forward  attention.py:323-350                                 (GroupNorm eps 1e-6, proj_in / proj_out Linear, + x_in)
  UNetModel layout / forward  openaimodel.py:520-710,760-786
  ControlledUnetModel.forward diffusion/cldm.py:32-55                              (h += control.pop(); cat([h, hs.pop() + control.pop()]))
  ControlNet layout / forward diffusion/cldm.py:143-292                            (cat(x, hint) into input_blocks.0, zero_convs, middle_block_out)
  Reflow_ControlLDM           diffusion/cldm.py:486-490 (apply_condition_encoder), :568-588 (sample_log: zT + v at t = num_timesteps - 1)
State dicts use the reference's own parameter names (`input_blocks.1.0.in_layers.2.weight`, ...), addressed under a prefix.
"""
import math

import torch
import torch.nn.functional as F

DEFAULT_CFG = dict(model_channels=320, channel_mult=(1, 2, 4, 4), num_res_blocks=2, attention_resolutions=(4, 2, 1), num_head_channels=64,
                   context_dim=1024, in_channels=4, hint_channels=4, out_channels=4)


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def layout(cfg, control=False):
    """Block structure of input_blocks / output_blocks: lists of layer tuples ('conv', cin, cout) | ('res', cin, cout) | ('xf', ch) |
    ('down', ch) | ('up', ch), plus the channel count entering the middle block."""
    mc, mult, nrb = cfg["model_channels"], list(cfg["channel_mult"]), cfg["num_res_blocks"]
    cin0 = cfg["in_channels"] + (cfg["hint_channels"] if control else 0)
    inb, chans, ch, ds = [[("conv", cin0, mc)]], [mc], mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            layers = [("res", ch, m * mc)]
            ch = m * mc
            if ds in cfg["attention_resolutions"]:
                layers.append(("xf", ch))
            inb.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            inb.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    mid_ch, outb, in_chans = ch, [], list(chans)
    if not control:
        for level, m in list(enumerate(mult))[::-1]:
            for i in range(nrb + 1):
                ich = chans.pop()
                layers = [("res", ch + ich, mc * m)]
                ch = mc * m
                if ds in cfg["attention_resolutions"]:
                    layers.append(("xf", ch))
                if level and i == nrb:
                    layers.append(("up", ch))
                    ds //= 2
                outb.append(layers)
    return inb, mid_ch, outb, in_chans


def state_dict_shapes(cfg=None, control=False):
    """{name: shape} of UNetModel / ControlNet parameters for `cfg`."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    mc, temb, ctx = cfg["model_channels"], 4 * cfg["model_channels"], cfg["context_dim"]
    inb, mid_ch, outb, in_chans = layout(cfg, control)
    s = {"time_embed.0.weight": (temb, mc), "time_embed.0.bias": (temb,), "time_embed.2.weight": (temb, temb), "time_embed.2.bias": (temb,)}

    def conv(p, cin, cout, k):
        s[p + ".weight"], s[p + ".bias"] = (cout, cin, k, k), (cout,)

    def lin(p, cin, cout, bias=True):
        s[p + ".weight"] = (cout, cin)
        if bias:
            s[p + ".bias"] = (cout,)

    def norm(p, c):
        s[p + ".weight"], s[p + ".bias"] = (c,), (c,)

    def res(p, cin, cout):
        norm(p + ".in_layers.0", cin); conv(p + ".in_layers.2", cin, cout, 3); lin(p + ".emb_layers.1", temb, cout)
        norm(p + ".out_layers.0", cout); conv(p + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            conv(p + ".skip_connection", cin, cout, 1)

    def xf(p, c):
        norm(p + ".norm", c); lin(p + ".proj_in", c, c); lin(p + ".proj_out", c, c)
        t = p + ".transformer_blocks.0"
        for a, kd in (("attn1", c), ("attn2", ctx)):
            lin(f"{t}.{a}.to_q", c, c, False); lin(f"{t}.{a}.to_k", kd, c, False); lin(f"{t}.{a}.to_v", kd, c, False); lin(f"{t}.{a}.to_out.0", c, c)
        lin(t + ".ff.net.0.proj", c, 8 * c); lin(t + ".ff.net.2", 4 * c, c)
        for n in ("norm1", "norm2", "norm3"):
            norm(f"{t}.{n}", c)

    def block(p, layers):
        for k, L in enumerate(layers):
            if L[0] == "conv":
                conv(f"{p}.{k}", L[1], L[2], 3)
            elif L[0] == "res":
                res(f"{p}.{k}", L[1], L[2])
            elif L[0] == "xf":
                xf(f"{p}.{k}", L[1])
            elif L[0] == "down":
                conv(f"{p}.{k}.op", L[1], L[1], 3)
            elif L[0] == "up":
                conv(f"{p}.{k}.conv", L[1], L[1], 3)

    for i, layers in enumerate(inb):
        block(f"input_blocks.{i}", layers)
    block("middle_block", [("res", mid_ch, mid_ch), ("xf", mid_ch), ("res", mid_ch, mid_ch)])
    if control:
        for i, c in enumerate(in_chans):
            conv(f"zero_convs.{i}.0", c, c, 1)
        conv("middle_block_out.0", mid_ch, mid_ch, 1)
    else:
        for i, layers in enumerate(outb):
            block(f"output_blocks.{i}", layers)
        norm("out.0", mc)
        conv("out.2", mc, cfg["out_channels"], 3)
    return s


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _res(sd, p, x, emb):
    h = _conv(sd, p + ".in_layers.2", F.silu(F.group_norm(x, 32, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"], eps=1e-5)))
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = _conv(sd, p + ".out_layers.3", F.silu(F.group_norm(h, 32, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"], eps=1e-5)))
    if p + ".skip_connection.weight" in sd:
        x = _conv(sd, p + ".skip_connection", x, padding=0)
    return x + h


def _attn(sd, p, x, ctx, heads):
    B, T, C = x.shape
    d = C // heads
    q = F.linear(x, sd[p + ".to_q.weight"]).view(B, T, heads, d).transpose(1, 2)
    k = F.linear(ctx, sd[p + ".to_k.weight"]).view(B, ctx.shape[1], heads, d).transpose(1, 2)
    v = F.linear(ctx, sd[p + ".to_v.weight"]).view(B, ctx.shape[1], heads, d).transpose(1, 2)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * d ** -0.5
    o = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v).transpose(1, 2).reshape(B, T, C)
    return F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def _xf(sd, p, x, context, d_head):
    B, C, H, W = x.shape
    heads = C // d_head
    h = F.group_norm(x, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-6)
    h = h.flatten(2).transpose(1, 2)
    h = F.linear(h, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    t = p + ".transformer_blocks.0"
    n1 = _ln(sd, t + ".norm1", h)
    h = _attn(sd, t + ".attn1", n1, n1, heads) + h
    h = _attn(sd, t + ".attn2", _ln(sd, t + ".norm2", h), context, heads) + h
    a, g = F.linear(_ln(sd, t + ".norm3", h), sd[t + ".ff.net.0.proj.weight"], sd[t + ".ff.net.0.proj.bias"]).chunk(2, dim=-1)
    h = F.linear(a * F.gelu(g), sd[t + ".ff.net.2.weight"], sd[t + ".ff.net.2.bias"]) + h
    h = F.linear(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return h.transpose(1, 2).reshape(B, C, H, W) + x


def _block(sd, p, layers, h, emb, context, d_head):
    for k, L in enumerate(layers):
        if L[0] == "conv":
            h = _conv(sd, f"{p}.{k}", h)
        elif L[0] == "res":
            h = _res(sd, f"{p}.{k}", h, emb)
        elif L[0] == "xf":
            h = _xf(sd, f"{p}.{k}", h, context, d_head)
        elif L[0] == "down":
            h = _conv(sd, f"{p}.{k}.op", h, stride=2)
        elif L[0] == "up":
            h = _conv(sd, f"{p}.{k}.conv", F.interpolate(h, scale_factor=2, mode="nearest"))
    return h


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd


def _emb(sd, t, mc):
    e = F.linear(timestep_embedding(t, mc), sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    return F.linear(F.silu(e), sd["time_embed.2.weight"], sd["time_embed.2.bias"])


@torch.no_grad()
def controlnet_forward(sd, x, hint, t, context, cfg=None, prefix=""):
    """ControlNet.forward (cldm.py:276-292): the 13 control tensors."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    sd = _sub(sd, prefix)
    inb, mid_ch, _, _ = layout(cfg, control=True)
    emb, dh = _emb(sd, t, cfg["model_channels"]), cfg["num_head_channels"]
    h, outs = torch.cat((x, hint), dim=1), []
    for i, layers in enumerate(inb):
        h = _block(sd, f"input_blocks.{i}", layers, h, emb, context, dh)
        outs.append(_conv(sd, f"zero_convs.{i}.0", h, padding=0))
    h = _block(sd, "middle_block", [("res", mid_ch, mid_ch), ("xf", mid_ch), ("res", mid_ch, mid_ch)], h, emb, context, dh)
    outs.append(_conv(sd, "middle_block_out.0", h, padding=0))
    return outs


@torch.no_grad()
def unet_forward(sd, x, t, context, control=None, cfg=None, prefix=""):
    """ControlledUnetModel.forward (cldm.py:32-55) with only_mid_control = False."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    sd = _sub(sd, prefix)
    inb, mid_ch, outb, _ = layout(cfg)
    emb, dh = _emb(sd, t, cfg["model_channels"]), cfg["num_head_channels"]
    control = list(control) if control is not None else None
    hs, h = [], x
    for i, layers in enumerate(inb):
        h = _block(sd, f"input_blocks.{i}", layers, h, emb, context, dh)
        hs.append(h)
    h = _block(sd, "middle_block", [("res", mid_ch, mid_ch), ("xf", mid_ch), ("res", mid_ch, mid_ch)], h, emb, context, dh)
    if control is not None:
        h = h + control.pop()
    for i, layers in enumerate(outb):
        skip = hs.pop() if control is None else hs.pop() + control.pop()
        h = _block(sd, f"output_blocks.{i}", layers, torch.cat([h, skip], dim=1), emb, context, dh)
    return _conv(sd, "out.2", F.silu(F.group_norm(h, 32, sd["out.0.weight"], sd["out.0.bias"], eps=1e-5)))


@torch.no_grad()
def reflow_sample(sd, zT, c_latent, context, cfg=None, num_timesteps=1000, unet_prefix="model.diffusion_model.", control_prefix="control_model."):
    """Reflow_ControlLDM.sample_log (cldm.py:568-588) for a given zT: zT + v with control_scales = 1."""
    t = torch.ones(zT.shape[0]) * (num_timesteps - 1)
    control = None if c_latent is None else controlnet_forward(sd, zT, c_latent, t, context, cfg, control_prefix)
    return zT + unet_forward(sd, zT, t, context, control, cfg, unet_prefix)
