"""CPU restatement of the SwinIR stage-1 restorer (test infrastructure; see oracle/__init__.py).

Follows /root/reference/diffusion/model/swinir.py: Mlp :35-41, window_partition/reverse :44-73, WindowAttention.forward
:125-156, calculate_mask :227-248, SwinTransformerBlock.forward :250-290, RSTB.forward :492-493, forward_features
:852-865, SwinIR.forward ('nearest+conv', sf 8, PixelUnshuffle) :867-905. Functional: weights come from a state dict with
the reference checkpoint's key names.
"""
import torch
import torch.nn.functional as F

DEFAULT_CFG = dict(embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8, mlp_ratio=2, sf=8, img_range=1.0,
                   unshuffle_scale=8)
RGB_MEAN = (0.4488, 0.4371, 0.4040)  # swinir.py:693


def _relative_position_index(ws):  # swinir.py:104-113
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _partition(x, ws):
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)


def _reverse(win, ws, H, W):
    B = win.shape[0] // ((H // ws) * (W // ws))
    return win.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def shift_mask(H, W, ws, shift):  # swinir.py:227-248
    img = torch.zeros(1, H, W, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = _partition(img, ws).squeeze(-1)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def _block(sd, p, x, H, W, heads, ws, shift, rpi):
    B, L, C = x.shape
    shortcut = x
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5).view(B, H, W, C)
    if shift:
        h = torch.roll(h, (-shift, -shift), (1, 2))
    win = _partition(h, ws)
    Bw, N, _ = win.shape
    qkv = F.linear(win, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(Bw, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = sd[p + "attn.relative_position_bias_table"][rpi.view(-1)].view(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift:
        mask = shift_mask(H, W, ws, shift)
        nW = mask.shape[0]
        attn = (attn.view(Bw // nW, nW, heads, N, N) + mask[None, :, None]).view(-1, heads, N, N)
    attn = attn.softmax(-1)
    o = (attn @ v).transpose(1, 2).reshape(Bw, N, C)
    o = F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h = _reverse(o, ws, H, W)
    if shift:
        h = torch.roll(h, (shift, shift), (1, 2))
    x = shortcut + h.view(B, H * W, C)
    m = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    m = F.linear(F.gelu(F.linear(m, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + m


@torch.no_grad()
def swinir_forward(sd, x, cfg=None):
    """x: [B,3,H,W] fp32 in [0,1]; returns [B,3,H,W] (sf 8 after unshuffle 8)."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    ws, C = cfg["window_size"], cfg["embed_dim"]
    assert cfg["sf"] == 8 and cfg["unshuffle_scale"] == 8
    H0, W0 = x.shape[2:]
    x = F.pad(x, (0, (ws - W0 % ws) % ws, 0, (ws - H0 % ws) % ws), "reflect")
    mean = torch.tensor(RGB_MEAN).view(1, 3, 1, 1)
    x = (x - mean) * cfg["img_range"]
    x = F.conv2d(F.pixel_unshuffle(x, 8), sd["conv_first.1.weight"], sd["conv_first.1.bias"], padding=1)
    first = x
    B, _, H, W = x.shape
    rpi = _relative_position_index(ws)
    t = x.flatten(2).transpose(1, 2)
    t = F.layer_norm(t, (C,), sd["patch_embed.norm.weight"], sd["patch_embed.norm.bias"], 1e-5)
    for i, depth in enumerate(cfg["depths"]):
        inp = t
        for j in range(depth):
            t = _block(sd, f"layers.{i}.residual_group.blocks.{j}.", t, H, W, cfg["num_heads"][i], ws, 0 if j % 2 == 0 else ws // 2, rpi)
        img = t.transpose(1, 2).reshape(B, C, H, W)
        img = F.conv2d(img, sd[f"layers.{i}.conv.weight"], sd[f"layers.{i}.conv.bias"], padding=1)
        t = img.flatten(2).transpose(1, 2) + inp
    t = F.layer_norm(t, (C,), sd["norm.weight"], sd["norm.bias"], 1e-5)
    x = t.transpose(1, 2).reshape(B, C, H, W)
    x = F.conv2d(x, sd["conv_after_body.weight"], sd["conv_after_body.bias"], padding=1) + first
    x = F.leaky_relu(F.conv2d(x, sd["conv_before_upsample.0.weight"], sd["conv_before_upsample.0.bias"], padding=1), 0.01)
    for name in ("conv_up1", "conv_up2", "conv_up3"):
        x = F.leaky_relu(F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), sd[name + ".weight"], sd[name + ".bias"], padding=1), 0.2)
    x = F.conv2d(F.leaky_relu(F.conv2d(x, sd["conv_hr.weight"], sd["conv_hr.bias"], padding=1), 0.2), sd["conv_last.weight"],
                 sd["conv_last.bias"], padding=1)
    x = x / cfg["img_range"] + mean
    return x[:, :, :H0 * cfg["sf"], :W0 * cfg["sf"]]  # swinir.py:905 (a no-op crop: output is input-sized)


def state_dict_shapes(cfg=None):
    """Parameter names/shapes of the reference SwinIR state dict (buffers relative_position_index/attn_mask excluded)."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    C, ws, hid = cfg["embed_dim"], cfg["window_size"], int(cfg["embed_dim"] * cfg["mlp_ratio"])
    s = {"conv_first.1.weight": (C, 3 * 64, 3, 3), "conv_first.1.bias": (C,), "patch_embed.norm.weight": (C,), "patch_embed.norm.bias": (C,)}
    for i, depth in enumerate(cfg["depths"]):
        nh = cfg["num_heads"][i]
        for j in range(depth):
            p = f"layers.{i}.residual_group.blocks.{j}."
            s.update({p + "norm1.weight": (C,), p + "norm1.bias": (C,), p + "attn.relative_position_bias_table": ((2 * ws - 1) ** 2, nh),
                      p + "attn.qkv.weight": (3 * C, C), p + "attn.qkv.bias": (3 * C,), p + "attn.proj.weight": (C, C), p + "attn.proj.bias": (C,),
                      p + "norm2.weight": (C,), p + "norm2.bias": (C,), p + "mlp.fc1.weight": (hid, C), p + "mlp.fc1.bias": (hid,),
                      p + "mlp.fc2.weight": (C, hid), p + "mlp.fc2.bias": (C,)})
        s.update({f"layers.{i}.conv.weight": (C, C, 3, 3), f"layers.{i}.conv.bias": (C,)})
    s.update({"norm.weight": (C,), "norm.bias": (C,), "conv_after_body.weight": (C, C, 3, 3), "conv_after_body.bias": (C,),
              "conv_before_upsample.0.weight": (64, C, 3, 3), "conv_before_upsample.0.bias": (64,)})
    for n in ("conv_up1", "conv_up2", "conv_up3", "conv_hr"):
        s.update({n + ".weight": (64, 64, 3, 3), n + ".bias": (64,)})
    s.update({"conv_last.weight": (3, 64, 3, 3), "conv_last.bias": (3,)})
    return s
