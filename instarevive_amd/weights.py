"""Checkpoint surface -> device layouts.

Takes state dicts with the REFERENCE's key names (SwinIR: diffusion/model/swinir.py module tree; VAE: diffusers
AutoencoderKL; DiT: diffusers Transformer2DModel, i.e. tools/convert_pixart_to_diffusers.py:30-154 output) and produces
the named, packed tensors the C ABI expects (ir_upload / ir_*_configure in include/instarevive_hip.h):

  * conv weights   [Cout][Cin][3][3] fp32 -> bf16 [Cout_pad][9][Cin_pad]  (tap-major, channel-minor = K contiguous)
  * linear weights [N][K] fp32           -> bf16 [N_pad][K_pad]
  * biases / norm affine / tables        -> fp32
Padding is zero-filled, so padded channels carry exact zeros through the network. This is load-time plumbing on the
host (the reference does the equivalent in nn.Module.load_state_dict + .to(device)); no arithmetic of the hot path.
"""
import math

import numpy as np
import torch


def _bf16(t):
    return t.to(torch.float32).to(torch.bfloat16).contiguous().view(torch.int16)


def pad_to(x, m):
    return (x + m - 1) // m * m


def pack_conv3x3(w, cin_pad, cout_pad, scale=1.0):
    co, ci, kh, kw = w.shape
    p = torch.zeros(cout_pad, kh * kw, cin_pad, dtype=torch.float32)
    p[:co, :, :ci] = w.to(torch.float32).permute(0, 2, 3, 1).reshape(co, kh * kw, ci) * scale
    return _bf16(p.reshape(cout_pad, kh * kw * cin_pad))


def pack_conv_up2x2(w):
    """A 3x3 conv that follows a nearest-2x upsample (ldm model.py:63-67) as FOUR 2x2 convs on the low-resolution tensor, one per output
    phase (dy, dx): output pixel (2y + dy, 2x + dx) reads the upsampled rows 2y + dy - 1 .. + 1, i.e. the low-resolution rows
    y - 1 + dy + sy (sy = 0, 1) - for dy = 0 tap ky = 0 comes from sy = 0 and taps 1, 2 from sy = 1; for dy = 1 taps 0, 1 from sy = 0 and tap 2
    from sy = 1 - and likewise for the columns. The taps that land on one source pixel are SUMMED here in fp32 and rounded to bf16 once.
    w: [Cout][Cin][3][3] -> [4 phases = 2 dy + dx][Cout][4 taps = 2 sy + sx][Cin] bf16 (csrc/conv_s1.hip, conv_halo_s1_kernel<0, 4>)."""
    co, ci = w.shape[:2]
    wf = w.to(torch.float32)
    groups = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}   # phase offset -> taps of source offset 0 / 1
    out = torch.zeros(4, co, 4, ci, dtype=torch.float32)
    for dy in (0, 1):
        for dx in (0, 1):
            for sy in (0, 1):
                for sx in (0, 1):
                    acc = torch.zeros(co, ci, dtype=torch.float32)
                    for ky in groups[dy][sy]:
                        for kx in groups[dx][sx]:
                            acc += wf[:, :, ky, kx]
                    out[2 * dy + dx, :, 2 * sy + sx, :] = acc
    return _bf16(out.reshape(4 * co, 4 * ci))


FP8_ACT_SCALE = 16.0   # GroupNorm+SiLU outputs reach the fp8 convs as e4m3(x * 16) (csrc/api.cpp: FP8_ACT_SCALE)
FP8_MAX = 448.0        # largest finite OCP e4m3 value


def pack_conv3x3_fp8(w, bias):
    """3x3 conv weights [Cout][Cin][3][3] -> (OCP e4m3 bytes [Cout][9*Cin] quantised per OUTPUT channel with scale amax / 448,
    dequantisation factor per channel = weight scale / activation scale, bias / that factor): the conv epilogue computes
    (acc + bias') * factor. BASELINE.json configs[4], "fp8 MFMA for ... VAE conv weights"."""
    co, ci = w.shape[:2]
    wf = w.to(torch.float32).permute(0, 2, 3, 1).reshape(co, 9 * ci)
    ws = (wf.abs().amax(dim=1) / FP8_MAX).clamp_min(1e-12)
    q = (wf / ws[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    factor = (ws / FP8_ACT_SCALE).contiguous()
    return q, factor, (bias.to(torch.float32) / factor).contiguous()


def pack_linear(w, n_pad, k_pad, row_map=None, col_map=None):
    """row_map / col_map: LongTensor giving, for each ORIGINAL row / column, its position in the padded layout."""
    w = w.to(torch.float32).reshape(w.shape[0], -1)
    n, k = w.shape
    rows = torch.arange(n) if row_map is None else row_map
    cols = torch.arange(k) if col_map is None else col_map
    p = torch.zeros(n_pad, k_pad, dtype=torch.float32)
    p[rows[:, None], cols[None, :]] = w
    return _bf16(p)


def pad_vec(b, n_pad, idx=None, scale=1.0, shift=None):
    p = torch.zeros(n_pad, dtype=torch.float32)
    v = b.to(torch.float32) * scale
    if shift is not None:
        v = v + shift
    if idx is None:
        p[: v.numel()] = v
    else:
        p[idx] = v
    return p


# ------------------------------------------------------------------------------------------------ SwinIR
SWIN_MEAN = (0.4488, 0.4371, 0.4040)  # diffusion/model/swinir.py:693


def swin_relative_position_index(ws=8):
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def swinir_expected_keys(cfg):
    """Names of the reference SwinIR state dict (parameters + the two kinds of buffers a strict load carries)."""
    C, ws = cfg["embed_dim"], cfg["window_size"]
    keys = ["conv_first.1.weight", "conv_first.1.bias", "patch_embed.norm.weight", "patch_embed.norm.bias"]
    for i, depth in enumerate(cfg["depths"]):
        for j in range(depth):
            p = f"layers.{i}.residual_group.blocks.{j}."
            keys += [p + k for k in ("norm1.weight", "norm1.bias", "attn.relative_position_bias_table", "attn.relative_position_index",
                                     "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias",
                                     "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")]
            if j % 2 == 1:
                keys.append(p + "attn_mask")
        keys += [f"layers.{i}.conv.weight", f"layers.{i}.conv.bias"]
    keys += ["norm.weight", "norm.bias", "conv_after_body.weight", "conv_after_body.bias", "conv_before_upsample.0.weight",
             "conv_before_upsample.0.bias"]
    for n in ("conv_up1", "conv_up2", "conv_up3", "conv_hr", "conv_last"):
        keys += [n + ".weight", n + ".bias"]
    return keys


def _acc_order(n):
    """k positions -> source index for the transposed-register kernels (csrc/swin_fused.hip): position 16s + 8h + j of k-step s holds
    element 16s + 8(j >> 2) + 4h + (j & 3) — the order in which a 32x32 MFMA accumulator tile, packed to bf16, presents its rows."""
    p = torch.arange(n)
    s, h, j = p // 16, (p // 8) % 2, p % 8
    return 16 * s + 8 * (j // 4) + 4 * h + (j % 4)


def pack_swin_mlp(fc1_w, fc1_b, fc2_w, fc2_b, n2_g, n2_b, C, Cp, hid_p):
    """Weight tiles + vectors of swin_mlp_kernel. Per 32 hidden units jt one 28 KB step: W1 tile [32 rows = hidden units][192 k positions]
    bf16 with 400-byte rows (16 bytes of padding: conflict-free ds_read_b128 across rows) padded to 13 KB, then W2 tile [192 output
    channels][32 k positions = the tile's hidden units] with 80-byte rows (15 KB). k positions follow _acc_order. Vectors (fp32):
    LayerNorm gain / bias padded to 192 (zero beyond C: padded channels normalise to exactly 0), fc1 bias padded to hid_p, fc2 bias."""
    hid = fc1_w.shape[0]
    nj = hid_p // 32
    w1 = torch.zeros(hid_p, Cp)
    w1[:hid, :C] = fc1_w.float()
    w1 = w1[:, _acc_order(Cp)]                                     # input channels in accumulator order
    w2 = torch.zeros(Cp, hid_p)
    w2[:C, :hid] = fc2_w.float()
    w2 = w2[:, _acc_order(hid_p)]                                  # hidden units in accumulator order (within each 16: tiles of 32 stay intact)
    tiles = torch.zeros(nj, 13312 + 15360, dtype=torch.uint8)
    for jt in range(nj):
        t1 = torch.zeros(32, 200, dtype=torch.int16)               # 400-byte rows
        t1[:, :Cp] = _bf16(w1[32 * jt:32 * jt + 32])
        t2 = torch.zeros(Cp, 40, dtype=torch.int16)                # 80-byte rows
        t2[:, :32] = _bf16(w2[:, 32 * jt:32 * jt + 32])
        tiles[jt, :12800] = t1.reshape(-1).view(torch.uint8)
        tiles[jt, 13312:] = t2.reshape(-1).view(torch.uint8)
    vec = torch.zeros(3 * Cp + hid_p)
    vec[:C], vec[Cp:Cp + C] = n2_g.float(), n2_b.float()
    vec[2 * Cp:2 * Cp + hid] = fc1_b.float()
    vec[2 * Cp + hid_p:2 * Cp + hid_p + C] = fc2_b.float()
    return tiles.contiguous(), vec.contiguous()


def swin_masked_bias(biasT, ws=8, shift=4):
    """Bias tables of a SHIFTED block for the fused kernels: [4 window classes][head][key][query], log2 domain. The attention mask of
    SwinTransformerBlock.calculate_mask (swinir.py:227-248) only depends on whether a window of the shifted frame lies in the last window row and / or
    column: there a token's region is 1 before row / column ws - shift of the window and 2 from it on, elsewhere 0; attn_mask = -100 where key and
    query regions differ. Class = 2 * (last row) + (last column); class 0 is the plain table."""
    i = torch.arange(ws * ws)
    part = lambda c: torch.where(c < ws - shift, 1, 2)
    out = []
    for cls in range(4):
        rh = part(i // ws) if cls & 2 else torch.zeros_like(i)
        rw = part(i % ws) if cls & 1 else torch.zeros_like(i)
        rid = rh * 3 + rw
        mask = (rid[:, None] != rid[None, :]).float() * (-100.0 * math.log2(math.e))   # [key][query] (symmetric)
        out.append(biasT + mask[None])
    return torch.stack(out).contiguous()


def pack_swin_qkv_tiles(qkv_bits, Cp):
    """The padded qkv weight [3 Cp][Cp] (bf16 bits, rows already in the q | k | v x head x 32 layout) as ring slots of swin_mlp_kernel's qkv
    stage: per 32 output channels one W1-format tile (400-byte rows, k positions in _acc_order, 13 KB), two tiles per 28 KB slot."""
    w = qkv_bits[:, _acc_order(Cp)]
    nq = w.shape[0] // 32
    slots = torch.zeros((nq + 1) // 2, 28672, dtype=torch.uint8)
    for t in range(nq):
        t1 = torch.zeros(32, 200, dtype=torch.int16)
        t1[:, :Cp] = w[32 * t:32 * t + 32]
        slots[t // 2, (t % 2) * 13312:(t % 2) * 13312 + 12800] = t1.reshape(-1).view(torch.uint8)
    return slots.contiguous()


def pack_swinir(sd, cfg):
    C, heads = cfg["embed_dim"], cfg["num_heads"][0]
    hd, Cp = C // heads, heads * 32
    hid = int(C * cfg["mlp_ratio"])
    hid_p = pad_to(hid, 32)
    nf = 64
    r = float(cfg["img_range"])
    out = {}
    # channel maps: qkv rows which*C + h*hd + d -> which*Cp + h*32 + d ; attention output columns h*hd + d -> h*32 + d
    hmap = (torch.arange(C) // hd) * 32 + torch.arange(C) % hd
    qmap = torch.cat([hmap + i * Cp for i in range(3)])
    rpi = swin_relative_position_index(cfg["window_size"])
    out["swin.conv_first.w"] = pack_conv3x3(sd["conv_first.1.weight"], 192, Cp)
    out["swin.conv_first.b"] = pad_vec(sd["conv_first.1.bias"], Cp)
    out["swin.pe_norm.g"], out["swin.pe_norm.b"] = sd["patch_embed.norm.weight"].float(), sd["patch_embed.norm.bias"].float()
    out["swin.norm.g"], out["swin.norm.b"] = sd["norm.weight"].float(), sd["norm.bias"].float()
    for i, depth in enumerate(cfg["depths"]):
        for j in range(depth):
            s, d = f"layers.{i}.residual_group.blocks.{j}.", f"swin.l{i}.b{j}."
            out[d + "n1.g"], out[d + "n1.b"] = sd[s + "norm1.weight"].float(), sd[s + "norm1.bias"].float()
            out[d + "n2.g"], out[d + "n2.b"] = sd[s + "norm2.weight"].float(), sd[s + "norm2.bias"].float()
            out[d + "qkv.w"] = pack_linear(sd[s + "attn.qkv.weight"], 3 * Cp, Cp, row_map=qmap)
            out[d + "qkv.b"] = pad_vec(sd[s + "attn.qkv.bias"], 3 * Cp, idx=qmap)
            out[d + "proj.w"] = pack_linear(sd[s + "attn.proj.weight"], Cp, Cp, col_map=hmap)
            out[d + "proj.b"] = pad_vec(sd[s + "attn.proj.bias"], Cp)
            if Cp == 192:   # fused window attention + projection (swin_attn_proj_kernel): proj input channels in accumulator order, stored
                # in MFMA-fragment order - block (head, channel tile ct, k-step s2) = the 64 lanes' 16-byte pieces back to back (lane l:
                # row 32 ct + (l & 31), columns head * 32 + 16 s2 + 8 (l >> 5) .. + 7), so an operand load reads ONE contiguous KB
                pw = out[d + "proj.w"][:, _acc_order(Cp)]                       # [192 rows][192 permuted columns], bf16 bits
                pw = pw.reshape(6, 32, 6, 2, 2, 8)                             # [ct][r][head][s2][h][8]
                out[d + "proj_t"] = pw.permute(2, 0, 3, 4, 1, 5).contiguous()  # [head][ct][s2][h][r][8] -> lane = h * 32 + r
                if j > 0:   # norm1 + qkv of blocks 1.. of an RSTB run inside the previous block's fused MLP launch
                    out[d + "qkv_t"] = pack_swin_qkv_tiles(out[d + "qkv.w"], Cp)
            out[d + "fc1.w"] = pack_linear(sd[s + "mlp.fc1.weight"], hid_p, Cp)
            out[d + "fc1.b"] = pad_vec(sd[s + "mlp.fc1.bias"], hid_p)
            out[d + "fc2.w"] = pack_linear(sd[s + "mlp.fc2.weight"], Cp, hid_p)
            out[d + "fc2.b"] = pad_vec(sd[s + "mlp.fc2.bias"], Cp)
            if Cp == 192 and hid_p <= 512:   # the fused MLP kernel's tile layout is the 192-channel one and its registers hold <= 512 hidden units;
                # other widths (num_heads != 6, mlp_ratio >= 3 at embed_dim 180) run LN2 / fc1 / fc2 as separate launches
                out[d + "mlp_t"], out[d + "mlp_v"] = pack_swin_mlp(sd[s + "mlp.fc1.weight"], sd[s + "mlp.fc1.bias"], sd[s + "mlp.fc2.weight"],
                                                                  sd[s + "mlp.fc2.bias"], sd[s + "norm2.weight"], sd[s + "norm2.bias"], C, Cp, hid_p)
            table = sd[s + "attn.relative_position_bias_table"].float()
            bias = table[rpi.view(-1)].view(64, 64, heads)                       # [query][key][head]  (swinir.py:138-140)
            out[d + "biasT"] = (bias.permute(2, 1, 0) * math.log2(math.e)).contiguous()  # [head][key][query], log2 domain
            if j % 2 == 1 and Cp == 192:   # shifted block (swinir.py:421: shift_size = 0 if i % 2 == 0 else window_size // 2): mask folded into the tables
                out[d + "biasM"] = swin_masked_bias(out[d + "biasT"], cfg["window_size"], cfg["window_size"] // 2)
        out[f"swin.l{i}.conv.w"] = pack_conv3x3(sd[f"layers.{i}.conv.weight"], Cp, Cp)
        out[f"swin.l{i}.conv.b"] = pad_vec(sd[f"layers.{i}.conv.bias"], Cp)
    out["swin.after_body.w"] = pack_conv3x3(sd["conv_after_body.weight"], Cp, Cp)
    out["swin.after_body.b"] = pad_vec(sd["conv_after_body.bias"], Cp)
    out["swin.before_up.w"] = pack_conv3x3(sd["conv_before_upsample.0.weight"], Cp, nf)
    out["swin.before_up.b"] = pad_vec(sd["conv_before_upsample.0.bias"], nf)
    for src, dst in (("conv_up1", "up1"), ("conv_up2", "up2"), ("conv_up3", "up3"), ("conv_hr", "hr")):
        out[f"swin.{dst}.w"] = pack_conv3x3(sd[src + ".weight"], nf, nf)
        out[f"swin.{dst}.b"] = pad_vec(sd[src + ".bias"], nf)
        if dst != "hr" and nf % 64 == 0:   # F.interpolate(nearest, x2) + conv (swinir.py:880-886): the sub-pixel phase form (conv_halo_kernel<.., PH>)
            out[f"swin.{dst}.wup"] = pack_conv_up2x2(sd[src + ".weight"])
    # conv_last with `x / img_range + mean` (swinir.py:903) folded in
    out["swin.last.w"] = pack_conv3x3(sd["conv_last.weight"], nf, 32, scale=1.0 / r)
    out["swin.last.b"] = pad_vec(sd["conv_last.bias"], 32, scale=1.0 / r, shift=torch.tensor(SWIN_MEAN))
    return out


# ------------------------------------------------------------------------------------------------ VAE (diffusers keys)
def _attn_key(sd, p, new, old, leaf):
    k = f"{p}.{new}.{leaf}"
    return sd[k] if k in sd else sd[f"{p}.{old}.{leaf}"]


def pack_vae(sd, cfg, encoder=True, decoder=True, fp8=False):
    ch, mult, nrb = cfg["ch"], list(cfg["ch_mult"]), cfg["num_res_blocks"]
    nl = len(mult)
    out = {}

    def conv(dst, src, cin_pad=None, cout_pad=None):
        w = sd[src + ".weight"]
        co, ci = w.shape[:2]
        out[dst + ".w"] = pack_conv3x3(w, cin_pad or ci, cout_pad or co)
        out[dst + ".b"] = pad_vec(sd[src + ".bias"], cout_pad or co)

    def norm(dst, src):
        out[dst + ".g"], out[dst + ".b"] = sd[src + ".weight"].float().contiguous(), sd[src + ".bias"].float().contiguous()

    def res(dst, src):
        norm(dst + ".n1", src + ".norm1"); conv(dst + ".c1", src + ".conv1")
        norm(dst + ".n2", src + ".norm2"); conv(dst + ".c2", src + ".conv2")
        if fp8:  # fp8 forms of the two 3x3 convs (the halo kernel's fp8 path takes channel counts that are multiples of 128)
            for cname, sname in ((".c1", ".conv1"), (".c2", ".conv2")):
                w = sd[src + sname + ".weight"]
                if w.shape[0] % 128 == 0 and w.shape[1] % 128 == 0:
                    out[dst + cname + ".w8"], out[dst + cname + ".g8"], out[dst + cname + ".b8"] = pack_conv3x3_fp8(w, sd[src + sname + ".bias"])
        if src + ".conv_shortcut.weight" in sd:
            w = sd[src + ".conv_shortcut.weight"]
            out[dst + ".sc.w"] = pack_linear(w, w.shape[0], w.shape[1])
            out[dst + ".sc.b"] = sd[src + ".conv_shortcut.bias"].float().contiguous()

    def attn(dst, src):
        norm(dst + ".n", src + ".group_norm")
        for a, new, old in (("q", "to_q", "query"), ("k", "to_k", "key"), ("v", "to_v", "value"), ("o", "to_out.0", "proj_attn")):
            w = _attn_key(sd, src, new, old, "weight")
            out[f"{dst}.{a}.w"] = pack_linear(w, w.shape[0], w.shape[0])
            out[f"{dst}.{a}.b"] = _attn_key(sd, src, new, old, "bias").float().contiguous()

    if encoder:
        conv("vae.enc.conv_in", "encoder.conv_in", cin_pad=32)
        for l in range(nl):
            for j in range(nrb):
                res(f"vae.enc.down{l}.res{j}", f"encoder.down_blocks.{l}.resnets.{j}")
            if l != nl - 1:
                conv(f"vae.enc.down{l}.ds", f"encoder.down_blocks.{l}.downsamplers.0.conv")
        res("vae.enc.mid.res0", "encoder.mid_block.resnets.0")
        attn("vae.enc.mid.attn", "encoder.mid_block.attentions.0")
        res("vae.enc.mid.res1", "encoder.mid_block.resnets.1")
        norm("vae.enc.norm_out", "encoder.conv_norm_out")
        conv("vae.enc.conv_out", "encoder.conv_out", cout_pad=32)
        out["vae.quant.w"] = sd["quant_conv.weight"].float().reshape(8, 8).contiguous()
        out["vae.quant.b"] = sd["quant_conv.bias"].float().contiguous()
    if decoder:
        conv("vae.dec.conv_in", "decoder.conv_in", cin_pad=32)
        res("vae.dec.mid.res0", "decoder.mid_block.resnets.0")
        attn("vae.dec.mid.attn", "decoder.mid_block.attentions.0")
        res("vae.dec.mid.res1", "decoder.mid_block.resnets.1")
        for i in range(nl):  # diffusers up_blocks.i  <->  ldm up[nl-1-i]
            l = nl - 1 - i
            for j in range(nrb + 1):
                res(f"vae.dec.up{l}.res{j}", f"decoder.up_blocks.{i}.resnets.{j}")
            if l != 0:
                conv(f"vae.dec.up{l}.us", f"decoder.up_blocks.{i}.upsamplers.0.conv")
                wu = sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"]
                if wu.shape[0] % 128 == 0 and wu.shape[1] % 128 == 0:   # the shapes conv_halo_s1_kernel takes
                    out[f"vae.dec.up{l}.us.wup"] = pack_conv_up2x2(wu)
        norm("vae.dec.norm_out", "decoder.conv_norm_out")
        conv("vae.dec.conv_out", "decoder.conv_out", cout_pad=32)
        out["vae.post_quant.w"] = sd["post_quant_conv.weight"].float().reshape(4, 4).contiguous()
        out["vae.post_quant.b"] = sd["post_quant_conv.bias"].float().contiguous()
    return out


# ------------------------------------------------------------------------------------------------ DiT (diffusers keys)
def _dit_block_keys(p):
    keys = [p + "scale_shift_table"]
    for a in ("attn1", "attn2"):
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            keys += [p + f"{a}.{n}.weight", p + f"{a}.{n}.bias"]
    return keys + [p + "ff.net.0.proj.weight", p + "ff.net.0.proj.bias", p + "ff.net.2.weight", p + "ff.net.2.bias"]


def dit_expected_keys(cfg):
    keys = []
    for base in ("pos_embed.proj", "caption_projection.linear_1", "caption_projection.linear_2", "adaln_single.emb.timestep_embedder.linear_1",
                 "adaln_single.emb.timestep_embedder.linear_2", "adaln_single.linear", "proj_out"):
        keys += [base + ".weight", base + ".bias"]
    keys.append("scale_shift_table")
    if cfg.get("micro"):
        for e in ("resolution_embedder", "aspect_ratio_embedder"):
            for l in ("linear_1", "linear_2"):
                keys += [f"adaln_single.emb.{e}.{l}.weight", f"adaln_single.emb.{e}.{l}.bias"]
    kvc = cfg.get("kv_compress") or {}
    for d in range(cfg["num_layers"]):
        p = f"transformer_blocks.{d}."
        keys += _dit_block_keys(p)
        if kvc.get("sampling") == "conv" and d in kvc.get("layers", ()) and int(kvc.get("scale_factor", 1)) > 1:
            keys += [p + "attn1.sr.weight", p + "attn1.sr.bias", p + "attn1.norm.weight", p + "attn1.norm.bias"]
        if cfg.get("qk_norm"):
            keys += [p + "attn1.q_norm.weight", p + "attn1.q_norm.bias", p + "attn1.k_norm.weight", p + "attn1.k_norm.bias"]
    return keys


def dit_control_expected_keys(copy_blocks_num):
    """Parameters ControlTransformerHalf adds to its base model (transformer_controlnet.py:19-39,62-76)."""
    keys = []
    for i in range(copy_blocks_num):
        keys += _dit_block_keys(f"controlnet.{i}.copied_block.")
        for n in (("before_proj",) if i == 0 else ()) + ("after_proj",):
            keys += [f"controlnet.{i}.{n}.weight", f"controlnet.{i}.{n}.bias"]
    return keys


def _pack_lin(out, dst, w, b, k_pad=None, n_pad=None):
    w = w.reshape(w.shape[0], -1)
    out[dst + ".w"] = pack_linear(w, n_pad or w.shape[0], k_pad or w.shape[1])
    out[dst + ".b"] = pad_vec(b, n_pad or w.shape[0])


def _pack_dit_block(out, sd, s, p):
    """One BasicTransformerBlock: diffusers keys under prefix s -> device tensors under prefix p (q|k|v and k|v fused)."""
    out[p + "sst"] = sd[s + "scale_shift_table"].float().contiguous()
    _pack_lin(out, p + "qkv", torch.cat([sd[s + f"attn1.{n}.weight"] for n in ("to_q", "to_k", "to_v")], 0),
              torch.cat([sd[s + f"attn1.{n}.bias"] for n in ("to_q", "to_k", "to_v")], 0))
    _pack_lin(out, p + "ao", sd[s + "attn1.to_out.0.weight"], sd[s + "attn1.to_out.0.bias"])
    _pack_lin(out, p + "cq", sd[s + "attn2.to_q.weight"], sd[s + "attn2.to_q.bias"])
    _pack_lin(out, p + "ckv", torch.cat([sd[s + "attn2.to_k.weight"], sd[s + "attn2.to_v.weight"]], 0),
              torch.cat([sd[s + "attn2.to_k.bias"], sd[s + "attn2.to_v.bias"]], 0))
    _pack_lin(out, p + "co", sd[s + "attn2.to_out.0.weight"], sd[s + "attn2.to_out.0.bias"])
    _pack_lin(out, p + "fc1", sd[s + "ff.net.0.proj.weight"], sd[s + "ff.net.0.proj.bias"])
    _pack_lin(out, p + "fc2", sd[s + "ff.net.2.weight"], sd[s + "ff.net.2.bias"])


def pack_dit(sd, cfg):
    out = {}
    _pack_lin(out, "dit.patch", sd["pos_embed.proj.weight"], sd["pos_embed.proj.bias"], k_pad=32)  # k = c*4 + p*2 + q
    _pack_lin(out, "dit.cap1", sd["caption_projection.linear_1.weight"], sd["caption_projection.linear_1.bias"])
    _pack_lin(out, "dit.cap2", sd["caption_projection.linear_2.weight"], sd["caption_projection.linear_2.bias"])
    _pack_lin(out, "dit.final", sd["proj_out.weight"], sd["proj_out.bias"], n_pad=32)
    for dst, src in (("dit.temb1", "adaln_single.emb.timestep_embedder.linear_1"), ("dit.temb2", "adaln_single.emb.timestep_embedder.linear_2"),
                     ("dit.tblock", "adaln_single.linear")):
        out[dst + ".w"], out[dst + ".b"] = sd[src + ".weight"].float().contiguous(), sd[src + ".bias"].float().contiguous()
    out["dit.final_sst"] = sd["scale_shift_table"].float().contiguous()
    if cfg.get("micro"):
        # micro-conditioning (sample_size 128; diffusers PixArtAlphaCombinedTimestepSizeEmbeddings, in-tree SizeEmbedder PixArt_blocks.py:366-399): the two
        # embedders' first linears and the weights of their second ones; the second linears' biases ride in the timestep embedder's (the library adds
        # W2 h onto the slices [0:S), [S:2S) - the resolution embedder twice, for height and width - and [2S:3S) of emb = temb2(...))
        r, a = "adaln_single.emb.resolution_embedder.", "adaln_single.emb.aspect_ratio_embedder."
        for dst, src in (("dit.res1", r + "linear_1"), ("dit.ar1", a + "linear_1")):
            out[dst + ".w"], out[dst + ".b"] = sd[src + ".weight"].float().contiguous(), sd[src + ".bias"].float().contiguous()
        out["dit.res2.w"], out["dit.ar2.w"] = sd[r + "linear_2.weight"].float().contiguous(), sd[a + "linear_2.weight"].float().contiguous()
        out["dit.temb2.b"] = (out["dit.temb2.b"] + torch.cat([sd[r + "linear_2.bias"], sd[r + "linear_2.bias"], sd[a + "linear_2.bias"]]).float()).contiguous()
    kvc = cfg.get("kv_compress") or {}
    C = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    for d in range(cfg["num_layers"]):
        s, p = f"transformer_blocks.{d}.", f"dit.l{d}."
        _pack_dit_block(out, sd, s, p)
        # optional branches of the self-attention (AttentionKVCompress, PixArt_blocks.py:60-158; keys as pixart_to_diffusers names them)
        if d in kvc.get("layers", ()) and int(kvc.get("scale_factor", 1)) > 1:
            r = int(kvc["scale_factor"])
            if kvc["sampling"] == "conv":   # depthwise r x r / stride r conv + LayerNorm
                out[p + "kvc_w"] = sd[s + "attn1.sr.weight"].float().reshape(C, r * r).contiguous()
                out[p + "kvc_b"] = sd[s + "attn1.sr.bias"].float().contiguous()
                out[p + "kvc_g"], out[p + "kvc_beta"] = sd[s + "attn1.norm.weight"].float().contiguous(), sd[s + "attn1.norm.bias"].float().contiguous()
            elif kvc["sampling"] in ("uniform", "ave"):   # every r-th row / column ('ave' = F.interpolate(nearest) picks the same tokens): weight 1 on the first tap
                w = torch.zeros(C, r * r)
                w[:, 0] = 1.0
                out[p + "kvc_w"], out[p + "kvc_b"] = w, torch.zeros(C)
            else:
                raise NotImplementedError(f"kv_compress sampling {kvc['sampling']!r} (the MI355X path offers conv, uniform, ave)")
        if cfg.get("qk_norm"):
            for dst, src in (("qn", "q_norm"), ("kn", "k_norm")):
                out[p + dst + "_g"], out[p + dst + "_b"] = sd[s + f"attn1.{src}.weight"].float().contiguous(), sd[s + f"attn1.{src}.bias"].float().contiguous()
    return out


def pack_dit_control(sd, copy_blocks_num):
    """controlnet.{i}.copied_block.* / .after_proj / controlnet.0.before_proj -> dit.ctrl{i}.* (ir_dit_control_configure)."""
    out = {}
    for i in range(copy_blocks_num):
        _pack_dit_block(out, sd, f"controlnet.{i}.copied_block.", f"dit.ctrl{i}.")
        _pack_lin(out, f"dit.ctrl{i}.after", sd[f"controlnet.{i}.after_proj.weight"], sd[f"controlnet.{i}.after_proj.bias"])
    _pack_lin(out, "dit.ctrl0.before", sd["controlnet.0.before_proj.weight"], sd["controlnet.0.before_proj.bias"])
    return out


# ------------------------------------------------------------------------------------------------ T5 v1.1 encoder (prompt producer)
def t5_expected_keys(cfg):
    keys = ["shared.weight", "encoder.final_layer_norm.weight", "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
    for i in range(cfg["num_layers"]):
        p = f"encoder.block.{i}.layer."
        keys += [p + f"0.SelfAttention.{n}.weight" for n in ("q", "k", "v", "o")] + [p + "0.layer_norm.weight", p + "1.layer_norm.weight"]
        keys += [p + f"1.DenseReluDense.{n}.weight" for n in ("wi_0", "wi_1", "wo")]
    return keys


def pack_t5(sd, cfg):
    """transformers T5EncoderModel keys -> device tensors of ir_t5_configure (q|k|v and wi_0|wi_1 fused; zero biases)."""
    out = {"t5.embed": _bf16(sd["shared.weight"].to(torch.float32)), "t5.final_ln": sd["encoder.final_layer_norm.weight"].float().contiguous()}
    for i in range(cfg["num_layers"]):
        s_, p = f"encoder.block.{i}.layer.", f"t5.l{i}."
        out[p + "ln1"] = sd[s_ + "0.layer_norm.weight"].float().contiguous()
        out[p + "ln2"] = sd[s_ + "1.layer_norm.weight"].float().contiguous()
        for dst, w in ((p + "qkv", torch.cat([sd[s_ + f"0.SelfAttention.{n}.weight"] for n in ("q", "k", "v")], 0)),
                       (p + "o", sd[s_ + "0.SelfAttention.o.weight"]),
                       (p + "wi", torch.cat([sd[s_ + "1.DenseReluDense.wi_0.weight"], sd[s_ + "1.DenseReluDense.wi_1.weight"]], 0)),
                       (p + "wo", sd[s_ + "1.DenseReluDense.wo.weight"])):
            _pack_lin(out, dst, w, torch.zeros(w.shape[0]))
    return out


def t5_position_bias(table, T, num_buckets=32, max_distance=128):
    """relative_attention_bias.weight [buckets, heads] -> additive bias [heads, T, T] (query, key), computed on the host as
    transformers' T5Attention.compute_bias does (bidirectional buckets: exact up to +-7, log-spaced to max_distance, saturating)."""
    pos = torch.arange(T)
    rel = pos[None, :] - pos[:, None]
    nb = num_buckets // 2
    ret = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float().clamp(min=1) / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    buckets = ret + torch.where(n < max_exact, n, large)
    return table.to(torch.float32)[buckets].permute(2, 0, 1).contiguous()


def sincos_pos_embed(embed_dim, gh, gw, base_size, interpolation_scale=1.0):
    """2-D sin-cos table regenerated per latent size, as the reference does on the host with numpy
    (PixArtMS.py:177-182 / PixArt.py:258-307; diffusers PatchEmbed cropped/regenerated table). fp32 [gh*gw, D]."""
    grid_h = np.arange(gh, dtype=np.float32) / (gh / base_size) / interpolation_scale
    grid_w = np.arange(gw, dtype=np.float32) / (gw / base_size) / interpolation_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, gw, gh])

    def one(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        o = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(o), np.cos(o)], axis=1)

    emb = np.concatenate([one(embed_dim // 2, grid[0]), one(embed_dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb).to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------ checkpoint shape tables
def swinir_shapes(cfg):
    """name -> shape of the reference SwinIR parameters (buffers relative_position_index / attn_mask are derived, not stored)."""
    C, ws, hid = cfg["embed_dim"], cfg["window_size"], int(cfg["embed_dim"] * cfg["mlp_ratio"])
    s = {"conv_first.1.weight": (C, 192, 3, 3), "conv_first.1.bias": (C,), "patch_embed.norm.weight": (C,), "patch_embed.norm.bias": (C,)}
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


def vae_shapes(cfg):
    ch, mult, nrb, z = cfg["ch"], list(cfg["ch_mult"]), cfg["num_res_blocks"], cfg.get("z_channels", 4)
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


def dit_shapes(cfg):
    C = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    cap, mlp = cfg["caption_channels"], cfg.get("mlp", 4 * C)
    s = {"pos_embed.proj.weight": (C, 4, 2, 2), "pos_embed.proj.bias": (C,),
         "caption_projection.linear_1.weight": (C, cap), "caption_projection.linear_1.bias": (C,),
         "caption_projection.linear_2.weight": (C, C), "caption_projection.linear_2.bias": (C,),
         "adaln_single.emb.timestep_embedder.linear_1.weight": (C, 256), "adaln_single.emb.timestep_embedder.linear_1.bias": (C,),
         "adaln_single.emb.timestep_embedder.linear_2.weight": (C, C), "adaln_single.emb.timestep_embedder.linear_2.bias": (C,),
         "adaln_single.linear.weight": (6 * C, C), "adaln_single.linear.bias": (6 * C,),
         "proj_out.weight": (32, C), "proj_out.bias": (32,), "scale_shift_table": (2, C)}
    if cfg.get("micro"):   # use_additional_conditions (sample_size 128): two TimestepEmbedding(256 -> C / 3) stacks
        S = C // 3
        for e in ("resolution_embedder", "aspect_ratio_embedder"):
            s[f"adaln_single.emb.{e}.linear_1.weight"], s[f"adaln_single.emb.{e}.linear_1.bias"] = (S, 256), (S,)
            s[f"adaln_single.emb.{e}.linear_2.weight"], s[f"adaln_single.emb.{e}.linear_2.bias"] = (S, S), (S,)
    kvc = cfg.get("kv_compress") or {}
    for d in range(cfg["num_layers"]):
        p = f"transformer_blocks.{d}."
        if kvc.get("sampling") == "conv" and d in kvc.get("layers", ()) and int(kvc.get("scale_factor", 1)) > 1:
            r = int(kvc["scale_factor"])
            s[p + "attn1.sr.weight"], s[p + "attn1.sr.bias"], s[p + "attn1.norm.weight"], s[p + "attn1.norm.bias"] = (C, 1, r, r), (C,), (C,), (C,)
        if cfg.get("qk_norm"):
            for nn_ in ("q_norm", "k_norm"):
                s[p + f"attn1.{nn_}.weight"], s[p + f"attn1.{nn_}.bias"] = (C,), (C,)
        s[p + "scale_shift_table"] = (6, C)
        for a in ("attn1", "attn2"):
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                s[p + f"{a}.{n}.weight"], s[p + f"{a}.{n}.bias"] = (C, C), (C,)
        s[p + "ff.net.0.proj.weight"], s[p + "ff.net.0.proj.bias"] = (mlp, C), (mlp,)
        s[p + "ff.net.2.weight"], s[p + "ff.net.2.bias"] = (C, mlp), (C,)
    return s


# ------------------------------------------------------------------------------------------------ SD-UNet / ControlNet (ControlLDM, N4)
def unet_layout(cfg, control=False):
    """Block structure UNetModel.__init__ / ControlNet.__init__ build (openaimodel.py:520-700, cldm.py:143-274) for the options of
    configs/cldm.yaml: per input / output block the list of layers ('conv', cin, cout) | ('res', cin, cout) | ('xf', ch) | ('down', ch) |
    ('up', ch); returns (input_blocks, middle channels, output_blocks, channels of the 12 skips)."""
    mc, mult, nrb = cfg["model_channels"], list(cfg["channel_mult"]), cfg["num_res_blocks"]
    att = set(cfg["attention_resolutions"])
    inb = [[("conv", cfg["in_channels"] + (cfg["hint_channels"] if control else 0), mc)]]
    chans, ch, ds = [mc], mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            inb.append([("res", ch, m * mc)] + ([("xf", m * mc)] if ds in att else []))
            ch = m * mc
            chans.append(ch)
        if level != len(mult) - 1:
            inb.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    mid, skips, outb = ch, list(chans), []
    if not control:
        for level in range(len(mult) - 1, -1, -1):
            for i in range(nrb + 1):
                layers = [("res", ch + chans.pop(), mc * mult[level])]
                ch = mc * mult[level]
                if ds in att:
                    layers.append(("xf", ch))
                if level and i == nrb:
                    layers.append(("up", ch))
                    ds //= 2
                outb.append(layers)
    return inb, mid, outb, skips


def _unet_walk(cfg, control):
    """(reference parameter prefix, device tensor prefix, layer tuple) of every layer."""
    inb, mid, outb, skips = unet_layout(cfg, control)
    P = "cnet" if control else "unet"
    leaf = {"conv": "conv", "res": "res", "xf": "xf", "down": "down", "up": "up"}
    for i, layers in enumerate(inb):
        for k, L in enumerate(layers):
            yield f"input_blocks.{i}.{k}", f"{P}.in{i}.{leaf[L[0]]}", L
    for k, L in enumerate([("res", mid, mid), ("xf", mid), ("res", mid, mid)]):
        yield f"middle_block.{k}", f"{P}.mid{k}.{leaf[L[0]]}", L
    for i, layers in enumerate(outb):
        for k, L in enumerate(layers):
            yield f"output_blocks.{i}.{k}", f"{P}.out{i}.{leaf[L[0]]}", L


def unet_expected_keys(cfg, control=False):
    keys = [f"time_embed.{i}.{t}" for i in (0, 2) for t in ("weight", "bias")]
    wb = ("weight", "bias")
    for src, _, L in _unet_walk(cfg, control):
        if L[0] == "conv":
            keys += [f"{src}.{t}" for t in wb]
        elif L[0] == "res":
            keys += [f"{src}.{n}.{t}" for n in ("in_layers.0", "in_layers.2", "emb_layers.1", "out_layers.0", "out_layers.3") for t in wb]
            if L[1] != L[2]:
                keys += [f"{src}.skip_connection.{t}" for t in wb]
        elif L[0] == "xf":
            keys += [f"{src}.{n}.{t}" for n in ("norm", "proj_in", "proj_out") for t in wb]
            tb = f"{src}.transformer_blocks.0"
            for a in ("attn1", "attn2"):
                keys += [f"{tb}.{a}.{n}.weight" for n in ("to_q", "to_k", "to_v", "to_out.0")] + [f"{tb}.{a}.to_out.0.bias"]
            keys += [f"{tb}.{n}.{t}" for n in ("ff.net.0.proj", "ff.net.2", "norm1", "norm2", "norm3") for t in wb]
        elif L[0] == "down":
            keys += [f"{src}.op.{t}" for t in wb]
        elif L[0] == "up":
            keys += [f"{src}.conv.{t}" for t in wb]
    if control:
        skips = unet_layout(cfg, True)[3]
        keys += [f"zero_convs.{i}.0.{t}" for i in range(len(skips)) for t in wb] + [f"middle_block_out.0.{t}" for t in wb]
    else:
        keys += [f"out.{i}.{t}" for i in (0, 2) for t in wb]
    return keys


def pack_unet(sd, cfg, control=False):
    """UNetModel / ControlNet state dict (reference names) -> the device tensors ir_unet_configure binds (include/instarevive_hip.h)."""
    P = "cnet" if control else "unet"
    out = {}
    for dst, src in ((f"{P}.temb1", "time_embed.0"), (f"{P}.temb2", "time_embed.2")):
        out[dst + ".w"], out[dst + ".b"] = sd[src + ".weight"].float().contiguous(), sd[src + ".bias"].float().contiguous()

    def conv3(dst, src, cin_pad=None, cout_pad=None):
        w, b = sd[src + ".weight"], sd[src + ".bias"]
        out[dst + ".w"] = pack_conv3x3(w, cin_pad or w.shape[1], cout_pad or w.shape[0])
        out[dst + ".b"] = pad_vec(b, cout_pad or w.shape[0])

    def norm(dst, src):
        out[dst + ".g"], out[dst + ".b"] = sd[src + ".weight"].float().contiguous(), sd[src + ".bias"].float().contiguous()

    for src, dst, L in _unet_walk(cfg, control):
        if L[0] == "conv":
            conv3(dst, src, cin_pad=32)
        elif L[0] == "res":
            norm(dst + ".n1", src + ".in_layers.0")
            conv3(dst + ".c1", src + ".in_layers.2")
            out[dst + ".emb.w"] = sd[src + ".emb_layers.1.weight"].float().contiguous()
            out[dst + ".emb.b"] = (sd[src + ".emb_layers.1.bias"].float() + sd[src + ".in_layers.2.bias"].float()).contiguous()
            norm(dst + ".n2", src + ".out_layers.0")
            conv3(dst + ".c2", src + ".out_layers.3")
            if L[1] != L[2]:
                _pack_lin(out, dst + ".sc", sd[src + ".skip_connection.weight"], sd[src + ".skip_connection.bias"])
        elif L[0] == "xf":
            c = L[1]
            tb = src + ".transformer_blocks.0."
            zeros = torch.zeros
            norm(dst + ".gn", src + ".norm")
            _pack_lin(out, dst + ".pin", sd[src + ".proj_in.weight"], sd[src + ".proj_in.bias"])
            _pack_lin(out, dst + ".pout", sd[src + ".proj_out.weight"], sd[src + ".proj_out.bias"])
            for i in (1, 2, 3):
                norm(f"{dst}.ln{i}", f"{tb}norm{i}")
            _pack_lin(out, dst + ".qkv", torch.cat([sd[tb + f"attn1.{n}.weight"] for n in ("to_q", "to_k", "to_v")], 0), zeros(3 * c))
            _pack_lin(out, dst + ".ao", sd[tb + "attn1.to_out.0.weight"], sd[tb + "attn1.to_out.0.bias"])
            _pack_lin(out, dst + ".cq", sd[tb + "attn2.to_q.weight"], zeros(c))
            _pack_lin(out, dst + ".ckv", torch.cat([sd[tb + "attn2.to_k.weight"], sd[tb + "attn2.to_v.weight"]], 0), zeros(2 * c))
            _pack_lin(out, dst + ".co", sd[tb + "attn2.to_out.0.weight"], sd[tb + "attn2.to_out.0.bias"])
            _pack_lin(out, dst + ".ff1", sd[tb + "ff.net.0.proj.weight"], sd[tb + "ff.net.0.proj.bias"])
            _pack_lin(out, dst + ".ff2", sd[tb + "ff.net.2.weight"], sd[tb + "ff.net.2.bias"])
        elif L[0] == "down":
            conv3(dst, src + ".op")
        elif L[0] == "up":
            conv3(dst, src + ".conv")
    if control:
        skips = unet_layout(cfg, True)[3]
        for i in range(len(skips)):
            _pack_lin(out, f"{P}.zero{i}", sd[f"zero_convs.{i}.0.weight"], sd[f"zero_convs.{i}.0.bias"])
        _pack_lin(out, f"{P}.midzero", sd["middle_block_out.0.weight"], sd["middle_block_out.0.bias"])
    else:
        norm(f"{P}.out.norm", "out.0")
        conv3(f"{P}.out.conv", "out.2", cout_pad=32)
    return out


def vae_ldm_to_diffusers(sd, n_levels=4, num_res_blocks=2):
    """Parameter names of the LDM AutoencoderKL (ldm/models/autoencoder.py + ldm/modules/diffusionmodules/model.py: `encoder.down.0.block.0.
    norm1.weight`, `decoder.up.3.upsample.conv.weight`, `encoder.mid.attn_1.q.weight` [C,C,1,1], ...) -> the diffusers names models.
    AutoencoderKL loads. Both halves optional (the cond_encoder of cldm.py:476-480 has an encoder and quant_conv only)."""
    out = {}
    res_pairs = (("norm1", "norm1"), ("conv1", "conv1"), ("norm2", "norm2"), ("conv2", "conv2"), ("nin_shortcut", "conv_shortcut"))
    attn_pairs = (("norm", "group_norm"), ("q", "to_q"), ("k", "to_k"), ("v", "to_v"), ("proj_out", "to_out.0"))

    def mv(src, dst, flatten=False):
        for t in ("weight", "bias"):
            if f"{src}.{t}" in sd:
                v = sd[f"{src}.{t}"]
                out[f"{dst}.{t}"] = v.reshape(v.shape[0], -1) if flatten and t == "weight" else v

    def res(src, dst):
        for a, b in res_pairs:
            mv(f"{src}.{a}", f"{dst}.{b}")

    for half in ("encoder", "decoder"):
        if f"{half}.conv_in.weight" not in sd:
            continue
        mv(f"{half}.conv_in", f"{half}.conv_in"); mv(f"{half}.conv_out", f"{half}.conv_out"); mv(f"{half}.norm_out", f"{half}.conv_norm_out")
        res(f"{half}.mid.block_1", f"{half}.mid_block.resnets.0"); res(f"{half}.mid.block_2", f"{half}.mid_block.resnets.1")
        for a, b in attn_pairs:
            mv(f"{half}.mid.attn_1.{a}", f"{half}.mid_block.attentions.0.{b}", flatten=a != "norm")
        for l in range(n_levels):
            if half == "encoder":
                for j in range(num_res_blocks):
                    res(f"encoder.down.{l}.block.{j}", f"encoder.down_blocks.{l}.resnets.{j}")
                mv(f"encoder.down.{l}.downsample.conv", f"encoder.down_blocks.{l}.downsamplers.0.conv")
            else:
                i = n_levels - 1 - l
                for j in range(num_res_blocks + 1):
                    res(f"decoder.up.{l}.block.{j}", f"decoder.up_blocks.{i}.resnets.{j}")
                mv(f"decoder.up.{l}.upsample.conv", f"decoder.up_blocks.{i}.upsamplers.0.conv")
    mv("quant_conv", "quant_conv"); mv("post_quant_conv", "post_quant_conv")
    return out


def unet_shapes(cfg, control=False):
    """{parameter name: shape} of the reference's UNetModel / ControlNet for cfg (synthetic weights of bench tools)."""
    mc, temb, ctx = cfg["model_channels"], 4 * cfg["model_channels"], cfg["context_dim"]
    s = {"time_embed.0.weight": (temb, mc), "time_embed.0.bias": (temb,), "time_embed.2.weight": (temb, temb), "time_embed.2.bias": (temb,)}

    def put(p, *shape):
        s[p + ".weight"], s[p + ".bias"] = tuple(shape), (shape[0],)

    for src, _, L in _unet_walk(cfg, control):
        if L[0] == "conv":
            put(src, L[2], L[1], 3, 3)
        elif L[0] == "res":
            put(src + ".in_layers.0", L[1]); put(src + ".in_layers.2", L[2], L[1], 3, 3); put(src + ".emb_layers.1", L[2], temb)
            put(src + ".out_layers.0", L[2]); put(src + ".out_layers.3", L[2], L[2], 3, 3)
            if L[1] != L[2]:
                put(src + ".skip_connection", L[2], L[1], 1, 1)
        elif L[0] == "xf":
            c, tb = L[1], src + ".transformer_blocks.0"
            put(src + ".norm", c); put(src + ".proj_in", c, c); put(src + ".proj_out", c, c)
            for a, kd in (("attn1", c), ("attn2", ctx)):
                s[f"{tb}.{a}.to_q.weight"], s[f"{tb}.{a}.to_k.weight"], s[f"{tb}.{a}.to_v.weight"] = (c, c), (c, kd), (c, kd)
                put(f"{tb}.{a}.to_out.0", c, c)
            put(tb + ".ff.net.0.proj", 8 * c, c); put(tb + ".ff.net.2", c, 4 * c)
            for n in ("norm1", "norm2", "norm3"):
                put(f"{tb}.{n}", c)
        elif L[0] == "down":
            put(src + ".op", L[1], L[1], 3, 3)
        elif L[0] == "up":
            put(src + ".conv", L[1], L[1], 3, 3)
    if control:
        skips = unet_layout(cfg, True)[3]
        for i, c in enumerate(skips):
            put(f"zero_convs.{i}.0", c, c, 1, 1)
        put("middle_block_out.0", skips[-1], skips[-1], 1, 1)
    else:
        put("out.0", mc); put("out.2", cfg["out_channels"], mc, 3, 3)
    return s


# ------------------------------------------------------------------------------------------------ OpenCLIP text tower (ControlLDM's cond_stage_model)
def clip_text_expected_keys(cfg):
    keys = ["token_embedding.weight", "positional_embedding", "ln_final.weight", "ln_final.bias"]
    for i in range(cfg["layers"]):
        p = f"transformer.resblocks.{i}"
        keys += [p + s for s in (".ln_1.weight", ".ln_1.bias", ".ln_2.weight", ".ln_2.bias", ".attn.in_proj_weight", ".attn.in_proj_bias",
                                 ".attn.out_proj.weight", ".attn.out_proj.bias", ".mlp.c_fc.weight", ".mlp.c_fc.bias", ".mlp.c_proj.weight", ".mlp.c_proj.bias")]
    return keys


def clip_text_shapes(cfg):
    d, f = cfg["width"], int(cfg["width"] * cfg.get("mlp_ratio", 4.0))
    s = {"token_embedding.weight": (cfg["vocab_size"], d), "positional_embedding": (cfg["context_length"], d), "ln_final.weight": (d,), "ln_final.bias": (d,)}
    for i in range(cfg["layers"]):
        p = f"transformer.resblocks.{i}"
        s.update({p + ".ln_1.weight": (d,), p + ".ln_1.bias": (d,), p + ".ln_2.weight": (d,), p + ".ln_2.bias": (d,),
                  p + ".attn.in_proj_weight": (3 * d, d), p + ".attn.in_proj_bias": (3 * d,), p + ".attn.out_proj.weight": (d, d), p + ".attn.out_proj.bias": (d,),
                  p + ".mlp.c_fc.weight": (f, d), p + ".mlp.c_fc.bias": (f,), p + ".mlp.c_proj.weight": (d, f), p + ".mlp.c_proj.bias": (d,)})
    return s


def pack_clip_text(sd, cfg, n_run):
    """open_clip text-tower parameters -> the tensors ir_clip_text_configure binds. The softmax scale d_head^-0.5 is folded into the q rows of
    in_proj (the attention kernel, shared with the T5 encoder, does not scale); the causal mask is an additive [heads][T][T] table."""
    d, H, T = cfg["width"], cfg["heads"], cfg["context_length"]
    out = {"clip.embed": _bf16(sd["token_embedding.weight"]), "clip.pos": sd["positional_embedding"].float().contiguous(),
           "clip.final_ln.g": sd["ln_final.weight"].float().contiguous(), "clip.final_ln.b": sd["ln_final.bias"].float().contiguous()}
    mask = torch.zeros(T, T)
    mask[torch.triu(torch.ones(T, T, dtype=torch.bool), 1)] = -3.0e38
    out["clip.causal"] = mask[None].expand(H, T, T).contiguous()
    scale = (d // H) ** -0.5
    for i in range(n_run):
        s, p = f"transformer.resblocks.{i}", f"clip.l{i}"
        for n, src in (("ln1", ".ln_1"), ("ln2", ".ln_2")):
            out[f"{p}.{n}.g"], out[f"{p}.{n}.b"] = sd[s + src + ".weight"].float().contiguous(), sd[s + src + ".bias"].float().contiguous()
        w, b = sd[s + ".attn.in_proj_weight"].float().clone(), sd[s + ".attn.in_proj_bias"].float().clone()
        w[:d] *= scale
        b[:d] *= scale
        _pack_lin(out, p + ".qkv", w, b)
        _pack_lin(out, p + ".o", sd[s + ".attn.out_proj.weight"], sd[s + ".attn.out_proj.bias"])
        _pack_lin(out, p + ".fc", sd[s + ".mlp.c_fc.weight"], sd[s + ".mlp.c_fc.bias"])
        _pack_lin(out, p + ".proj", sd[s + ".mlp.c_proj.weight"], sd[s + ".mlp.c_proj.bias"])
    return out
