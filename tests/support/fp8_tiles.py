"""Test infrastructure: decode the e4m3 tile images attn_fp8_prep_kernel writes (instarevive_amd/csrc/attn_fp8.hip) and restate the
kernel's quantisation of Q in PyTorch, so the fp8 attention kernel can be checked against an fp64 softmax over exactly the operands
its MFMAs saw. Layout (per batch, head, 64-key tile; 10240 bytes):
    K8    64 rows x 80 B: bytes 0..63 = e4m3(K[key][d] / 2^(bk - 127)), d = 0..63; bytes 64..79 = K[key][64..71] as raw bf16
    V8^T  80 rows x 64 B: row d < 72, logical byte L = 32 h + j <-> key 32 (j >> 4) + (j & 3) + 8 ((j & 15) >> 2) + 4 h, value
          e4m3(V[key][d] / 2^(bv - 127)); logical 16-byte chunk c is stored at chunk c ^ ((d >> 2) & 3); row 72 = ones;
          row 79 starts with the bytes bk, bv (E8M0 exponents of the tile)."""
import torch

TILE_BYTES, K_BYTES, KROW = 10240, 5120, 80


def _key_of_logical():
    L = torch.arange(64)
    h, j = L >> 5, L & 31
    return 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * h


def decode_tiles(ws_u8, b, heads, t):
    """ws_u8: uint8 tensor holding the tile images at its start -> (K, V) dequantised, float32 [b][heads][t][72], on ws_u8's device."""
    nt = t // 64
    img = ws_u8[: b * heads * nt * TILE_BYTES].view(b, heads, nt, TILE_BYTES)
    dev = img.device
    kpart = img[..., :K_BYTES].reshape(b, heads, nt, 64, KROW)
    vpart = img[..., K_BYTES:].reshape(b, heads, nt, 80, 64)
    bk = vpart[..., 79, 0].to(torch.int32)
    bv = vpart[..., 79, 1].to(torch.int32)
    sk = torch.pow(2.0, (bk - 127).float())[..., None, None]
    sv = torch.pow(2.0, (bv - 127).float())[..., None, None]
    k8 = kpart[..., :64].contiguous().view(torch.float8_e4m3fn).float() * sk
    kr = kpart[..., 64:80].contiguous().view(torch.bfloat16).float()
    K = torch.cat([k8, kr], dim=-1)                                            # [b][heads][nt][64][72]
    d = torch.arange(72, device=dev)
    pc = torch.arange(4, device=dev)
    lc = pc[None, :] ^ ((d[:, None] >> 2) & 3)                                 # [72][4] logical chunk stored at physical chunk pc
    rows = vpart[..., :72, :].reshape(b, heads, nt, 72, 4, 16)                 # physical chunks
    logical = torch.empty_like(rows)
    logical.scatter_(4, lc[None, None, None, :, :, None].expand(b, heads, nt, 72, 4, 16), rows)
    v8 = logical.reshape(b, heads, nt, 72, 64).contiguous().view(torch.float8_e4m3fn).float() * sv   # [..][d][logical byte]
    key = _key_of_logical().to(dev)
    Vt = torch.empty_like(v8)
    Vt[..., key] = v8                                                          # [..][d][key]
    assert bool((vpart[..., 72, :] == 0x38).all()), "ones row"
    return K.reshape(b, heads, t, 72), Vt.transpose(-1, -2).reshape(b, heads, t, 72)


def quantise_q(q_bf16, scale_log2):
    """q_bf16: [..., 72] bfloat16 -> what the kernel multiplies with: d 0..63 scaled by scale * log2(e) in fp32, e4m3 per 32-d block with
    exponent floor(log2 max) - 7 (blocks as the MFMA defines them, see below); d 64..71 scaled and rounded to bf16."""
    qs = q_bf16.float() * scale_log2
    # the MFMA's scale block b is bytes 16b .. 16b+15 of both lane halves: d in [16b, 16b + 16) and [32 + 16b, 48 + 16b)
    blocks = qs[..., :64].reshape(*qs.shape[:-1], 2, 2, 16)                     # [half h][block b][16]
    mx = blocks.abs().amax(dim=(-3, -1), keepdim=True)
    byte = ((mx.view(torch.int32) >> 23) - 7).clamp(1, 254)
    s = torch.pow(2.0, (byte - 127).float())
    q8 = (blocks / s).to(torch.float8_e4m3fn).float() * s
    return torch.cat([q8.reshape(*qs.shape[:-1], 64), qs[..., 64:].to(torch.bfloat16).float()], dim=-1)


# ---- attn_d512_fp8.hip (VAE mid-block attention, d = 512): per batch and 64-key tile 66 560 bytes
#    K8    64 rows x 528 B: bytes 0..511 = e4m3(K[key][d] / 2^(bk - 127)); bytes 512..515 of row 0 = {bk, bv, 0, 0}
#    V8^T  512 rows x 64 B, logical byte / chunk swizzle as above, value e4m3(V[key][d] / 2^(bv - 127))
D512_TILE_BYTES, D512_K_BYTES, D512_KROW = 66560, 33792, 528


def decode_tiles_d512(ws_u8, b, t):
    """-> (K, V) dequantised, float32 [b][t][512], on ws_u8's device."""
    nt = t // 64
    img = ws_u8[: b * nt * D512_TILE_BYTES].view(b, nt, D512_TILE_BYTES)
    dev = img.device
    kpart = img[..., :D512_K_BYTES].reshape(b, nt, 64, D512_KROW)
    vpart = img[..., D512_K_BYTES:].reshape(b, nt, 512, 64)
    bk = kpart[..., 0, 512].to(torch.int32)
    bv = kpart[..., 0, 513].to(torch.int32)
    sk = torch.pow(2.0, (bk - 127).float())[..., None, None]
    sv = torch.pow(2.0, (bv - 127).float())[..., None, None]
    K = kpart[..., :512].contiguous().view(torch.float8_e4m3fn).float() * sk        # [b][nt][64][512]
    d = torch.arange(512, device=dev)
    pc = torch.arange(4, device=dev)
    lc = pc[None, :] ^ ((d[:, None] >> 2) & 3)
    rows = vpart.reshape(b, nt, 512, 4, 16)
    logical = torch.empty_like(rows)
    logical.scatter_(3, lc[None, None, :, :, None].expand(b, nt, 512, 4, 16), rows)
    v8 = logical.reshape(b, nt, 512, 64).contiguous().view(torch.float8_e4m3fn).float() * sv   # [..][d][logical byte]
    key = _key_of_logical().to(dev)
    Vt = torch.empty_like(v8)
    Vt[..., key] = v8
    return K.reshape(b, t, 512), Vt.transpose(-1, -2).reshape(b, t, 512)


def quantise_q_d512(q_bf16, scale_log2):
    """[..., 512] bfloat16 -> the kernel's Q operand: times scale * log2(e) in fp32, e4m3 with ONE exponent per query
    (floor(log2 of the row maximum) - 7, at least 2^-126 * 2)."""
    qs = q_bf16.float() * scale_log2
    mx = qs.abs().amax(dim=-1, keepdim=True)
    byte = ((mx.view(torch.int32) >> 23).clamp_min(8) - 7)
    s = torch.pow(2.0, (byte - 127).float())
    return (qs / s).to(torch.float8_e4m3fn).float() * s
