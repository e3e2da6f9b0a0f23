"""Correctness at the HEADLINE sizes (BASELINE.json configs[1] 2048 x 2048 untiled, configs[2] 4K --tiled + hipGraph), where the fp32
oracle cannot run in seconds: the largest kernels are checked one by one against fp64 / fp32 PyTorch references on sampled rows
and crops, and the whole path through two independent kernel sets (ping-pong / register-resident kernels vs the older 4-wave
kernels, ir_set_plain_kernels) plus size-independent properties. Reduced architectures are checked against the output of the
reference's own process() (tests/golden/process_small.npz), snapped tiles included."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from instarevive_amd import _lib as L

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _bf16_bits(t):  # fp32 / bf16 device tensor -> int16 bit pattern (device)
    return t.to(torch.bfloat16).contiguous().view(torch.int16)


def _psnr(a, b):
    mse = float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean())
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


# ------------------------------------------------------------------------------------------------ attention at T = 16384 / 65536
@pytest.mark.parametrize("heads,t,d,gain", [(16, 16384, 72, 3.0), (1, 65536, 512, 3.0)])
def test_attention_at_headline_size(ctx, heads, t, d, gain):
    """DiT self-attention of the 2048 x 2048 image (16 heads x 72, 16384 tokens: flash_attn_pp_kernel) and the VAE mid-block
    attention (1 head x 512, 65536 tokens: the d = 512 kernel) against an fp64 softmax on 320 sampled query rows (block
    boundaries, first / last rows, random ones). Logits have a standard deviation of `gain`, so a handful of the keys carry each row."""
    g = torch.Generator(device="cuda").manual_seed(t + d)
    q = (torch.randn(1, t, heads, d, generator=g, device="cuda") * gain).to(torch.bfloat16)
    k = torch.randn(1, t, heads, d, generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn(1, t, heads, d, generator=g, device="cuda").to(torch.bfloat16)
    o = torch.empty(1, t, heads, d, dtype=torch.int16, device="cuda")
    ws = torch.empty((heads * ((d + 31) // 32 * 32) if d != 512 else 2 * 512) * (t + 64) * 2 + 8192, dtype=torch.uint8, device="cuda")
    scale = d ** -0.5
    ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), L.ptr(q.view(torch.int16)), L.ptr(k.view(torch.int16)), L.ptr(v.view(torch.int16)),
                                      L.ptr(o), 1, heads, t, t, d, scale, None, L.ptr(ws), ws.numel()), "attention")
    torch.cuda.synchronize()
    rows = sorted(set([0, 1, 31, 32, 127, 128, 255, 256, 257, t // 2 - 1, t // 2, t - 257, t - 256, t - 33, t - 2, t - 1] +
                      torch.randint(0, t, (304,), generator=torch.Generator().manual_seed(5)).tolist()))
    idx = torch.tensor(rows, device="cuda")
    got = o.view(torch.bfloat16)[0, idx].float()                                # [R, heads, d]
    # The kernels' defined arithmetic: Q is multiplied by scale * log2(e) and rounded to bf16 ONCE when it is loaded (the scores then
    # come out of the MFMA in the exp2 domain). Against an fp64 softmax over exactly those operands only the bf16 rounding of P and of
    # the output remain: 2^-6 relative + 6e-3, the tolerance of test_ops_gpu.py::test_flash_attention. Against the softmax of the
    # un-prescaled operands the re-rounding of Q shows as well (2^-9 relative per element on logits of +-4 sigma = +-12: about 1 % on
    # the weight of a dominant key), so that comparison gets 2^-5 relative + 1.5e-2.
    qs = (q.float() * (scale * 1.4426950408889634)).to(torch.bfloat16)
    worst = 0.0
    for hd in range(heads):
        kd, vd = k[0, :, hd].double(), v[0, :, hd].double()
        s2 = qs[0, idx, hd].double() @ kd.t()                                   # [R, T] fp64, log2 domain
        ref = (torch.softmax(s2 * 0.6931471805599453, dim=-1) @ vd).float()
        err = (got[:, hd] - ref).abs()
        bad = err > 6e-3 + 2 ** -6 * ref.abs()
        worst = max(worst, float(err.max()))
        assert not bad.any(), f"head {hd}: {int(bad.sum())}/{bad.numel()} off, max abs err {float(err.max()):.4g}, |ref| max {float(ref.abs().max()):.3g}"
        ref0 = (torch.softmax((q[0, idx, hd].double() @ kd.t()) * scale, dim=-1) @ vd).float()
        err0 = (got[:, hd] - ref0).abs()
        assert not (err0 > 1.5e-2 + 2 ** -5 * ref0.abs()).any(), f"head {hd} vs un-prescaled operands: max abs err {float(err0.max()):.4g}"
        assert float(ref.abs().max()) > 0.3                                    # a peaked softmax: the output is not an average of everything
    print(f"attention {heads}x{d} T={t}: max abs err {worst:.4g} on {len(rows)} sampled rows")


@pytest.mark.parametrize("heads,t,gain", [(2, 1024, 3.0), (16, 16384, 3.0), (4, 4096, 0.5)])
def test_attention_fp8(ctx, heads, t, gain):
    """BASELINE.json configs[4], DiT self-attention on e4m3 MFMA operands (flash_attn_fp8_kernel): against an fp64 softmax over the
    DEQUANTISED operands the kernel's MFMAs saw (K8 / V8^T tile images read back from the workspace, Q quantised as the kernel does),
    where only the e4m3 rounding of the probabilities (per query and 32-key block, relative 2^-4 at worst) and the bf16 output
    rounding remain; and against the fp64 softmax of the bf16 operands, which prices the whole fp8 error. T = 16384 is the headline
    size; gain 0.5 is a flat softmax (thousands of keys carry each row), gain 3 a peaked one."""
    from tests.support.fp8_tiles import decode_tiles, quantise_q
    d = 72
    g = torch.Generator(device="cuda").manual_seed(t + heads)
    q = (torch.randn(1, t, heads, d, generator=g, device="cuda") * gain).to(torch.bfloat16)
    k = torch.randn(1, t, heads, d, generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn(1, t, heads, d, generator=g, device="cuda").to(torch.bfloat16)
    o = torch.empty(1, t, heads, d, dtype=torch.int16, device="cuda")
    ws = torch.zeros(heads * (t // 64) * 10240 + heads * 96 * (t + 128) * 2 + 8192, dtype=torch.uint8, device="cuda")
    scale = d ** -0.5
    ctx.check(ctx.lib.ir_op_attention_fp8(ctx.h, ctx.stream(), L.ptr(q.view(torch.int16)), L.ptr(k.view(torch.int16)), L.ptr(v.view(torch.int16)),
                                          L.ptr(o), 1, heads, t, scale, L.ptr(ws), ws.numel()), "attention_fp8")
    torch.cuda.synchronize()
    Kd, Vd = decode_tiles(ws, 1, heads, t)                                     # [1][heads][t][72] fp32
    # the tile images must be the e4m3 rounding of K / V: relative 2^-4 of the tile maximum at worst
    for name, deq, src in (("K", Kd, k), ("V", Vd, v)):
        ref = src[0].float().permute(1, 0, 2)
        err = (deq[0] - ref).abs().max() / ref.abs().max()
        assert float(err) <= 2 ** -4, f"{name} tile images: max error {float(err):.4f} of the maximum"
    assert torch.equal(Kd[0, :, :, 64:], k[0].float().permute(1, 0, 2)[:, :, 64:]), "d 64..71 of K stay bf16"
    rows = sorted(set([0, 1, 31, 32, 63, 64, 255, 256, t // 2, t - 257, t - 64, t - 1] + torch.randint(0, t, (244,), generator=torch.Generator().manual_seed(5)).tolist()))
    idx = torch.tensor(rows, device="cuda")
    got = o.view(torch.bfloat16)[0, idx].float()                               # [R][heads][d]
    qd = quantise_q(q[0, idx], scale * 1.4426950408889634)                     # [R][heads][72], log2 domain
    worst, worst_bf, num, den, num_bf = 0.0, 0.0, 0.0, 0.0, 0.0
    for hd in range(heads):
        s2 = qd[:, hd].double() @ Kd[0, hd].double().t()
        ref = (torch.softmax(s2 * 0.6931471805599453, dim=-1) @ Vd[0, hd].double()).float()
        ref_bf = (torch.softmax((q[0, idx, hd].double() @ k[0, :, hd].double().t()) * scale, dim=-1) @ v[0, :, hd].double()).float()
        err, err_bf = (got[:, hd] - ref).abs(), (got[:, hd] - ref_bf).abs()
        worst, worst_bf = max(worst, float(err.max())), max(worst_bf, float(err_bf.max()))
        num, num_bf, den = num + float((err ** 2).sum()), num_bf + float((err_bf ** 2).sum()), den + float((ref_bf ** 2).sum())
        assert float(ref_bf.abs().max()) > (0.3 if gain >= 3 else 0.02)
    r, r_bf = (num / den) ** 0.5, (num_bf / den) ** 0.5
    print(f"attention fp8 {heads}x72 T={t} gain {gain}: vs dequantised operands rel-L2 {r:.4f} max abs {worst:.4f}; vs bf16 operands rel-L2 {r_bf:.4f} max abs {worst_bf:.4f}")
    assert r <= 0.03 and r_bf <= 0.12   # measured 0.017 / 0.083 at gain 3 (peaked softmax: the e4m3 rounding of Q and K moves the logits by ~0.08 nat)


@pytest.mark.parametrize("b,t,gain", [(1, 1024, 1.0), (2, 4096, 2.0), (1, 65536, 1.0)])
def test_attention_d512_fp8(ctx, b, t, gain):
    """BASELINE.json configs[4], the VAE mid-block attention (one head, d = 512) on e4m3 MFMA operands (flash_attn_d512_fp8_kernel): against
    an fp64 softmax over the DEQUANTISED operands its MFMAs saw (K8 / V8^T tile images read back, Q quantised as the kernel does), where only
    the e4m3 rounding of the probabilities and the bf16 output rounding remain, and against the fp64 softmax of the bf16 operands, which
    prices the whole fp8 error. T = 65 536 is the headline size (2048 x 2048 image)."""
    from tests.support.fp8_tiles import decode_tiles_d512, quantise_q_d512, D512_TILE_BYTES
    d = 512
    g = torch.Generator(device="cuda").manual_seed(t + b)
    q = (torch.randn(b, t, d, generator=g, device="cuda") * gain).to(torch.bfloat16)
    k = torch.randn(b, t, d, generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn(b, t, d, generator=g, device="cuda").to(torch.bfloat16)
    o = torch.empty(b, t, d, dtype=torch.int16, device="cuda")
    ws = torch.zeros(b * (t // 64) * D512_TILE_BYTES + 4096 + (t + 64) * 512 * 2, dtype=torch.uint8, device="cuda")
    scale = d ** -0.5
    ctx.check(ctx.lib.ir_op_attention_d512_fp8(ctx.h, ctx.stream(), L.ptr(q.view(torch.int16)), L.ptr(k.view(torch.int16)), L.ptr(v.view(torch.int16)),
                                               L.ptr(o), b, t, scale, L.ptr(ws), ws.numel()), "attention_d512_fp8")
    torch.cuda.synchronize()
    tb = (b * (t // 64) * D512_TILE_BYTES + 255) // 256 * 256
    assert int(ws[tb:tb + 4].view(torch.int32)[0]) == 0, "the fixed softmax reference must hold on this input (no fallback)"
    Kd, Vd = decode_tiles_d512(ws, b, t)
    for name, deq, src in (("K", Kd, k), ("V", Vd, v)):
        err = (deq - src.float()).abs().max() / src.float().abs().max()
        assert float(err) <= 2 ** -4, f"{name} tile images: max error {float(err):.4f} of the maximum"
    rows = sorted(set([0, 1, 31, 32, 127, 128, t // 2, t - 129, t - 1] + torch.randint(0, t, (120,), generator=torch.Generator().manual_seed(5)).tolist()))
    idx = torch.tensor(rows, device="cuda")
    num = num_bf = den = 0.0
    worst = worst_bf = 0.0
    for bi in range(b):
        got = o.view(torch.bfloat16)[bi, idx].float()
        qd = quantise_q_d512(q[bi, idx], scale * 1.4426950408889634)
        s2 = qd.double() @ Kd[bi].double().t()
        ref = (torch.softmax(s2 * 0.6931471805599453, dim=-1) @ Vd[bi].double()).float()
        ref_bf = (torch.softmax((q[bi, idx].double() @ k[bi].double().t()) * scale, dim=-1) @ v[bi].double()).float()
        err, err_bf = (got - ref).abs(), (got - ref_bf).abs()
        worst, worst_bf = max(worst, float(err.max())), max(worst_bf, float(err_bf.max()))
        num, num_bf, den = num + float((err ** 2).sum()), num_bf + float((err_bf ** 2).sum()), den + float((ref_bf ** 2).sum())
    r, r_bf = (num / den) ** 0.5, (num_bf / den) ** 0.5
    print(f"attention d512 fp8 b={b} T={t} gain {gain}: vs dequantised operands rel-L2 {r:.4f} max abs {worst:.4f}; vs bf16 operands rel-L2 {r_bf:.4f} max abs {worst_bf:.4f}")
    assert r <= 0.03 and r_bf <= 0.12


# ------------------------------------------------------------------------------------------------ 3x3 convs on 2048 x 2048 x 256
@pytest.mark.parametrize("cin,cout,up", [(256, 256, 0), (256, 128, 0), (256, 256, 1)])
def test_conv_at_headline_size(ctx, cin, cout, up):
    """The decoder's full-resolution convolutions: 256 -> 256 and 256 -> 128 on a 2048 x 2048 x 256 activation (2^31 bytes: the
    largest tensor of the path, last rows at byte offsets just below 2^31 and element offsets above 2^29), and the nearest-2x
    upsampling conv 1024^2 -> 2048^2. Output crops (top, middle, bottom rows; full width) against F.conv2d in fp32 on the same
    bf16-rounded operands. Tolerance: fp32 accumulation, bf16 output rounding (2^-7 relative + 4e-3)."""
    h = w = 1024 if up else 2048
    g = torch.Generator(device="cuda").manual_seed(cin + cout + up)
    x = torch.randn(h, w, cin, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)     # NHWC
    wt = (torch.randn(cout, cin, 3, 3, generator=g, device="cuda") / (9 * cin) ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, generator=g, device="cuda")
    wp = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()                                      # [Cout][tap][Cin]
    ho, wo = (2 * h, 2 * w) if up else (h, w)
    out = torch.empty(ho, wo, cout, dtype=torch.int16, device="cuda")
    ctx.check(ctx.lib.ir_op_conv(ctx.h, ctx.stream(), L.ptr(x.view(torch.int16)), L.ptr(wp.view(torch.int16)), L.ptr(b), L.ptr(out), 1, h, w, cin, cout,
                                 cout, 9, 1, 1, up, L.ACT_NONE, 0.0, None, 0, 0), "conv")
    torch.cuda.synchronize()
    got_all = out.view(torch.bfloat16)
    for r0, r1 in ((0, 18), (ho // 2 - 9, ho // 2 + 9), (ho - 18, ho)):                                   # output row ranges
        lo, hi = max(r0 - 1, 0), min(r1 + 1, ho)                                                         # input rows needed (conv grid)
        if up:
            src = x[lo // 2:(hi + 1) // 2].float().permute(2, 0, 1)[None]
            src = F.interpolate(src, scale_factor=2.0, mode="nearest")[:, :, lo - 2 * (lo // 2): lo - 2 * (lo // 2) + (hi - lo)]
        else:
            src = x[lo:hi].float().permute(2, 0, 1)[None]
        src = F.pad(src, (1, 1, 1 if lo == 0 and r0 == 0 else 0, 1 if hi == ho and r1 == ho else 0))
        ref = F.conv2d(src, wt.float(), b)[0].permute(1, 2, 0)                                           # rows lo' .. : [rows, wo, cout]
        ref = ref[: r1 - r0]   # without a top pad the first valid output row is lo + 1 = r0; with it (r0 = 0) it is row 0
        got = got_all[r0:r1].float()
        err = (got - ref).abs()
        bad = err > 4e-3 + 2 ** -7 * ref.abs()
        assert not bad.any(), f"rows {r0}:{r1}: {int(bad.sum())}/{bad.numel()} off, max abs err {float(err.max()):.4g}"
    print(f"conv {cin}->{cout} up={up} at {ho}x{wo}: crops ok")


def test_vae_io_kernels_at_headline_size(ctx):
    """csrc/vae_io.hip at the size they run at: Encoder.conv_in (3 -> 128) and Decoder.norm_out + SiLU + conv_out (128 -> 3) on 2048 x 2048
    images - TWO images for the latter, so that the second one's 1 GiB of activations lies wholly past 2^30 elements. Row bands (top, middle,
    bottom; full width) against F.conv2d on the same bf16-rounded operands; conv_in's GroupNorm partial sums against fp64 sums of what it stored."""
    import ctypes
    H = W = 2048
    g = torch.Generator(device="cuda").manual_seed(77)
    # ---- conv_in
    x = torch.rand(1, 3, H, W, generator=g, device="cuda")
    wt = ((torch.rand(128, 3, 3, 3, generator=g, device="cuda") - 0.5) * 0.6).to(torch.bfloat16)
    b = (torch.rand(128, generator=g, device="cuda") - 0.5) * 0.2
    wp = torch.zeros(128, 9, 32, device="cuda", dtype=torch.bfloat16)
    wp[:, :, :3] = wt.permute(0, 2, 3, 1).reshape(128, 9, 3)
    out = torch.empty(1, H, W, 128, dtype=torch.int16, device="cuda")
    per = (H // 8) * (W // 64)
    part = torch.zeros(1, per, 2, 32, device="cuda")
    tiles = ctypes.c_int(0)
    ctx.check(ctx.lib.ir_op_vae_conv_in(ctx.h, ctx.stream(), L.ptr(x), L.ptr(wp.view(torch.int16)), L.ptr(b), L.ptr(out), L.ptr(part), 1, H, W, 2.0, -1.0,
                                        ctypes.byref(tiles)), "vae_conv_in")
    torch.cuda.synchronize()
    assert tiles.value == per
    xin = (x * 2.0 - 1.0).to(torch.bfloat16).float()
    got_all = out.view(torch.bfloat16)[0]
    for r0, r1 in ((0, 18), (H // 2 - 9, H // 2 + 9), (H - 18, H)):
        lo, hi = max(r0 - 1, 0), min(r1 + 1, H)
        src = F.pad(xin[:, :, lo:hi], (1, 1, 1 if r0 == 0 else 0, 1 if r1 == H else 0))
        ref = F.conv2d(src, wt.float(), b)[0].permute(1, 2, 0)[: r1 - r0]
        err = (got_all[r0:r1].float() - ref).abs()
        assert not (err > 2e-3 + 2 ** -7 * ref.abs()).any(), f"conv_in rows {r0}:{r1}: max abs err {float(err.max()):.4g}"
    gv = got_all.view(H // 8, 8, W // 64, 64, 32, 4)
    s1 = torch.zeros(H // 8, W // 64, 32, dtype=torch.float64, device="cuda")
    s2 = torch.zeros_like(s1)
    for ty in range(0, H // 8, 16):   # chunked: no fp64 copy of the gigabyte
        blk = gv[ty:ty + 16].double()
        s1[ty:ty + 16] = blk.sum(dim=(1, 3, 5))
        s2[ty:ty + 16] = (blk * blk).sum(dim=(1, 3, 5))
    want = torch.stack([s1.view(per, 32), s2.view(per, 32)], dim=1)[None]
    rel = float(((part.double() - want).abs() / (want.abs() + 1.0)).max())
    assert rel <= 2e-4, rel
    del out, gv, got_all
    # ---- norm_out + SiLU + conv_out, two images
    xa = (torch.randn(2, H, W, 128, generator=g, device="cuda") * 1.5).to(torch.bfloat16)
    sc = 0.5 + torch.rand(2, 128, generator=g, device="cuda")
    sh = torch.rand(2, 128, generator=g, device="cuda") - 0.5
    wo = ((torch.rand(3, 128, 3, 3, generator=g, device="cuda") - 0.5) * 0.1).to(torch.bfloat16)
    bo = (torch.rand(3, generator=g, device="cuda") - 0.5) * 0.2
    wpo = torch.zeros(32, 9, 128, device="cuda", dtype=torch.bfloat16)
    wpo[:3] = wo.permute(0, 2, 3, 1).reshape(3, 9, 128)
    bpo = torch.zeros(32, device="cuda")
    bpo[:3] = bo
    o4 = torch.full((2, H, W, 4), float("nan"), device="cuda")
    ctx.check(ctx.lib.ir_op_vae_norm_conv_out(ctx.h, ctx.stream(), L.ptr(xa.view(torch.int16)), L.ptr(sc), L.ptr(sh), L.ptr(wpo.view(torch.int16)), L.ptr(bpo),
                                              L.ptr(o4), 2, H, W), "vae_norm_conv_out")
    torch.cuda.synchronize()
    worst = 0.0
    for n in range(2):
        for r0, r1 in ((0, 18), (H // 2 - 9, H // 2 + 9), (H - 18, H)):
            lo, hi = max(r0 - 1, 0), min(r1 + 1, H)
            act = F.silu(xa[n, lo:hi].float() * sc[n] + sh[n]).to(torch.bfloat16).float().permute(2, 0, 1)[None]
            src = F.pad(act, (1, 1, 1 if r0 == 0 else 0, 1 if r1 == H else 0))
            ref = F.conv2d(src, wo.float(), bo)[0].permute(1, 2, 0)[: r1 - r0]
            err = (o4[n, r0:r1, :, :3] - ref).abs()
            worst = max(worst, float(err.max()))
            assert not (err > 3e-3 + 4e-3 * ref.abs()).any(), f"conv_out image {n} rows {r0}:{r1}: max abs err {float(err.max()):.4g}"
    assert float(o4[..., 3].abs().max()) == 0.0
    print(f"vae_io at 2048x2048: conv_in bands + statistics ok (rel {rel:.1e}), norm_conv_out bands ok on both images (max abs err {worst:.4f})")


# ------------------------------------------------------------------------------------------------ norms and layout kernels at 2048 x 2048
def _gn_reference_rows(xb, rows, gamma, beta, silu, groups=32):
    """fp64 GroupNorm statistics over the whole image xb [HW, C] (bf16 device tensor), applied to the sampled rows: -> fp32 [len(rows), C]."""
    hw, c = xb.shape
    cpg = c // groups
    s1 = torch.zeros(groups, dtype=torch.float64, device=xb.device)
    s2 = torch.zeros_like(s1)
    for r0 in range(0, hw, 1 << 18):   # chunked: an fp64 copy of the 2^30-element tensor would not fit beside it
        blk = xb[r0:r0 + (1 << 18)].double().view(-1, groups, cpg)
        s1 += blk.sum(dim=(0, 2))
        s2 += (blk * blk).sum(dim=(0, 2))
    cnt = float(hw) * cpg
    mean = s1 / cnt
    rstd = 1.0 / torch.sqrt(s2 / cnt - mean * mean + 1e-6)
    y = (xb[rows].double().view(-1, groups, cpg) - mean[None, :, None]) * rstd[None, :, None]
    y = y.view(-1, c) * gamma.double() + beta.double()
    if silu:
        y = y * torch.sigmoid(y)
    return y.float()


def _sample_rows(hw, seed):
    fixed = [0, 1, 2047, 2048, hw // 2 - 1, hw // 2, hw - 2049, hw - 2048, hw - 2, hw - 1]
    rnd = torch.randint(0, hw, (4096,), generator=torch.Generator().manual_seed(seed)).tolist()
    return torch.tensor(sorted(set(fixed + rnd)), device="cuda")


@pytest.mark.parametrize("n,c", [(1, 128), (2, 256)])
def test_groupnorm_at_headline_size(ctx, n, c):
    """gn_partial / gn_finalize / gn_apply on the decoder's full-resolution activations: (1, 2048^2, 128) and (2, 2048^2, 256) - the second
    is 2 x 2^31 bytes, so image 1 lives wholly past the 2^31-byte mark and every element offset of it is above 2^30. Against fp64
    statistics over the whole image, on ~4100 sampled pixel rows per image (first / last rows, chunk boundaries, random ones).
    Tolerance as in test_ops_gpu.py::test_groupnorm: bf16 output rounding, 2^-7 relative + 4e-3."""
    hw = 2048 * 2048
    g = torch.Generator(device="cuda").manual_seed(c)
    x = torch.empty(n, hw, c, dtype=torch.bfloat16, device="cuda")
    for i in range(n):       # a per-channel offset / gain so that group statistics differ from group to group and image to image
        for r0 in range(0, hw, 1 << 20):
            blk = torch.randn(1 << 20, c, generator=g, device="cuda")
            x[i, r0:r0 + (1 << 20)] = (blk * (1.0 + 0.5 * torch.sin(torch.arange(c, device="cuda") * 0.37 + i)) + 0.25 * (i + 1)).to(torch.bfloat16)
    gamma, beta = torch.randn(c, generator=g, device="cuda"), torch.randn(c, generator=g, device="cuda")
    y = torch.empty(n, hw, c, dtype=torch.int16, device="cuda")
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.ir_op_groupnorm(ctx.h, ctx.stream(), L.ptr(x.view(torch.int16)), L.ptr(y), L.ptr(gamma), L.ptr(beta), n, hw, c, 32, 1e-6, 1,
                                      L.ptr(ws), ws.numel()), "groupnorm")
    torch.cuda.synchronize()
    worst = 0.0
    for i in range(n):
        rows = _sample_rows(hw, 7 + i)
        ref = _gn_reference_rows(x[i], rows, gamma, beta, True)
        got = y.view(torch.bfloat16)[i, rows].float()
        err = (got - ref).abs()
        bad = err > 4e-3 + 2 ** -7 * ref.abs()
        worst = max(worst, float(err.max()))
        assert not bad.any(), f"image {i}: {int(bad.sum())}/{bad.numel()} off, max abs err {float(err.max()):.4g}"
    print(f"groupnorm ({n}, 2048^2, {c}): max abs err {worst:.4g}")


def test_conv_groupnorm_fused_statistics_at_headline_size(ctx):
    """The 128 -> 128 ResnetBlock conv of the 2048 x 2048 level with its GroupNorm statistics produced by the conv epilogue
    (conv_halo_s1_kernel: 128 x 64 = 8192 patch tiles -> gn_reduce_groups -> gn_finalize_groups -> gn_apply), residual included.
    The normalised output is checked against fp64 statistics of the kernel's OWN stored conv output (what GroupNorm is defined on),
    the conv output itself on row crops against F.conv2d."""
    import ctypes
    h = w = 2048
    cin = cout = 128
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.randn(h, w, cin, generator=g, device="cuda").to(torch.bfloat16)
    res = torch.randn(h, w, cout, generator=g, device="cuda").to(torch.bfloat16)
    wt = (torch.randn(cout, cin, 3, 3, generator=g, device="cuda") / (9 * cin) ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, generator=g, device="cuda") * 0.1
    gamma, beta = torch.randn(cout, generator=g, device="cuda"), torch.randn(cout, generator=g, device="cuda")
    wp = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()
    conv_out = torch.empty(h, w, cout, dtype=torch.int16, device="cuda")
    y = torch.empty_like(conv_out)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    fused = ctypes.c_int(-1)
    ctx.check(ctx.lib.ir_op_conv_groupnorm(ctx.h, ctx.stream(), L.ptr(x.view(torch.int16)), L.ptr(wp.view(torch.int16)), L.ptr(b), L.ptr(conv_out), L.ptr(y),
                                           L.ptr(gamma), L.ptr(beta), 1, h, w, cin, cout, 1, 0, L.ptr(res.view(torch.int16)), 1, L.ptr(ws), ws.numel(),
                                           ctypes.byref(fused)), "conv_groupnorm")
    torch.cuda.synchronize()
    assert fused.value == (h // 16) * (w // 32), fused.value          # the one-wave-per-SIMD kernel ran and wrote one partial per patch
    co = conv_out.view(torch.bfloat16)
    for r0, r1 in ((0, 17), (h // 2 - 8, h // 2 + 9), (h - 17, h)):    # conv crops, as in test_conv_at_headline_size
        lo, hi = max(r0 - 1, 0), min(r1 + 1, h)
        src = F.pad(x[lo:hi].float().permute(2, 0, 1)[None], (1, 1, 1 if r0 == 0 else 0, 1 if r1 == h else 0))
        ref = F.conv2d(src, wt.float(), b)[0].permute(1, 2, 0)[: r1 - r0] + res[r0:r1].float()
        err = (co[r0:r1].float() - ref).abs()
        assert not (err > 4e-3 + 2 ** -7 * ref.abs()).any(), f"conv rows {r0}:{r1}: max abs err {float(err.max()):.4g}"
    rows = _sample_rows(h * w, 3)
    ref = _gn_reference_rows(co.view(h * w, cout), rows, gamma, beta, True)
    got = y.view(torch.bfloat16).view(h * w, cout)[rows].float()
    err = (got - ref).abs()
    assert not (err > 4e-3 + 2 ** -7 * ref.abs()).any(), f"groupnorm from fused statistics: max abs err {float(err.max()):.4g}"
    print(f"conv 128->128 + fused GroupNorm statistics at 2048^2: {fused.value} partial tiles, max abs err {float(err.max()):.4g}")


@pytest.mark.parametrize("rows,c,ld", [(65536, 180, 192), (65536, 192, 192), (16384, 1152, 1152)])
def test_layernorm_at_headline_size(ctx, rows, c, ld):
    """LayerNorm over the token streams of the 2048 x 2048 image: SwinIR's 65 536 tokens x 180 (of 192 stored) channels
    (layernorm_r16_kernel), the same with all 192 channels live, and the DiT's 16 384 x 1152 (layernorm_v4_kernel). Every row against
    F.layer_norm in fp32; 2^-7 relative + 2e-3 (bf16 output)."""
    g = torch.Generator(device="cuda").manual_seed(rows + c)
    x = torch.randn(rows, ld, generator=g, device="cuda") * 3 + torch.linspace(-2, 2, rows, device="cuda")[:, None]
    a, b = torch.randn(c, generator=g, device="cuda"), torch.randn(c, generator=g, device="cuda")
    ref = torch.zeros(rows, ld, device="cuda")
    ref[:, :c] = F.layer_norm(x[:, :c], (c,), None, None, 1e-6) * a + b
    y = torch.full((rows, ld), 0x7fff, dtype=torch.int16, device="cuda")
    ctx.check(ctx.lib.ir_op_layernorm(ctx.h, ctx.stream(), L.ptr(x), L.ptr(y), L.ptr(a), L.ptr(b), rows, c, ld, ld, 1e-6), "layernorm")
    torch.cuda.synchronize()
    err = (y.view(torch.bfloat16).float() - ref).abs()
    assert not (err > 2e-3 + 2 ** -7 * ref.abs()).any(), f"layernorm {rows}x{c}: max abs err {float(err.max()):.4g}"


def test_layout_kernels_at_headline_size(ctx):
    """The image <-> tensor kernels at 3 x 2048 x 2048: uint8 HWC -> fp32 NCHW (/255), fp32 NCHW -> bf16 NHWC padded to 32 channels
    with the 2x - 1 map of the VAE input, fp32 NHWC[4] -> fp32 NCHW with the /2 + 0.5 map of the decoder output, fp32 NCHW -> uint8
    HWC (clamp, x255, truncation): each against the same expression in PyTorch, exactly (bf16 rounding where the kernel rounds)."""
    h = w = 2048
    hw = h * w
    g = torch.Generator(device="cuda").manual_seed(5)
    img = torch.randint(0, 256, (1, h, w, 3), generator=g, device="cuda", dtype=torch.uint8)
    f = torch.empty(1, 3, h, w, device="cuda")
    ctx.check(ctx.lib.ir_u8_to_nchw(ctx.h, ctx.stream(), L.ptr(img), L.ptr(f), 1, h, w), "u8_to_nchw")
    assert torch.equal(f, (img.double() / 255.0).float().permute(0, 3, 1, 2))
    nhwc = torch.empty(hw, 32, dtype=torch.int16, device="cuda")
    ctx.check(ctx.lib.ir_op_nchw_to_nhwc(ctx.h, ctx.stream(), L.ptr(f), L.ptr(nhwc), 1, 3, hw, 32, 2.0, -1.0), "nchw_to_nhwc")
    want = torch.zeros(hw, 32, device="cuda")
    want[:, :3] = (f[0] * 2.0 - 1.0).view(3, hw).t()
    assert torch.equal(nhwc.view(torch.bfloat16), want.to(torch.bfloat16))
    o4 = torch.randn(hw, 4, generator=g, device="cuda") * 1.5
    nchw = torch.empty(1, 3, h, w, device="cuda")
    ctx.check(ctx.lib.ir_op_nhwc_to_nchw(ctx.h, ctx.stream(), L.ptr(o4), 4, L.ptr(nchw), 1, 3, hw, 0.5, 0.5, 0), "nhwc_to_nchw")
    assert torch.allclose(nchw.view(3, hw), (o4[:, :3] * 0.5 + 0.5).t(), rtol=0, atol=1e-7)   # one fma against mul + add
    u8 = torch.empty(1, h, w, 3, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.ir_nchw_to_u8(ctx.h, ctx.stream(), L.ptr(nchw), L.ptr(u8), 1, h, w), "nchw_to_u8")
    assert torch.equal(u8, (nchw.clamp(0, 1) * 255.0).permute(0, 2, 3, 1).to(torch.uint8))


# ------------------------------------------------------------------------------------------------ reduced architecture vs the reference's process()
@pytest.mark.parametrize("case", ["untiled", "nopre", "tiled_wavelet", "tiled_adain", "tiled_none"])
def test_process_vs_reference_process_fixture(case):
    """The HIP path against the uint8 OUTPUT OF THE REFERENCE's own process() (not the oracle): untiled, --disable_preprocess_model
    and --tiled with snapped last tiles (latent 24 x 32, tile 8, stride 5: 5 x 6 tiles) under the three colour-fix modes; fused
    ir_pipeline and the stage-by-stage form. >= 45 dB on the uint8 result, >= 50 dB on the stage-1 image."""
    from instarevive_amd.models import AutoencoderKL, SwinIR, Transformer2DModel
    from instarevive_amd.pipeline import process
    from tests.test_oracle_golden import PROCESS_CASES, _process_small_models
    fx = np.load(os.path.join(G, "process_small.npz"))
    sws, svae, _, dsd = _process_small_models()
    swin = SwinIR(img_size=64, patch_size=1, in_chans=3, embed_dim=60, depths=[2, 2], num_heads=[6, 6], window_size=8, mlp_ratio=2, sf=8, img_range=1.0,
                  upsampler="nearest+conv", resi_connection="1conv", unshuffle=True, unshuffle_scale=8)
    swin.load_state_dict(sws, strict=False)
    vae = AutoencoderKL(block_out_channels=(32, 64, 128, 128))
    vae.load_state_dict(svae)
    dit = Transformer2DModel(num_attention_heads=4, attention_head_dim=72, num_layers=2, sample_size=16, caption_channels=64, cross_attention_dim=288)
    dit.load_state_dict(dsd)
    for m in (swin, vae, dit):
        m.to("cuda")
    img_key, kw = PROCESS_CASES[case]
    want, want1 = fx[case + "_pred"], fx[case + "_stage1"]
    imgs = list(fx[img_key])[: want.shape[0]]
    y = torch.from_numpy(fx["y"]).cuda()
    for fused in (True, False):
        got, got1 = process(dit, imgs, 1, kw["color_fix_type"], kw.get("disable_preprocess_model", False), kw["tiled"], kw.get("tile_size", 512),
                            kw.get("tile_stride", 448), preprocess_model=swin, vae=vae, y=y, y_mask=None, fused=fused)
        p, p1 = _psnr(np.stack(got), want), _psnr(np.stack(got1), want1)
        print(f"{case} fused={fused}: PSNR vs the reference's process() {p:.2f} dB (stage-1 {p1:.2f} dB)")
        assert p >= 45.0 and p1 >= 50.0


# ------------------------------------------------------------------------------------------------ whole path at 2048 x 2048 and 4K tiled
def _process_full(full_models, img, tiled, graph=False, fix="wavelet"):
    from instarevive_amd.pipeline import process
    swin, vae, dit, sds, y, mask = full_models
    return process(dit, [img], 1, fix, False, tiled, 512, 448, preprocess_model=swin, vae=vae, y=full_models.y_cuda, y_mask=full_models.mask_cuda,
                   graph=graph)


def test_headline_2048_untiled_fast_vs_plain_kernels(full_models):
    """BASELINE configs[1] at full architecture: the 2048 x 2048 untiled pass through the default (fast) kernels and through the older
    4-wave kernel set — two independent implementations of every large contraction (ping-pong conv / GEMM / attention and the
    register-resident d = 512 attention vs their predecessors) — must agree to >= 45 dB on the uint8 result; plus determinism."""
    import bench
    swin, vae, dit, sds, y, mask = full_models
    ctx = dit.ctx
    img = bench.synthetic_lq(1, 2048, 2048, 31)[0].numpy()
    fast, st1 = _process_full(full_models, img, False)
    again, _ = _process_full(full_models, img, False)
    assert np.array_equal(fast[0], again[0]), "the path must be deterministic run to run"
    ctx.check(ctx.lib.ir_set_plain_kernels(ctx.h, 1), "ir_set_plain_kernels")
    try:
        plain, st1p = _process_full(full_models, img, False)
    finally:
        ctx.check(ctx.lib.ir_set_plain_kernels(ctx.h, 0), "ir_set_plain_kernels")
    p, p1 = _psnr(fast[0], plain[0]), _psnr(st1[0], st1p[0])
    print(f"2048x2048 untiled: fast vs plain kernels {p:.2f} dB (stage-1 {p1:.2f} dB), output std {fast[0].std():.1f}")
    assert p >= 45.0 and p1 >= 50.0 and fast[0].std() > 1.0 and st1[0].std() > 1.0


@pytest.mark.parametrize("size", [1024, 2048])
def test_headline_vs_oracle_crops(full_models, size):
    """BASELINE configs[1] pinned AT ITS OWN SIZE against the fp32 oracle: tests/golden/headline_crops.npz holds the
    x^_0 latent and 128 x 128 uint8 crops of ONE oracle pass per size (1 / 8 minutes of host time, made by the committed
    tests/golden/make_headline_crops.py from bench.py's seeded weights and synthetic input). The HIP path on the same input must match the
    crops to >= 45 dB (bf16) and the latent to <= 1.2 % relative L2; fp8 (cfg-5, reported separately): the default operand set >= 46.3 dB on the
    crops (the 0.1 dB tolerance at a 30 dB reference), the set of every part >= 41 dB (reported as out of tolerance). At 2048 this
    is the first whole-path oracle comparison that runs through gemm_pp_kernel, the 16384-token DiT attention and the 65536-token VAE
    attention in their real chain."""
    from instarevive_amd.models import DDPMScheduler
    from instarevive_amd.pipeline import process
    from tests.golden.make_headline_crops import CROP, inputs_for
    swin, vae, dit, sds, y, mask = full_models
    z = np.load(os.path.join(G, "headline_crops.npz"))
    img = inputs_for(size)
    assert img.shape == (size, size, 3)
    kw = dict(preprocess_model=swin, vae=vae, y=full_models.y_cuda, y_mask=full_models.mask_cuda)
    bf, st1 = process(dit, [img], 1, "wavelet", False, False, 512, 448, **kw)
    pos = z[f"pos_{size}"]
    take = lambda a, pp: np.stack([a[yy:yy + CROP, xx:xx + CROP] for yy, xx in pp])
    want, want1 = z[f"crops_{size}"], z[f"stage1_{size}"]
    got, got1 = take(bf[0], pos), take(st1[0], pos[:len(want1)])
    per = [_psnr(g, w) for g, w in zip(got, want)]
    p, p1 = _psnr(got, want), _psnr(got1, want1)
    print(f"{size}x{size} bf16 vs fp32 oracle: crops {p:.2f} dB (worst crop {min(per):.2f}), stage-1 crops {p1:.2f} dB; "
          f"byte sum {int(bf[0].astype(np.int64).sum())} vs oracle {int(z[f'sum_{size}'])}")
    assert p >= 45.0 and min(per) >= 43.0 and p1 >= 50.0
    assert abs(int(bf[0].astype(np.int64).sum()) - int(z[f"sum_{size}"])) <= 0.002 * int(z[f"sum_{size}"])   # no global shift in brightness
    # the x^_0 latent through the staged calls (SwinIR -> encode -> one DiT step), against the oracle's
    x = torch.from_numpy(img).cuda().permute(2, 0, 1)[None].float() / 255.0
    ctl = swin(x)
    lat = vae.encode(ctl * 2 - 1).latent_dist.mode() * float(vae.config.scaling_factor)
    x0 = dit.step(lat, 400.0, float(DDPMScheduler().alphas_cumprod[400]), full_models.y_cuda, full_models.mask_cuda)
    ref0 = torch.from_numpy(z[f"x0_{size}"].astype(np.float32)).cuda()[None]
    rel = float((x0 - ref0).norm() / ref0.norm())
    print(f"{size}x{size} x0 latent vs oracle: relative L2 {rel * 100:.3f} % (the fixture stores fp16: 0.03 %)")
    assert rel <= 0.012
    # cfg-5 (fp8), reported separately. The DEFAULT operand set (IR_FP8_MASK_DEFAULT) is chosen by north_star's tolerance: an error of >= 46.3 dB
    # against the oracle moves PSNR(., GT) by <= 0.1 dB up to a 30 dB reference (tests/support/psnr_guard.py). The set of EVERY
    # part (IR_FP8_MASK_ALL) is faster and out of that tolerance: reported, gated only as "still the same picture".
    from instarevive_amd import _lib as L
    ctx = dit.ctx
    vae.enable_fp8(True)
    try:
        f8, _ = process(dit, [img], 1, "wavelet", False, False, 512, 448, fp8=True, **kw)
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, L.FP8_MASK_ALL), "ir_set_fp8_mask")
        f8_all, _ = process(dit, [img], 1, "wavelet", False, False, 512, 448, fp8=True, **kw)
    finally:
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, L.FP8_MASK_DEFAULT), "ir_set_fp8_mask")
        vae.enable_fp8(False)
    p8, p8_all = _psnr(take(f8[0], pos), want), _psnr(take(f8_all[0], pos), want)
    print(f"{size}x{size} fp8 vs fp32 oracle: default operand set {p8:.2f} dB (within 0.1 dB up to a {p8 - 16.33:.1f} dB reference), every part {p8_all:.2f} dB "
          f"(up to {p8_all - 16.33:.1f} dB: out of tolerance, opt-in)")
    assert not np.array_equal(f8[0], bf[0]) and not np.array_equal(f8_all[0], f8[0])
    assert p8 >= 46.3, "the default fp8 operand set must stay within 0.1 dB of the reference's PSNR up to a 30 dB reference"
    assert p8_all >= 41.0


def test_batch8_at_2048_matches_batch1(full_models):
    """BASELINE configs[3]'s per-GPU workload (batch 8 of 512 x 512 LQ, sr_scale 4 -> eight 2048 x 2048 images through ONE ir_pipeline call:
    17 GB per 256-channel activation, 8.6 G elements - past 2^32 - per tensor): image i of the batch must be bit-identical to its batch-1
    result, which any 32-bit index overflow in any kernel of the chain would break. Batches of 2, 4 and 8 in turn, so that a failure
    names the size where an index first wraps."""
    import bench
    imgs = [bench.synthetic_lq(1, 2048, 2048, 70 + i)[0].numpy() for i in range(8)]
    single = {i: _process_full(full_models, imgs[i], False) for i in (0, 1, 3, 7)}
    from instarevive_amd.pipeline import process
    swin, vae, dit, sds, y, mask = full_models
    kw = dict(preprocess_model=swin, vae=vae, y=full_models.y_cuda, y_mask=full_models.mask_cuda)
    for nb in (2, 4, 8):
        preds, st1 = process(dit, imgs[:nb], 1, "wavelet", False, False, 512, 448, **kw)
        assert len(preds) == nb
        for i in (j for j in single if j < nb):
            assert np.array_equal(st1[i], single[i][1][0]), f"batch {nb}: stage-1 image {i} differs from its batch-1 result"
            assert np.array_equal(preds[i], single[i][0][0]), f"batch {nb}: image {i} differs from its batch-1 result"
    assert not np.array_equal(preds[7], preds[0]) and preds[7].std() > 1.0


def test_headline_2048_fp8_vs_bf16(full_models):
    """BASELINE configs[4] at the workload size of its bench line (2048 x 2048 network input): the fp8 forms of the path against the bf16 path on
    the same image. The default operand set (chosen by the 0.1 dB tolerance) measured 49.5 dB against bf16, gate 47; every part 42.5 dB, gate 38
    (e4m3 carries 3 mantissa bits). Switching fp8 off restores the bf16 result bit for bit."""
    import bench
    from instarevive_amd import _lib as L
    from instarevive_amd.pipeline import process
    swin, vae, dit, sds, y, mask = full_models
    ctx = dit.ctx
    img = bench.synthetic_lq(1, 2048, 2048, 33)[0].numpy()
    kw = dict(preprocess_model=swin, vae=vae, y=full_models.y_cuda, y_mask=full_models.mask_cuda)
    bf, _ = process(dit, [img], 1, "wavelet", False, False, 512, 448, **kw)
    vae.enable_fp8(True)
    try:
        f8, _ = process(dit, [img], 1, "wavelet", False, False, 512, 448, fp8=True, **kw)
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, L.FP8_MASK_ALL), "ir_set_fp8_mask")
        f8_all, _ = process(dit, [img], 1, "wavelet", False, False, 512, 448, fp8=True, **kw)
    finally:
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, L.FP8_MASK_DEFAULT), "ir_set_fp8_mask")
        vae.enable_fp8(False)
    again, _ = process(dit, [img], 1, "wavelet", False, False, 512, 448, **kw)
    assert np.array_equal(again[0], bf[0]), "switching fp8 off must restore the bf16 result bit for bit"
    p, p_all = _psnr(f8[0], bf[0]), _psnr(f8_all[0], bf[0])
    print(f"2048x2048 fp8 vs bf16: default operand set {p:.2f} dB, every part {p_all:.2f} dB")
    assert not np.array_equal(f8[0], bf[0]) and p >= 47.0 and 38.0 <= p_all < p


def test_4k_tiled_hipgraph(full_models):
    """BASELINE configs[2]: a padded 4K frame (2176 x 3840, 45 tiles of 512 px, stride 448: snapped last row of tiles) --tiled with
    the wavelet fix. The hipGraph replay must be bit-identical to the plain launch sequence (recording call and replay on new
    pixels), and the fast and the plain kernel sets must agree to >= 45 dB."""
    import bench
    swin, vae, dit, sds, y, mask = full_models
    ctx = dit.ctx
    a, b = (bench.synthetic_lq(1, 2176, 3840, s)[0].numpy() for s in (41, 42))
    assert ctx.lib.ir_tiled_count(2176, 3840, 512, 448) == 45
    want_a, _ = _process_full(full_models, a, True)
    rec_a, _ = _process_full(full_models, a, True, graph=True)       # records
    assert np.array_equal(rec_a[0], want_a[0])
    want_b, _ = _process_full(full_models, b, True)
    rep_b, _ = _process_full(full_models, b, True, graph=True)       # replays on new pixels
    assert np.array_equal(rep_b[0], want_b[0]) and not np.array_equal(want_a[0], want_b[0])
    ctx.check(ctx.lib.ir_set_plain_kernels(ctx.h, 1), "ir_set_plain_kernels")
    try:
        plain_a, _ = _process_full(full_models, a, True)
    finally:
        ctx.check(ctx.lib.ir_set_plain_kernels(ctx.h, 0), "ir_set_plain_kernels")
    p = _psnr(want_a[0], plain_a[0])
    print(f"4K tiled: fast vs plain kernels {p:.2f} dB, output std {want_a[0].std():.1f}")
    assert p >= 45.0 and want_a[0].std() > 1.0


def test_tile_engine_equals_ir_pipeline_tiled(full_models):
    """The five ir_tiled_* phases driven from the host (what a tile-sharded multi-GPU run executes, here as one rank and as two
    simulated ranks whose tile sets are exchanged in memory) reproduce ir_pipeline(IR_FLAG_TILED) bit for bit at 1024 x 1536."""
    import bench
    from instarevive_amd.parallel import sharded_tiled_process
    from instarevive_amd.pipeline import HipTileEngine
    swin, vae, dit, sds, y, mask = full_models
    img = bench.synthetic_lq(1, 1024, 1536, 51)[0].numpy()
    want, want1 = _process_full(full_models, img, True)
    eng = HipTileEngine(dit, vae, swin, full_models.y_cuda, full_models.mask_cuda, "wavelet", False, 512, 448)
    got, got1 = sharded_tiled_process(eng, [img], rank=0, world=1)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got1[0], want1[0])
    # two "ranks" on this one GPU: each runs its own tiles, the tile stacks are merged in loop order as the collectives would
    control, init = eng.encode([img])
    n_tiles = eng.count(1024, 1536)
    parts = [eng.dit_tiles(init, r, 2).clone() for r in range(2)]
    x0_all = torch.empty((n_tiles,) + tuple(parts[0].shape[1:]), device="cuda")
    for r in range(2):
        x0_all[r::2] = parts[r]
    nb = eng.blend_latent(x0_all)
    parts = [eng.decode_tiles(nb, control, r, 2).clone() for r in range(2)]
    px_all = torch.empty((n_tiles,) + tuple(parts[0].shape[1:]), device="cuda")
    for r in range(2):
        px_all[r::2] = parts[r]
    two = eng.blend_pixels(px_all)
    # the encoder's mid-block attention split by query rows (parallel.sharded_encode): three simulated ranks, rows merged as the all_gather would
    from instarevive_amd.parallel import row_shards
    assert eng.can_shard_encode([img])
    T = (1024 // 8) * (1536 // 8)
    shards = row_shards(T, 3)
    assert shards[0][0] == 0 and shards[-1][1] == T and all(a % 128 == 0 and b % 128 == 0 for a, b in shards)
    merged = None
    for a, b in shards:
        c_r, o_r, res_r = eng.encode_part0([img], a, b)
        merged = o_r.clone() if merged is None else merged
        merged[a:b] = o_r[a:b]
        assert torch.equal(c_r, control)
    init_sharded = eng.encode_part1(control, merged, res_r).clone()
    assert torch.equal(init_sharded, init), "row-sharded mid-block attention must reproduce the unsharded encoder bit for bit"
    d = np.abs(two[0].astype(int) - want[0].astype(int))
    print(f"two simulated ranks vs one call: max |diff| {d.max()} grey levels, {100 * (d != 0).mean():.4f} % of the values")
    assert d.max() <= 1   # bit-identical re-assembly; the per-tile kernels may differ in the last bit when the row count selects another GEMM tiling


def test_stress_weights_vs_oracle(full_models):
    """Parity where released weights live (reference shapes ldm/modules/diffusionmodules/model.py:181-205, PixArt_blocks.py:43-58,
    123-158): the full architectures at 512 x 512 with tests/support/stress_weights.py on top of the seeded weights - 1 % of the DiT residual-stream
    channels, MLP hidden units and VAE ResnetBlock channels x30, and the q / k projections of every self-attention scaled by the per-attention
    logit gains that tests/golden/make_stress_fixture.py calibrated on the oracle so that EVERY attention's rows are peaky (median max - min logit
    34 per row, median top-1 softmax mass 0.2 - 0.7 in the DiT) - against the fp32 oracle's result stored in tests/golden/stress_512.npz.
    Gate: bf16 >= 45 dB on the uint8 result, the one-step latent within 1.5 % relative L2; the fp8 default operand set is reported and gated 2 dB
    below; the number of attention launches that raised the fixed-reference overflow flag (and took the rescaling fallback) is reported."""
    import bench
    from instarevive_amd import _lib as L
    from instarevive_amd.models import DDPMScheduler
    from instarevive_amd.pipeline import process
    from tests.support.stress_weights import stress_state_dicts
    swin, vae, dit, sds, y, mask = full_models
    ctx = dit.ctx
    z = np.load(os.path.join(G, "stress_512.npz"))
    gains = {"dit": [float(v) for v in z["logit_gain_dit"]], "vae_encoder": float(z["logit_gain_vae"][0]), "vae_decoder": float(z["logit_gain_vae"][1])}
    st = stress_state_dicts(sds, float(z["frac"]), float(z["gain"]), gains)
    img = bench.synthetic_lq(1, 512, 512, int(z["lq_seed"]))[0].numpy()
    kw = dict(preprocess_model=swin, vae=vae, y=full_models.y_cuda, y_mask=full_models.mask_cuda)
    seeded, _ = process(dit, [img], 1, "wavelet", False, False, 512, 448, **kw)   # the session's weights, before anything is swapped
    try:
        vae.load_state_dict(st["vae"])
        dit.load_state_dict(st["dit"])
        dit.invalidate_prompt()
        ctx.check(ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), 1), "ir_attn_fallback_count")
        bf, st1 = process(dit, [img], 1, "wavelet", False, False, 512, 448, **kw)
        fallbacks = ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), 0)
        ctx.check(ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), -1), "ir_attn_fallback_count")
        x = torch.from_numpy(img).cuda().permute(2, 0, 1)[None].float() / 255.0
        lat = vae.encode(swin(x) * 2 - 1).latent_dist.mode() * float(vae.config.scaling_factor)
        x0 = dit.step(lat, 400.0, float(DDPMScheduler().alphas_cumprod[400]), full_models.y_cuda, full_models.mask_cuda)
        vae.enable_fp8(True)
        f8 = {}
        try:   # cfg-5 under the same stress, part by part: REPORTED (north_star's tolerance is stated on the bf16 path; see the assertion below)
            for name, m8 in (("default set", L.FP8_MASK_DEFAULT), ("qualified set", L.FP8_MASK_QUALIFIED), ("DiT self-attention only", 1), ("VAE mid-block attention only", 0b110), ("decoder level-0 / level-2 convs only", 0x5000)):
                ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, m8), "ir_set_fp8_mask")
                f8[name] = process(dit, [img], 1, "wavelet", False, False, 512, 448, fp8=True, **kw)[0][0]
        finally:
            ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, L.FP8_MASK_DEFAULT), "ir_set_fp8_mask")
            vae.enable_fp8(False)
    finally:   # the session's models go back to the seeded weights whatever happened
        vae.load_state_dict(sds["vae"])
        dit.load_state_dict(sds["dit"])
        dit.invalidate_prompt()
    p, p1 = _psnr(bf[0], z["pred"]), _psnr(st1[0], z["stage1"])
    p8 = {k: _psnr(v, z["pred"]) for k, v in f8.items()}
    ref0 = torch.from_numpy(z["x0"].astype(np.float32)).cuda()[None]
    rel = float((x0 - ref0).norm() / ref0.norm())
    print(f"stress weights (1 % channels x{float(z['gain']):.0f}, every attention's median logit spread {float(z['spread_median'].min()):.0f}-{float(z['spread_median'].max()):.0f}) "
          f"at 512 x 512 vs the fp32 oracle: bf16 {p:.2f} dB (stage-1 {p1:.2f} dB), x0 latent relative L2 {rel * 100:.3f} %; "
          f"{fallbacks} of 30 attention launches raised the overflow flag and took the rescaling fallback; fp8: " + ", ".join(f"{k} {v:.2f} dB" for k, v in p8.items()))
    assert fallbacks == 0, "the fixture's logit spread (median 34) stays below the fixed references' head room: no launch takes the rescaling fallback (profiles/r05_stress_parity.txt: 0 of 30)"
    assert p >= 45.0 and p1 >= 50.0 and rel <= 0.015
    # fp8 under such weights (REPORTED; profiles/r05_stress_parity.txt): an e4m3 q . k carries 3.7 % of the logit's size as error -
    # at a logit spread of 34 that is a factor e^0.6 on a softmax weight - and a row dominated by one or two keys no longer averages the P . V rounding
    # away: the DiT attention part alone falls to 37 dB (the VAE's single 512-wide head to 44 dB), and the e4m3 convs cost 3 dB instead of 1 once 1 % of
    # the channels carry 30x the scale. cfg-5's tolerance claim therefore holds for the flat-softmax seeded weights it was chosen on, not for these.
    # Asserted only: nothing breaks (every operand set still produces the picture), the conv parts stay within 4 dB of bf16.
    assert p8["decoder level-0 / level-2 convs only"] >= p - 4.0 and min(p8.values()) >= 33.0
    # the context's DEFAULT set (ABI v3: without the DiT self-attention) is gated here too - it stays within 5 dB of the bf16 path on these
    # weights (measured 4.4: its VAE attention parts cost 2.2 dB, its two decoder conv levels 3.1), where round 5's constant set (now
    # IR_FP8_MASK_QUALIFIED) loses 9.8. Neither holds north_star's tolerance here; the calibrated choice (test_fp8_auto_...) does
    assert p8["default set"] >= p - 5.0 and p8["qualified set"] < p8["default set"] - 3.0
    # and the session's weights are back: the seeded-weight result is EXACTLY what it was before the swap
    again, _ = process(dit, [img], 1, "wavelet", False, False, 512, 448, **kw)
    assert np.array_equal(again[0], seeded[0]) and not np.array_equal(again[0], bf[0])


@pytest.mark.parametrize("case", ["tiled1024", "2048"])
def test_stress_weights_at_headline_size_and_tiled(full_models, case):
    """the stress weights (1 % channels x30, peaky attention: tests/support/stress_weights.py with stress_512.npz's gains) where the
    headline kernels run - 2048 x 2048 untiled (16384 DiT tokens through flash_attn_pp2_kernel / gemm_pp_kernel, 65536 VAE tokens) and 1024 x 1024
    --tiled (9 tiles + wavelet fix) - against the fp32 oracle's crops of tests/golden/stress_headline.npz (make_stress_headline.py; the oracle saw a
    median DiT logit spread of 30-42 and top-1 softmass 0.11 at 16384 keys). Gates: bf16 >= 44.5 dB on the crops, the one-step latent within 1.5 %,
    no attention launch on the rescaling fallback (the sweep of tools/spread_sweep.py puts the first one at a median spread of 56). The figure is a
    sample from a +- 0.7 dB band that moves with the summation ORDER of single convs (profiles/r06_stress_sensitivity.txt: 44.30 ... 45.71 dB at 2048
    for four combinations of two SwinIR-tail kernels); the shipped kernels read 45.71 / 46.25 dB, so the gate keeps more than that band below them."""
    from instarevive_amd.models import DDPMScheduler
    from instarevive_amd.pipeline import process
    from tests.golden.make_headline_crops import CROP, inputs_for
    from tests.support.stress_weights import stress_state_dicts
    swin, vae, dit, sds, y, mask = full_models
    ctx = dit.ctx
    z5, z = np.load(os.path.join(G, "stress_512.npz")), np.load(os.path.join(G, "stress_headline.npz"))
    gains = {"dit": [float(v) for v in z5["logit_gain_dit"]], "vae_encoder": float(z5["logit_gain_vae"][0]), "vae_decoder": float(z5["logit_gain_vae"][1])}
    st = stress_state_dicts(sds, float(z5["frac"]), float(z5["gain"]), gains)
    size, tiled = (2048, False) if case == "2048" else (1024, True)
    img = inputs_for(size)
    kw = dict(preprocess_model=swin, vae=vae, y=full_models.y_cuda, y_mask=full_models.mask_cuda)
    try:
        vae.load_state_dict(st["vae"])
        dit.load_state_dict(st["dit"])
        dit.invalidate_prompt()
        ctx.check(ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), 1), "ir_attn_fallback_count")
        got, _ = process(dit, [img], 1, "wavelet", False, tiled, 512, 448, **kw)
        fallbacks = ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), 0)
        ctx.check(ctx.lib.ir_attn_fallback_count(ctx.h, ctx.stream(), -1), "ir_attn_fallback_count")
        rel = None
        if not tiled:
            x = torch.from_numpy(img).cuda().permute(2, 0, 1)[None].float() / 255.0
            lat = vae.encode(swin(x) * 2 - 1).latent_dist.mode() * float(vae.config.scaling_factor)
            x0 = dit.step(lat, 400.0, float(DDPMScheduler().alphas_cumprod[400]), full_models.y_cuda, full_models.mask_cuda)
            ref0 = torch.from_numpy(z[f"x0_{case}"].astype(np.float32)).cuda()[None]
            rel = float((x0 - ref0).norm() / ref0.norm())
    finally:
        vae.load_state_dict(sds["vae"])
        dit.load_state_dict(sds["dit"])
        dit.invalidate_prompt()
    pos, want = z[f"pos_{case}"], z[f"crops_{case}"]
    crops = np.stack([got[0][yy:yy + CROP, xx:xx + CROP] for yy, xx in pos])
    per = [_psnr(g, w) for g, w in zip(crops, want)]
    p = _psnr(crops, want)
    print(f"stress weights, {case}: bf16 {p:.2f} dB on the oracle's crops (worst crop {min(per):.2f}), x0 relative L2 {'-' if rel is None else f'{rel * 100:.3f} %'}, "
          f"{fallbacks} attention launches on the fallback; oracle-side median DiT logit spread {np.nanmin(z[f'spread_median_{case}'][:28]):.0f}-{np.nanmax(z[f'spread_median_{case}'][:28]):.0f}, "
          f"byte sum {int(got[0].astype(np.int64).sum())} vs {int(z[f'sum_{case}'])}")
    assert p >= 44.5 and min(per) >= 41.0 and fallbacks == 0 and (rel is None or rel <= 0.015)
    assert abs(int(got[0].astype(np.int64).sum()) - int(z[f"sum_{case}"])) <= 0.002 * int(z[f"sum_{case}"])


def test_fp8_auto_operand_set_on_seeded_and_stress_weights(full_models):
    """the fp8 operand set is chosen ON THE LOADED WEIGHTS (instarevive_amd/fp8_select.py: one 512 x 512 calibration image, every
    part alone against the bf16 pass, a part keeps its qualified cost only while it deviates as it did when it was qualified). Gates, against the fp32
    oracle's fixtures: on the seeded weights the chosen set is round 5's qualified set and holds >= 46.3 dB on the 2048 x 2048 crops (north_star's 0.1 dB
    at a 30 dB reference); on the stress weights (heavy-tailed channels, peaky attention: where that constant set lost 9 dB) the chosen set stays
    within 1.3 dB of the bf16 path. A D_REF that no longer describes the kernels (deviation of a part on the seeded weights off by more than 12 %) fails
    here, before it can skew a choice."""
    import bench
    from instarevive_amd import _lib as L
    from instarevive_amd import fp8_select as F
    from instarevive_amd.pipeline import process
    from tests.golden.make_headline_crops import CROP, inputs_for
    from tests.support.stress_weights import stress_state_dicts
    swin, vae, dit, sds, y, mask = full_models
    ctx = dit.ctx
    yc, mc = full_models.y_cuda, full_models.mask_cuda
    kw = dict(preprocess_model=swin, vae=vae, y=yc, y_mask=mc)
    logs = []
    # ---- seeded weights
    dev = F.measure_parts(swin, vae, dit, yc, mc)
    worst = max(abs(dev[b] / F.D_REF[b] - 1.0) for b in F.D_REF)
    print("fp8 auto, seeded weights: deviation / D_REF per part: " + ", ".join(f"{n} {dev[b] / F.D_REF[b]:.3f}" for b, n, _ in F.PARTS))
    assert worst <= 0.12, "fp8_select.D_REF is stale: re-run tools/fp8_auto_calib.py and update the constants"
    m_seed = F.auto_mask(swin, vae, dit, yc, mc, log=logs.append, use_cache=False)
    key_seed = F.weights_key(dit, vae)
    assert m_seed == L.FP8_MASK_QUALIFIED, hex(m_seed)
    z = np.load(os.path.join(G, "headline_crops.npz"))
    img = inputs_for(2048)
    pos, want = z["pos_2048"], z["crops_2048"]
    take = lambda a: np.stack([a[yy:yy + CROP, xx:xx + CROP] for yy, xx in pos])
    vae.enable_fp8(True)
    try:
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, m_seed), "ir_set_fp8_mask")
        f8, _ = process(dit, [img], 1, "wavelet", False, False, 512, 448, fp8=True, **kw)
    finally:
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, L.FP8_MASK_DEFAULT), "ir_set_fp8_mask")
        vae.enable_fp8(False)
    p_seed = _psnr(take(f8[0]), want)
    # ---- stress weights
    zs = np.load(os.path.join(G, "stress_512.npz"))
    gains = {"dit": [float(v) for v in zs["logit_gain_dit"]], "vae_encoder": float(zs["logit_gain_vae"][0]), "vae_decoder": float(zs["logit_gain_vae"][1])}
    st = stress_state_dicts(sds, float(zs["frac"]), float(zs["gain"]), gains)
    simg = bench.synthetic_lq(1, 512, 512, int(zs["lq_seed"]))[0].numpy()
    try:
        vae.load_state_dict(st["vae"])
        dit.load_state_dict(st["dit"])
        dit.invalidate_prompt()
        assert F.weights_key(dit, vae) != key_seed, "the cache key must tell the two weight sets apart"
        m_stress = F.auto_mask(swin, vae, dit, yc, mc, log=logs.append, use_cache=False)
        bf, _ = process(dit, [simg], 1, "wavelet", False, False, 512, 448, **kw)
        if m_stress:
            vae.enable_fp8(True)
            try:
                ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, m_stress), "ir_set_fp8_mask")
                s8, _ = process(dit, [simg], 1, "wavelet", False, False, 512, 448, fp8=True, **kw)
            finally:
                ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, L.FP8_MASK_DEFAULT), "ir_set_fp8_mask")
                vae.enable_fp8(False)
        else:
            s8 = bf   # nothing qualifies: --fp8 default runs bf16 throughout on these weights
    finally:
        vae.enable_fp8(False)
        vae.load_state_dict(sds["vae"])
        dit.load_state_dict(sds["dit"])
        dit.invalidate_prompt()
    p_bf, p_s8 = _psnr(bf[0], zs["pred"]), _psnr(s8[0], zs["pred"])
    for line in logs:
        print(line)
    print(f"fp8 auto: seeded weights -> {m_seed:#x}, {p_seed:.2f} dB on the 2048 x 2048 oracle crops (gate 46.3); stress weights -> {m_stress:#x}, "
          f"{p_s8:.2f} dB against the oracle at 512 x 512 (bf16 {p_bf:.2f}, gate bf16 - 1.3)")
    assert p_seed >= 46.3
    assert p_s8 >= p_bf - 1.3 and not (m_stress & 1), "the DiT self-attention must not be chosen on peaky weights"


def test_tiled_encode_force_fallback_path(full_models):
    """The cross-rank overflow protocol of the sharded encode on the GPU (IR_ENCODE_PART_FORCE_FALLBACK only ran in a CPU mock): part 0
    with the fallback FORCED (a rank whose own rows did not overflow, after another rank's did) must (a) report the flag as set, (b) deliver ALL
    rows of attn_o from the rescaling kernel whatever row range it was given - two ranks' forced calls are bit-identical -, (c) agree with the
    fixed-reference kernel's rows to bf16 accuracy, and (d) continue through part 1 to an init latent within 1 % of the unsharded encode's. The
    un-forced sharded form stays bit-identical to the unsharded encode."""
    import bench
    from instarevive_amd.pipeline import HipTileEngine
    swin, vae, dit, sds, y, mask = full_models
    eng = HipTileEngine(dit, vae, swin, full_models.y_cuda, full_models.mask_cuda, "wavelet", False, 512, 448)
    img = [bench.synthetic_lq(1, 1024, 1024, 55)[0].numpy()]
    assert eng.can_shard_encode(img)
    T = (1024 // 8) ** 2
    control0, init0 = eng.encode(img)
    control0, init0 = control0.clone(), init0.clone()
    # un-forced: the two halves' rows combined, then part 1
    ca, oa, ra = eng.encode_part0(img, 0, T // 2)
    assert eng.encode_overflow() == 0
    oa, ra = oa.clone(), ra.clone()
    cb, ob, rb = eng.encode_part0(img, T // 2, T)
    ob[: T // 2] = oa[: T // 2]
    init_sharded = eng.encode_part1(cb, ob, rb).clone()
    assert torch.equal(cb, control0) and torch.equal(init_sharded, init0), "the sharded encode must be bit-identical to the unsharded one"
    normal_rows = ob.clone()
    # forced fallback, as the two ranks would call it
    c1, o1, r1 = eng.encode_part0(img, 0, T // 2, force_fallback=True)
    assert eng.encode_overflow() == 1, "a forced fallback reports the flag as set"
    o1 = o1.clone()
    c2, o2, r2 = eng.encode_part0(img, T // 2, T, force_fallback=True)
    assert torch.equal(o1, o2), "with the fallback forced every rank holds ALL rows from the rescaling kernel"
    rel_rows = float((o2.float() - normal_rows.float()).norm() / normal_rows.float().norm())
    init_forced = eng.encode_part1(c2, o2, r2)
    rel = float((init_forced - init0).norm() / init0.norm())
    print(f"forced fallback: attention rows vs the fixed-reference kernel's rel-L2 {rel_rows:.5f}, init latent vs the unsharded encode rel-L2 {rel:.5f}")
    assert 0 < rel_rows <= 0.01 and rel <= 0.01
    assert float(init_forced.abs().max()) > 0


def test_process_stream_fp8_equals_process_fp8(full_models):
    """cfg-5 from the streaming entry the command line uses (inference.py --fp8): process_stream(fp8=True) over three 512 x 512 images gives, image by
    image, exactly what process(fp8=True) gives, differs from the bf16 stream, and refuses to run without the fp8 weight forms."""
    import bench
    from instarevive_amd.pipeline import process, process_stream
    swin, vae, dit, sds, y, mask = full_models
    imgs = [bench.synthetic_lq(1, 512, 512, 90 + i)[0].numpy() for i in range(3)]
    kw = dict(preprocess_model=swin, vae=vae, y=full_models.y_cuda, y_mask=full_models.mask_cuda)
    vae.enable_fp8(False)
    vae.load_state_dict(sds["vae"])   # the precondition of the guard below, made explicit: a VAE whose fp8 weight forms are NOT uploaded,
    assert not vae.__dict__.get("_fp8_uploaded")   # whatever ran before in this session (the fixture is session-scoped; enable_fp8(False) keeps the forms)
    with pytest.raises(RuntimeError):
        list(process_stream(dit, ([im] for im in imgs), "wavelet", False, False, 512, 448, fp8=True, **kw))
    bf = [p[0] for p, _ in process_stream(dit, ([im] for im in imgs), "wavelet", False, False, 512, 448, **kw)]
    vae.enable_fp8(True)
    try:
        dit.ctx.check(dit.ctx.lib.ir_set_fp8(dit.ctx.h, 0), "ir_set_fp8")   # the mode is switched per call
        one = [process(dit, [im], 1, "wavelet", False, False, 512, 448, fp8=True, **kw)[0][0] for im in imgs]
        st = [p[0] for p, _ in process_stream(dit, ([im] for im in imgs), "wavelet", False, False, 512, 448, fp8=True, **kw)]
    finally:
        vae.enable_fp8(False)
    for a, b, c in zip(st, one, bf):
        assert np.array_equal(a, b) and not np.array_equal(a, c)
    print(f"process_stream fp8 vs bf16: {_psnr(np.stack(st), np.stack(bf)):.2f} dB")
