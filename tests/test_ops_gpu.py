"""Kernel-level parity: every HIP kernel is called through the C ABI and compared with a plain PyTorch fp32
restatement of the same op on the same (bf16-rounded) inputs. Tolerances are stated per test: inputs/outputs are
bf16 (8 mantissa bits, rel. 2^-9 per rounding), accumulation is fp32."""
import math

import pytest
import torch
import torch.nn.functional as F

from instarevive_amd import _lib as L

pytestmark = pytest.mark.gpu


def rb(t):  # round to bf16 and back (fp32, cpu)
    return t.to(torch.bfloat16).to(torch.float32)


_KEEP = []


def dev_bf16(t):
    return L.bf16_bits(t).cuda()


def P(t):
    """Pointer of a device tensor that stays referenced (a temporary freed before the launch could be re-used by the
    next .cuda() in the same argument list)."""
    if t is None:
        return None
    _KEEP.append(t)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return L.ptr(t)


def close(got, ref, rtol, atol, what=""):
    err = (got - ref).abs()
    bad = err > (atol + rtol * ref.abs())
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max abs err {err.max():.4g}, ref max {ref.abs().max():.4g}"


def pack_conv(w, cin_pad, cout_pad):
    """[Cout][Cin][kh][kw] -> [Cout_pad][kh*kw][Cin_pad] (tap-major, channel-minor), fp32."""
    co, ci, kh, kw = w.shape
    p = torch.zeros(cout_pad, kh * kw, cin_pad)
    p[:co, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
    return p.reshape(cout_pad, kh * kw * cin_pad)


@pytest.mark.parametrize("m,k,n", [(128, 32, 128), (300, 96, 192), (1000, 1152, 64), (77, 64, 32), (256, 4608, 1152),
                                   (2304, 2304, 128), (2100, 4608, 256)])  # long K loops, whole and ragged M
def test_linear(ctx, m, k, n):
    g = torch.Generator().manual_seed(m + k + n)
    x = rb(torch.randn(m, k, generator=g))
    w = rb(torch.randn(n, k, generator=g) / math.sqrt(k))
    b = torch.randn(n, generator=g)
    ref = x @ w.t() + b
    out = torch.empty(m, n, dtype=torch.float32, device="cuda")
    xd, wd, bd = dev_bf16(x), dev_bf16(w), b.cuda()
    ctx.check(ctx.lib.ir_op_linear(ctx.h, ctx.stream(), P(xd), P(wd), P(bd), P(out), m, k, n, n, L.ACT_NONE,
                                   None, None, 0, 1, 1.0), "linear")
    torch.cuda.synchronize()
    close(out.cpu(), ref, 1e-4, 2e-4 * math.sqrt(k), "linear f32")  # fp32 accumulate of exact bf16 products


@pytest.mark.parametrize("m,k,n", [(16384, 1152, 1152), (12545, 1152, 3456), (16384, 4608, 1152), (6400, 128, 4608), (65536, 192, 576), (25000, 192, 576)])
def test_linear_big_tile(ctx, m, k, n):
    """Shapes the 256 x 288 ping-pong GEMM takes (Cout % 288 == 0, >= 192 workgroups): the DiT linears at 2048 px, whole and ragged M,
    short and long K, with every epilogue form the DiT uses."""
    g = torch.Generator().manual_seed(m + k + n)
    x = rb(torch.randn(m, k, generator=g))
    w = rb(torch.randn(n, k, generator=g) / math.sqrt(k))
    b = torch.randn(n, generator=g)
    gate = torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g)
    y = x @ w.t() + b
    xd, wd, bd = dev_bf16(x), dev_bf16(w), b.cuda()
    out = torch.empty(m, n, dtype=torch.float32, device="cuda")
    ctx.check(ctx.lib.ir_op_linear(ctx.h, ctx.stream(), P(xd), P(wd), P(bd), P(out), m, k, n, n, L.ACT_NONE, None, None, 0, 1, 1.0), "linear")
    close(out.cpu(), y, 1e-4, 2e-4 * math.sqrt(k), "big tile f32")
    o16 = torch.empty(m, n, dtype=torch.int16, device="cuda")
    ctx.check(ctx.lib.ir_op_linear(ctx.h, ctx.stream(), P(xd), P(wd), P(bd), P(o16), m, k, n, n, L.ACT_GELU_TANH, None, None, 0, 0, 1.0), "gelu")
    close(L.from_bf16_bits(o16).cpu(), F.gelu(y, approximate="tanh"), 2 ** -7, 1e-3, "big tile gelu tanh bf16")
    resd = res.cuda().clone()
    ctx.check(ctx.lib.ir_op_linear(ctx.h, ctx.stream(), P(xd), P(wd), P(bd), P(resd), m, k, n, n, L.ACT_NONE, P(gate.cuda()), P(resd), 1, 1, 1.0),
              "gate res")
    close(resd.cpu(), res + gate * y, 1e-4, 1e-3, "big tile gate + fp32 residual in place")
    rbf = rb(res)
    ctx.check(ctx.lib.ir_op_linear(ctx.h, ctx.stream(), P(xd), P(wd), None, P(o16), m, k, n, n, L.ACT_NONE, None, P(dev_bf16(rbf)), 0, 0, 0.5), "res bf16")
    close(L.from_bf16_bits(o16).cpu(), rbf + 0.5 * (y - b), 2 ** -7, 1e-3, "big tile bf16 residual + out_scale")


def test_linear_epilogues(ctx):
    g = torch.Generator().manual_seed(5)
    m, k, n = 260, 128, 96
    x = rb(torch.randn(m, k, generator=g))
    w = rb(torch.randn(n, k, generator=g) / math.sqrt(k))
    b = torch.randn(n, generator=g)
    gate = torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g)
    xd, wd = dev_bf16(x), dev_bf16(w)
    # gelu-tanh, bf16 out
    out = torch.empty(m, n, dtype=torch.int16, device="cuda")
    ctx.check(ctx.lib.ir_op_linear(ctx.h, ctx.stream(), P(xd), P(wd), P(b.cuda()), P(out), m, k, n, n,
                                   L.ACT_GELU_TANH, None, None, 0, 0, 1.0), "linear gelu")
    ref = F.gelu(x @ w.t() + b, approximate="tanh")
    close(L.from_bf16_bits(out).cpu(), ref, 2 ** -7, 1e-3, "gelu tanh bf16")
    # gelu-erf
    ctx.check(ctx.lib.ir_op_linear(ctx.h, ctx.stream(), P(xd), P(wd), P(b.cuda()), P(out), m, k, n, n,
                                   L.ACT_GELU_ERF, None, None, 0, 0, 1.0), "linear gelu erf")
    close(L.from_bf16_bits(out).cpu(), F.gelu(x @ w.t() + b), 2 ** -7, 1e-3, "gelu erf bf16")
    # gate * (acc+b) + fp32 residual, fp32 out, in place on the residual buffer
    resd = res.cuda().clone()
    ctx.check(ctx.lib.ir_op_linear(ctx.h, ctx.stream(), P(xd), P(wd), P(b.cuda()), P(resd), m, k, n, n,
                                   L.ACT_NONE, P(gate.cuda()), P(resd), 1, 1, 1.0), "linear gate res")
    close(resd.cpu(), res + gate * (x @ w.t() + b), 1e-4, 1e-3, "gate+residual")
    # bf16 residual, bf16 out, out_scale
    rbf = rb(res)
    ctx.check(ctx.lib.ir_op_linear(ctx.h, ctx.stream(), P(xd), P(wd), None, P(out), m, k, n, n, L.ACT_NONE, None,
                                   P(dev_bf16(rbf)), 0, 0, 0.5), "linear res bf16")
    close(L.from_bf16_bits(out).cpu(), rbf + 0.5 * (x @ w.t()), 2 ** -7, 1e-3, "bf16 residual + out_scale")


CONV_CASES = [
    # n, h, w, cin, cout, stride, pad, up
    (1, 16, 16, 32, 64, 1, 1, 0),
    (2, 24, 40, 64, 128, 1, 1, 0),
    (1, 16, 24, 32, 32, 2, 0, 0),   # VAE Downsample: pad (0,1,0,1), stride 2
    (1, 12, 20, 64, 64, 1, 1, 1),   # nearest x2 folded in
    (1, 32, 32, 128, 256, 1, 1, 0),
    (1, 20, 28, 64, 3, 1, 1, 0),    # Cout=3 (padded to 32), scalar store path
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3(ctx, case):
    n, h, w, cin, cout, stride, pad, up = case
    g = torch.Generator().manual_seed(sum(case))
    x = rb(torch.randn(n, cin, h, w, generator=g))
    wt = rb(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = torch.randn(cout, generator=g)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    if stride == 2:
        ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), wt, b, stride=2)
    else:
        ref = F.conv2d(xin, wt, b, padding=1)
    cout_pad = (cout + 31) // 32 * 32
    wp = pack_conv(wt, cin, cout_pad)
    bp = torch.zeros(cout_pad)
    bp[:cout] = b
    ho, wo = ref.shape[-2:]
    out = torch.empty(n, ho, wo, cout, dtype=torch.float32, device="cuda")
    xd = dev_bf16(x.permute(0, 2, 3, 1).contiguous())
    ctx.check(ctx.lib.ir_op_conv(ctx.h, ctx.stream(), P(xd), P(dev_bf16(wp)), P(bp.cuda()), P(out), n, h, w, cin, cout,
                                 cout_pad, 9, stride, pad, up, L.ACT_NONE, 0.0, None, 0, 1), "conv")
    torch.cuda.synchronize()
    close(out.cpu().permute(0, 3, 1, 2), ref, 1e-4, 1e-3, f"conv {case}")


@pytest.mark.parametrize("n,h,w,act", [(1, 256, 256, "lrelu"), (2, 260, 300, "lrelu"), (1, 264, 288, "none")])
def test_conv64_full_resolution(ctx, n, h, w, act):
    """conv64_kernel (vae_io.hip, ir_op_conv64): SwinIR's conv_hr form - 3x3, 64 -> 64, bf16 NHWC in and out, bias + LeakyReLU(0.2) - on whole and
    ragged 8 x 32 tiles and two images, against F.conv2d on the same bf16 operands; and against conv_halo_kernel, which the pipeline runs by default
    (ir_op_conv): the same accuracy (RMS error against the fp32 reference within 2 %), not the same bits - another summation order."""
    g = torch.Generator().manual_seed(h + w)
    x = rb(torch.randn(n, 64, h, w, generator=g))
    wt = rb(torch.randn(64, 64, 3, 3, generator=g) / math.sqrt(9 * 64))
    b = torch.randn(64, generator=g)
    ref = F.conv2d(x, wt, b, padding=1)
    code = L.ACT_NONE
    if act == "lrelu":
        ref, code = F.leaky_relu(ref, 0.2), L.ACT_LRELU
    xd = dev_bf16(x.permute(0, 2, 3, 1).contiguous())
    wp = dev_bf16(pack_conv(wt, 64, 64))
    out_new = torch.empty(n, h, w, 64, dtype=torch.int16, device="cuda")
    ctx.check(ctx.lib.ir_op_conv64(ctx.h, ctx.stream(), P(xd), P(wp), P(b.cuda()), P(out_new), n, h, w, code, 0.2), "conv64")
    out_old = torch.empty(n, h, w, 64, dtype=torch.int16, device="cuda")
    ctx.check(ctx.lib.ir_op_conv(ctx.h, ctx.stream(), P(xd), P(wp), P(b.cuda()), P(out_old), n, h, w, 64, 64, 64, 9, 1, 1, 0, code, 0.2, None, 0, 0), "conv")
    torch.cuda.synchronize()
    new, old = (L.from_bf16_bits(o).cpu().permute(0, 3, 1, 2) for o in (out_new, out_old))
    close(new, ref, 2 ** -7, 2e-3, f"conv64 {n}x{h}x{w} {act}")
    close(old, ref, 2 ** -7, 2e-3, f"conv_halo_kernel {n}x{h}x{w} {act}")
    e_new, e_old = float((new - ref).pow(2).mean().sqrt()), float((old - ref).pow(2).mean().sqrt())
    assert e_new <= 1.02 * e_old, f"conv64_kernel rms error {e_new:.3e} against conv_halo_kernel's {e_old:.3e}"


@pytest.mark.parametrize("n,h,w", [(1, 64, 96), (2, 100, 72), (1, 256, 320)])
def test_conv64_to3(ctx, n, h, w):
    """vae_norm_conv_out_kernel<1, false> (ir_op_conv64_to3): SwinIR's conv_last - 3x3, 64 -> 3, fp32 [pixel][4] out - against F.conv2d on the same bf16
    operands and against the generic implicit GEMM it replaces from 1024 x 1024 up (ir_op_conv with Cout = 3 padded to 32): the same accuracy."""
    g = torch.Generator().manual_seed(3 * h + w)
    x = rb(torch.randn(n, 64, h, w, generator=g))
    wt = rb(torch.randn(3, 64, 3, 3, generator=g) / math.sqrt(9 * 64))
    b = torch.randn(3, generator=g)
    ref = F.conv2d(x, wt, b, padding=1)
    xd = dev_bf16(x.permute(0, 2, 3, 1).contiguous())
    wp = dev_bf16(pack_conv(wt, 64, 32))
    bp = torch.zeros(32)
    bp[:3] = b
    out_new = torch.empty(n, h, w, 4, dtype=torch.float32, device="cuda")
    ctx.check(ctx.lib.ir_op_conv64_to3(ctx.h, ctx.stream(), P(xd), P(wp), P(bp.cuda()), P(out_new), n, h, w), "conv64_to3")
    out_old = torch.empty(n, h, w, 3, dtype=torch.float32, device="cuda")
    ctx.check(ctx.lib.ir_op_conv(ctx.h, ctx.stream(), P(xd), P(wp), P(bp.cuda()), P(out_old), n, h, w, 64, 3, 32, 9, 1, 1, 0, L.ACT_NONE, 0.0, None, 0, 1), "conv")
    torch.cuda.synchronize()
    new, old = out_new.cpu()[..., :3].permute(0, 3, 1, 2), out_old.cpu().permute(0, 3, 1, 2)
    close(new, ref, 1e-4, 1e-3, f"conv64_to3 {n}x{h}x{w}")
    e_new, e_old = float((new - ref).pow(2).mean().sqrt()), float((old - ref).pow(2).mean().sqrt())
    assert e_new <= 1e-6 and e_old <= 1e-6, f"rms error {e_new:.3e} / the generic kernel's {e_old:.3e}: fp32 out, both at the fp32 summation-order level"
    assert float(out_new[..., 3].abs().max()) == 0.0


@pytest.mark.parametrize("n,h,w,cin,cout,res,s1", [
    (1, 16, 16, 128, 128, False, False), (2, 24, 40, 256, 128, True, False), (1, 20, 28, 512, 64, False, False),   # conv_halo_kernel<.., FP8>
    (1, 128, 256, 128, 128, True, True),     # conv_halo_s1_fp8_kernel: one period of 9 MFMA steps, whole patches, residual
    (2, 200, 488, 128, 128, True, True),     # ragged 16 x 32 patches, two images
    (1, 100, 120, 256, 256, False, True),    # two periods, two channel tiles
    (1, 128, 128, 512, 128, True, True)])    # four periods
def test_conv3x3_fp8(ctx, n, h, w, cin, cout, res, s1):
    """3x3 conv on OCP e4m3 operands through the MX-scaled MFMAs (conv_halo_kernel<.., FP8> and, from 32 patch tiles per image up,
    conv_halo_s1_fp8_kernel; BASELINE.json configs[4]). Products of two e4m3 values are exact in fp32, so against a float64 convolution
    of the DEQUANTISED operands only the fp32 accumulation order and the bf16 rounding of the output remain (2^-7 relative + 2e-3)."""
    assert (ctx.lib.ir_op_conv_fp8_route(ctx.h, n, h, w, cin, cout, 1 if res else 0) == 0) == s1
    g = torch.Generator().manual_seed(n + h + w + cin + cout)
    x8 = (torch.randn(n, cin, h, w, generator=g) * 4).clamp(-448, 448).to(torch.float8_e4m3fn)
    w8 = (torch.randn(cout, cin, 3, 3, generator=g) * 64).clamp(-448, 448).to(torch.float8_e4m3fn)
    deq = torch.rand(cout, generator=g) * 1e-3 + 1e-4
    bias = torch.randn(cout, generator=g)
    r_ = rb(torch.randn(n, cout, h, w, generator=g)) if res else None
    ref = F.conv2d(x8.double(), w8.double(), None, padding=1) * deq.double()[None, :, None, None] + bias.double()[None, :, None, None]
    if res:
        ref = ref + r_.double()
    xin = x8.permute(0, 2, 3, 1).contiguous().view(torch.uint8).cuda()
    wp = w8.permute(0, 2, 3, 1).contiguous().view(torch.uint8).cuda()      # [Cout][tap][Cin]
    out = torch.empty(n, h, w, cout, dtype=torch.int16, device="cuda")
    rd = dev_bf16(r_.permute(0, 2, 3, 1).contiguous()) if res else None
    ctx.check(ctx.lib.ir_op_conv_fp8(ctx.h, ctx.stream(), P(xin), P(wp), P(deq.cuda()), P((bias / deq).cuda()), P(out), n, h, w, cin, cout, P(rd)), "conv_fp8")
    torch.cuda.synchronize()
    close(L.from_bf16_bits(out).cpu().permute(0, 3, 1, 2).double(), ref, 2 ** -7, 2e-3, "conv fp8")


def test_conv_lrelu_residual_bf16(ctx):
    g = torch.Generator().manual_seed(11)
    n, h, w, cin, cout = 1, 16, 16, 64, 64
    x = rb(torch.randn(n, cin, h, w, generator=g))
    wt = rb(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = torch.randn(cout, generator=g)
    res = rb(torch.randn(n, cout, h, w, generator=g))
    y = F.conv2d(x, wt, b, padding=1)
    ref = F.leaky_relu(y, 0.2) + res
    out = torch.empty(n, h, w, cout, dtype=torch.int16, device="cuda")
    ctx.check(ctx.lib.ir_op_conv(ctx.h, ctx.stream(), P(dev_bf16(x.permute(0, 2, 3, 1).contiguous())), P(dev_bf16(pack_conv(wt, cin, cout))),
                                 P(b.cuda()), P(out), n, h, w, cin, cout, cout, 9, 1, 1, 0, L.ACT_LRELU, 0.2,
                                 P(dev_bf16(res.permute(0, 2, 3, 1).contiguous())), 0, 0), "conv lrelu")
    close(L.from_bf16_bits(out).cpu().permute(0, 3, 1, 2), ref, 2 ** -7, 2e-3, "conv lrelu + residual")


@pytest.mark.parametrize("n,hw,c,silu", [(1, 64 * 64, 128, 1), (2, 40 * 24, 256, 1), (1, 16 * 16, 512, 0), (1, 100 * 100, 32, 1)])
def test_groupnorm(ctx, n, hw, c, silu):
    g = torch.Generator().manual_seed(c + hw)
    x = rb(torch.randn(n, hw, c, generator=g) * 2 + 0.5)
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    ref = F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, eps=1e-6).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    y = torch.empty(n, hw, c, dtype=torch.int16, device="cuda")
    ws = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.ir_op_groupnorm(ctx.h, ctx.stream(), P(dev_bf16(x)), P(y), P(gamma.cuda()), P(beta.cuda()), n, hw, c,
                                      32, 1e-6, silu, P(ws), ws.numel()), "groupnorm")
    close(L.from_bf16_bits(y).cpu(), ref, 2 ** -7, 4e-3, "groupnorm")


@pytest.mark.parametrize("n,h,w,cin,cout,stride,up,res,expect_fused", [
    (1, 64, 64, 128, 128, 1, 0, True, True),      # halo kernel, whole tiles
    (2, 40, 24, 128, 256, 1, 0, False, True),     # halo kernel, partial tiles at the image border, two images
    (1, 16, 16, 256, 512, 1, 1, False, True),     # upsample conv -> 32 x 32
    (1, 32, 32, 128, 128, 2, 0, False, True),     # stride-2 downsample (generic implicit GEMM): 256 rows per image = 2 tiles
    (1, 24, 24, 128, 128, 2, 0, False, False),    # 144 rows per image: a tile would straddle images -> separate statistics pass
    (1, 16, 16, 64, 64, 1, 0, False, False),      # 2 channels per group: below the 4-channel vector of the epilogue -> not fused
])
def test_conv_groupnorm_fused(ctx, n, h, w, cin, cout, stride, up, res, expect_fused):
    """ResnetBlock's conv -> GroupNorm(32)+SiLU with the statistics produced by the conv epilogue (ldm model.py:131-151)."""
    _conv_groupnorm_case(ctx, n, h, w, cin, cout, stride, up, res, expect_fused)


@pytest.mark.parametrize("n,h,w,cin,cout,up,res", [
    (2, 200, 488, 128, 128, 0, True),     # ragged 16 x 32 patches (200 = 12.5 x 16, 488 = 15.25 x 32), two images, residual
    (1, 100, 120, 256, 256, 1, False),    # nearest-2x folded in -> 200 x 240, two channel tiles, 8 chunks
    (1, 256, 384, 512, 128, 0, True),     # 16 chunks, whole patches
    (3, 64, 128, 256, 512, 0, True),      # four channel tiles, 16 channels per group (group statistics combine four 4-channel units), three images
])
def test_conv_s1_kernel(ctx, n, h, w, cin, cout, up, res):
    """conv_halo_s1_kernel (one wave per SIMD, 16 x 32 patches x 128 channels): sizes with enough patches to be routed to it — the fused
    statistics come back as one partial per 16 x 32 patch, which is how the test knows which kernel ran."""
    ho, wo = (2 * h, 2 * w) if up else (h, w)
    _conv_groupnorm_case(ctx, n, h, w, cin, cout, 1, up, res, True, expect_chunks=((ho + 15) // 16) * ((wo + 31) // 32))


@pytest.mark.parametrize("n,h,w,cin,cout,res", [
    (1, 256, 384, 128, 128, False),    # whole patches, borders on all four sides
    (2, 200, 488, 128, 128, True),     # ragged patches, two images with different scale / shift through one workgroup's tile walk
    (1, 128, 384, 256, 256, True),     # 8 chunks, two channel tiles (the halo of a patch is normalised once per channel tile)
    (2, 128, 256, 512, 128, False),    # 16 chunks (the table's full 512 channels), two images
    (1, 1024, 1024, 128, 128, True),   # 2048 patches: eight per workgroup in a row (the stream runs through tile boundaries; round 4's unpinned-wait bug showed only here)
])
def test_conv_s1_norm_in_kernel(ctx, n, h, w, cin, cout, res):
    """conv_halo_s1_kernel<0, 9, NORM>: ResnetBlock's norm -> SiLU -> conv with the apply pass folded into the conv - every halo chunk is taken through
    scale / shift / SiLU in LDS one chunk ahead of its use, padding pixels stay zero AFTER the norm. Against conv2d(bf16(silu(x * scale + shift)))
    with the scale / shift of an fp64 GroupNorm over the bf16 input."""
    g = torch.Generator().manual_seed(h * w + cin)
    x = rb(torch.randn(n, cin, h, w, generator=g) * 1.7 + 0.3)
    wt = rb(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5))
    b = torch.randn(cout, generator=g) * 0.1
    gamma, beta = torch.randn(cin, generator=g), torch.randn(cin, generator=g)
    xg = x.double().reshape(n, 32, -1)
    mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
    rstd = (var + 1e-6).rsqrt()
    cpg = cin // 32
    scale = (rstd.repeat_interleave(cpg, 1) * gamma.double()).float()                       # [n][cin]
    shift = (beta.double() - mean.repeat_interleave(cpg, 1) * rstd.repeat_interleave(cpg, 1) * gamma.double()).float()
    xn = rb(F.silu(x * scale[:, :, None, None] + shift[:, :, None, None]))                # what gn_apply_kernel would have stored
    co = F.conv2d(xn, wt, b, padding=1)
    r = rb(torch.randn(n, cout, h, w, generator=g)) if res else None
    if res:
        co = co + r
    xin = dev_bf16(x.permute(0, 2, 3, 1).contiguous())
    wp = dev_bf16(wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous())
    rd = dev_bf16(r.permute(0, 2, 3, 1).contiguous()) if res else None
    out = torch.full((n, h, w, cout), 0x7fc0, dtype=torch.int16, device="cuda")
    sd, hd, bd = scale.contiguous().cuda(), shift.contiguous().cuda(), b.cuda()
    outs = []
    for _ in range(2):
        ctx.check(ctx.lib.ir_op_conv_norm(ctx.h, ctx.stream(), P(xin), P(sd), P(hd), P(wp), P(bd), P(rd) if res else None, P(out), n, h, w, cin, cout), "conv_norm")
        torch.cuda.synchronize()
        outs.append(L.from_bf16_bits(out).cpu().permute(0, 3, 1, 2).clone())
    assert torch.equal(outs[0], outs[1]), "not deterministic"
    close(outs[0], rb(co), 2 ** -7, 6e-3, "conv with the GroupNorm apply + SiLU inside")


def _conv_groupnorm_case(ctx, n, h, w, cin, cout, stride, up, res, expect_fused, expect_chunks=None):
    import ctypes
    g = torch.Generator().manual_seed(h * w + cout + stride)
    x = rb(torch.randn(n, cin, h, w, generator=g))
    wt = rb(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5))
    b = torch.randn(cout, generator=g) * 0.1
    gamma, beta = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    if stride == 2:
        co = F.conv2d(F.pad(xi, (0, 1, 0, 1)), wt, b, stride=2)
    else:
        co = F.conv2d(xi, wt, b, padding=1)
    ho, wo = co.shape[-2:]
    r = rb(torch.randn(n, cout, ho, wo, generator=g)) if res else None
    if res:
        co = co + r
    co = rb(co)  # the conv output is stored in bf16 and GroupNorm normalises what is stored
    ref = F.silu(F.group_norm(co, 32, gamma, beta, eps=1e-6))
    xin = dev_bf16(x.permute(0, 2, 3, 1).contiguous())
    wp = dev_bf16(wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous())
    rd = dev_bf16(r.permute(0, 2, 3, 1).contiguous()) if res else None
    conv_out = torch.empty(n, ho, wo, cout, dtype=torch.int16, device="cuda")
    y = torch.empty_like(conv_out)
    ws = torch.empty(16 << 20, dtype=torch.uint8, device="cuda")
    fused = ctypes.c_int(-1)
    bd, gd, btd = b.cuda(), gamma.cuda(), beta.cuda()
    ctx.check(ctx.lib.ir_op_conv_groupnorm(ctx.h, ctx.stream(), P(xin), P(wp), P(bd), P(conv_out), P(y), P(gd), P(btd), n, h, w, cin, cout,
                                           stride, up, P(rd) if res else None, 1, P(ws), ws.numel(), ctypes.byref(fused)), "conv_groupnorm")
    torch.cuda.synchronize()
    assert (fused.value > 0) == expect_fused, fused.value
    if expect_chunks is not None:
        assert fused.value == expect_chunks, (fused.value, expect_chunks)
    got_conv = L.from_bf16_bits(conv_out).cpu().permute(0, 3, 1, 2)
    close(got_conv, co, 2 ** -7, 4e-3, "conv (fused-GN launch)")
    # GroupNorm normalises the tensor as stored: reference statistics from the kernel's own conv output (an ulp flip of the conv
    # against the CPU reference must not count against the normalisation), plus a looser end-to-end check against the CPU chain
    ref_own = F.silu(F.group_norm(got_conv, 32, gamma, beta, eps=1e-6))
    close(L.from_bf16_bits(y).cpu().permute(0, 3, 1, 2), ref_own, 2 ** -7, 4e-3, "groupnorm from fused statistics")
    close(L.from_bf16_bits(y).cpu().permute(0, 3, 1, 2), ref, 2 ** -5, 3e-2, "conv -> groupnorm fused")
    # run-to-run determinism of the fixed-order reductions
    y2 = torch.empty_like(y)
    ctx.check(ctx.lib.ir_op_conv_groupnorm(ctx.h, ctx.stream(), P(xin), P(wp), P(bd), P(conv_out), P(y2), P(gd), P(btd), n, h, w, cin, cout,
                                           stride, up, P(rd) if res else None, 1, P(ws), ws.numel(), ctypes.byref(fused)), "conv_groupnorm")
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


@pytest.mark.parametrize("rows,c,ld", [(1000, 180, 192), (513, 1152, 1152), (64, 60, 192), (5000, 180, 192), (4104, 64, 64)])  # the last two: 16 lanes per row (layernorm_r16_kernel), ragged row count
def test_layernorm(ctx, rows, c, ld):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, ld, generator=g) * 3 + 1
    a, b = torch.randn(c, generator=g), torch.randn(c, generator=g)
    eps = 1e-5
    ref = torch.zeros(rows, ld)
    ref[:, :c] = F.layer_norm(x[:, :c], (c,), None, None, eps) * a + b
    y = torch.full((rows, ld), 0x7fff, dtype=torch.int16, device="cuda")
    ctx.check(ctx.lib.ir_op_layernorm(ctx.h, ctx.stream(), P(x.cuda()), P(y), P(a.cuda()), P(b.cuda()), rows, c, ld, ld, eps),
              "layernorm")
    close(L.from_bf16_bits(y).cpu(), ref, 2 ** -7, 2e-3, "layernorm")


@pytest.mark.parametrize("b,heads,tq,tk,d,bias", [
    (1, 2, 128, 128, 72, False), (2, 3, 200, 200, 72, False), (1, 16, 1024, 1024, 72, False),
    (2, 2, 384, 320, 72, False), (1, 1, 256, 64, 72, False), (1, 2, 300, 128, 72, False),  # ping-pong kernel: ragged Tq, odd / single tile counts
    (2, 2, 130, 300, 72, True), (1, 2, 64, 64, 32, False), (1, 1, 256, 192, 64, True),
    (1, 1, 256, 256, 512, False), (2, 1, 1024, 1024, 512, False), (3, 1, 448, 448, 512, False)])  # d = 512: batched, ragged last query block
def test_flash_attention(ctx, b, heads, tq, tk, d, bias):
    g = torch.Generator().manual_seed(tq + tk + d)
    q = rb(torch.randn(b, tq, heads, d, generator=g))
    k = rb(torch.randn(b, tk, heads, d, generator=g))
    v = rb(torch.randn(b, tk, heads, d, generator=g))
    kb = torch.randn(b, tk, generator=g) if bias else None
    scale = d ** -0.5
    mask = kb[:, None, None, :] if bias else None
    ref = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=mask, scale=scale).transpose(1, 2)
    o = torch.empty(b, tq, heads, d, dtype=torch.int16, device="cuda")
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), P(dev_bf16(q)), P(dev_bf16(k)), P(dev_bf16(v)), P(o), b, heads, tq, tk,
                                      d, scale, P(kb.cuda()) if bias else None, P(ws), ws.numel()), "attention")
    # P is rounded to bf16 before PV and O is stored in bf16: tolerance a few bf16 ulps of |v|~1 averages
    close(L.from_bf16_bits(o).cpu(), ref, 2 ** -6, 6e-3, "flash attention")


@pytest.mark.parametrize("b,heads,tq,tk,bias", [
    (1, 16, 1024, 300, "quirk"), (1, 16, 4096, 300, "mask"), (2, 16, 1000, 320, "none"), (3, 16, 512, 40, "quirk"), (1, 16, 2048, 129, "mask"),
    (1, 16, 1024, 300, "spike"), (1, 16, 16384, 300, "quirk"), (6, 16, 1024, 300, "mask"), (25, 16, 1024, 300, "quirk"), (2, 16, 4096, 300, "spike")])
def test_cross_attention_persistent(ctx, b, heads, tq, tk, bias):
    """flash_attn_x72_kernel (>= 64 items of 256 queries, <= 320 keys): the DiT cross-attention with the head's K / V^T resident in LDS and the
    additive key bias riding in K dims 72 / 73. quirk = the 0 / 1 mask ADDED to the logits (what the reference's 3-D mask does in diffusers),
    mask = 0 / -10000, spike = a late key 200 above everything (beyond the fixed reference: the wave repeats its item with the exact row maxima).
    The last four shapes give a workgroup 2..7 items in a row (the first ones one each): K / V^T reuse, the Q prefetch and the item-to-item state."""
    d = 72
    g = torch.Generator().manual_seed(tq + tk)
    q = rb(torch.randn(b, tq, heads, d, generator=g))
    k = rb(torch.randn(b, tk, heads, d, generator=g))
    v = rb(torch.randn(b, tk, heads, d, generator=g))
    kb = None
    if bias == "quirk":
        kb = (torch.rand(b, tk, generator=g) < 0.3).float()
    elif bias == "mask":
        kb = (torch.rand(b, tk, generator=g) < 0.4).float() * -10000.0
        kb[:, 0] = 0
    elif bias == "spike":
        kb = torch.zeros(b, tk)
        kb[:, tk - 7] = 200.0
    scale = d ** -0.5
    mask = kb[:, None, None, :].double() if kb is not None else None
    ref = F.scaled_dot_product_attention(q.double().transpose(1, 2), k.double().transpose(1, 2), v.double().transpose(1, 2), attn_mask=mask, scale=scale).transpose(1, 2).float()
    qd, kd, vd = dev_bf16(q), dev_bf16(k), dev_bf16(v)
    kbd = kb.cuda() if kb is not None else None
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(2):
        o = torch.full((b, tq, heads, d), 0x7fc0, dtype=torch.int16, device="cuda")
        ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), P(qd), P(kd), P(vd), P(o), b, heads, tq, tk, d, scale, P(kbd) if kbd is not None else None,
                                          P(ws), ws.numel()), "attention")
        torch.cuda.synchronize()
        outs.append(L.from_bf16_bits(o).cpu())
    assert torch.equal(outs[0], outs[1]), "the persistent cross-attention must be deterministic run to run"
    close(outs[0], ref, 2 ** -6, 6e-3, "persistent cross-attention")


def test_cross_attention_persistent_random_shapes(ctx):
    """flash_attn_x72_kernel over twelve random shapes of its domain (ragged query counts, 1..5 key tiles, 4..24 heads, 1..9 images, bias forms mixed,
    per-image keys): each against float64 softmax attention, and twice for run-to-run identity."""
    import numpy as np
    rng = np.random.default_rng(20261004)
    d, done = 72, 0
    while done < 12:
        heads, b = int(rng.integers(4, 25)), int(rng.integers(1, 10))
        tq, tk = int(rng.integers(256, 3000)), int(rng.integers(1, 321))
        if b * heads * ((tq + 255) // 256) < 64:
            continue
        done += 1
        g = torch.Generator().manual_seed(1000 + done)
        q = rb(torch.randn(b, tq, heads, d, generator=g) * float(rng.uniform(0.5, 2.0)))
        k = rb(torch.randn(b, tk, heads, d, generator=g))
        v = rb(torch.randn(b, tk, heads, d, generator=g))
        form = done % 3
        kb = None if form == 0 else ((torch.rand(b, tk, generator=g) < 0.5).float() * (1.0 if form == 1 else -10000.0))
        if kb is not None and form == 2:
            kb[:, 0] = 0
        scale = d ** -0.5
        mask = kb[:, None, None, :].double() if kb is not None else None
        qt, kt = q.double().transpose(1, 2), k.double().transpose(1, 2)
        ref = F.scaled_dot_product_attention(qt, kt, v.double().transpose(1, 2), attn_mask=mask, scale=scale).transpose(1, 2).float()
        # bf16 operands of both products: the probabilities are rounded (2^-9 each) before the second product, and K * scale * log2 e before the
        # first (with |q| up to 10 here that is up to 1 % in a probability): the error of an element scales with sum_j p_j |v_j|, not with |result|,
        # which may be small where terms cancel. Measured: max error / that sum = 1.29 % for this kernel AND for the 4-wave kernel it replaces.
        mag = F.scaled_dot_product_attention(qt, kt, v.double().abs().transpose(1, 2), attn_mask=mask, scale=scale).transpose(1, 2).float()
        qd, kd, vd = dev_bf16(q), dev_bf16(k), dev_bf16(v)
        kbd = kb.cuda() if kb is not None else None
        ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
        outs = []
        for _ in range(2):
            o = torch.full((b, tq, heads, d), 0x7fc0, dtype=torch.int16, device="cuda")
            ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), P(qd), P(kd), P(vd), P(o), b, heads, tq, tk, d, scale, P(kbd) if kbd is not None else None,
                                              P(ws), ws.numel()), "attention")
            torch.cuda.synchronize()
            outs.append(L.from_bf16_bits(o).cpu())
        assert torch.equal(outs[0], outs[1]), f"not deterministic at b={b} heads={heads} tq={tq} tk={tk}"
        err = (outs[0] - ref).abs()
        bad = err > 2 ** -6 * mag + 2e-3
        assert not bad.any(), (f"persistent cross-attention b={b} heads={heads} tq={tq} tk={tk} bias form {form}: {int(bad.sum())}/{bad.numel()} off, "
                               f"max abs err {float(err.max()):.4g}")


@pytest.mark.parametrize("t,gain", [(256, 4.0), (512, 8.0), (512, 12.0)])
def test_flash_attention_spike(ctx, t, gain):
    """A late key dominates one query (the maximum jumps in the last tile). gain 4: inside the fixed-reference range of the
    ping-pong kernel (scores up to 2^64 above the first tile's maximum); gain 8 / 12: beyond it -> its overflow flag and the
    rescaling fallback kernel."""
    g = torch.Generator().manual_seed(3)
    b, heads, d = 1, 1, 72
    q = rb(torch.randn(b, t, heads, d, generator=g))
    k = rb(torch.randn(b, t, heads, d, generator=g))
    v = rb(torch.randn(b, t, heads, d, generator=g))
    k[0, t - 6, 0] = q[0, 7, 0] * gain
    k = rb(k)
    scale = d ** -0.5
    # explicit float64 softmax: torch's CPU SDPA returns NaN for logits this large
    qd64, kd64, vd64 = (x_.transpose(1, 2).double() for x_ in (q, k, v))
    ref = (torch.softmax(qd64 @ kd64.transpose(-1, -2) * scale, dim=-1) @ vd64).transpose(1, 2).float()
    o = torch.empty(b, t, heads, d, dtype=torch.int16, device="cuda")
    ws = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), P(dev_bf16(q)), P(dev_bf16(k)), P(dev_bf16(v)), P(o), b, heads, t, t, d,
                                      scale, None, P(ws), ws.numel()), "attention")
    # logits of +-60..150 (log2 units) carry the bf16 rounding of q and k at full weight: 2^-9 relative is +-0.1..0.3 in the exponent
    # for the few queries whose second-largest key is close to the spike; the extreme case is here for the overflow path, not for ulps
    rtol, atol = (2 ** -6, 6e-3) if gain <= 8 else (2 ** -4, 4e-2)
    close(L.from_bf16_bits(o).cpu(), ref, rtol, atol, "flash attention spike")


def test_flash_attention_spike_recomputes_only_the_flagged_workgroups(ctx):
    """Round 6: flash_attn_pp2_kernel marks the 256-query workgroups whose fixed softmax reference was outgrown and the rescaling kernel behind it
    recomputes only those (before: the whole launch - one peaky row among 16384 x 16 sent all of a layer's queries through the slower kernel). Two heads
    of 1024 tokens = eight workgroups, a late dominating key for ONE query of head 1 (workgroup 2 of that head): that query returns the spiked key's
    value row, every other row is right as well, and the rows of the seven workgroups that did not overflow are exactly what the call without any
    spike gives for them (they were not touched by the fallback: the rescaling kernel's rounding differs in the last bit)."""
    g = torch.Generator().manual_seed(31)
    b, heads, d, t = 1, 2, 72, 1024
    q = rb(torch.randn(b, t, heads, d, generator=g))
    k = rb(torch.randn(b, t, heads, d, generator=g))
    v = rb(torch.randn(b, t, heads, d, generator=g))
    scale = d ** -0.5
    ws = torch.empty(16 << 20, dtype=torch.uint8, device="cuda")

    def run(kk):
        o = torch.empty(b, t, heads, d, dtype=torch.int16, device="cuda")
        ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), P(dev_bf16(q)), P(dev_bf16(kk)), P(dev_bf16(v)), P(o), b, heads, t, t, d, scale, None, P(ws), ws.numel()), "attention")
        return L.from_bf16_bits(o).cpu()
    plain = run(k)
    ks = k.clone()
    ks[0, t - 6, 1] = q[0, 700, 1] * 12.0     # head 1, query 700 (its workgroup: queries 512..767), key t - 6 (the last tile)
    ks = rb(ks)
    got = run(ks)
    qd64, kd64, vd64 = (x_.transpose(1, 2).double() for x_ in (q, ks, v))
    ref = (torch.softmax(qd64 @ kd64.transpose(-1, -2) * scale, dim=-1) @ vd64).transpose(1, 2).float()
    close(got, ref, 2 ** -4, 4e-2, "flash attention spike, partial fallback")
    assert (got[0, 700, 1] - v[0, t - 6, 1]).abs().max() <= 2 ** -6 * v[0, t - 6, 1].abs().max() + 1e-3
    assert torch.equal(got[:, :, 0], plain[:, :, 0]), "head 0 saw no spike: its rows must come from the fixed-reference kernel, untouched"
    # head 1: the spiked key changes every row a little (it is a key of all of them), so compare against the reference instead; the workgroup that
    # overflowed is the only one whose rows went through the rescaling kernel - visible as exact agreement of a re-run
    assert torch.equal(run(ks), got)


@pytest.mark.parametrize("t,gain", [(512, 4.0), (512, 12.0)])
def test_flash_attention_fp8_spike(ctx, t, gain):
    """flash_attn_fp8_kernel with its fixed softmax reference: a late key dominates one query. gain 4: the spike is about 2^40 above the first
    tile's maximum - inside the range the exponent bytes carry (probabilities relative to the fixed reference up to ~2^100); gain 12: beyond
    it -> overflow flag -> the bf16 V^T is built (transpose_v with only_if) and the rescaling bf16 kernel recomputes everything. Either way
    the spiked query must return the spiked key's value row (e4m3 rounding of V: 2^-4 relative in the first case)."""
    g = torch.Generator().manual_seed(3)
    b, heads, d = 1, 1, 72
    q = rb(torch.randn(b, t, heads, d, generator=g))
    k = rb(torch.randn(b, t, heads, d, generator=g))
    v = rb(torch.randn(b, t, heads, d, generator=g))
    k[0, t - 6, 0] = q[0, 7, 0] * gain
    k = rb(k)
    scale = d ** -0.5
    qd64, kd64, vd64 = (x_.transpose(1, 2).double() for x_ in (q, k, v))
    ref = (torch.softmax(qd64 @ kd64.transpose(-1, -2) * scale, dim=-1) @ vd64).transpose(1, 2).float()
    o = torch.empty(b, t, heads, d, dtype=torch.int16, device="cuda")
    ws = torch.zeros(16 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.ir_op_attention_fp8(ctx.h, ctx.stream(), P(dev_bf16(q)), P(dev_bf16(k)), P(dev_bf16(v)), P(o), b, heads, t, scale, P(ws), ws.numel()),
              "attention_fp8")
    torch.cuda.synchronize()
    got = L.from_bf16_bits(o).cpu()
    tiles = heads * (t // 64) * 10240
    flag = int(ws[((tiles + 255) & ~255) + ((heads * 96 * (t + 64) * 2 + 255) & ~255):][:4].view(torch.int32)[0])   # layout of ir_op_attention_fp8: tiles | V^T [heads][96][t + 64] | flag
    assert (flag != 0) == (gain > 8), f"overflow flag {flag} at gain {gain}"
    spike_err = float((got[0, 7, 0] - v[0, t - 6, 0]).abs().max())
    assert spike_err < (2 ** -6 if flag else 2 ** -3), f"the spiked query must return the spiked key's value row (max error {spike_err:.4f})"
    r = float((got - ref).norm() / ref.norm())
    print(f"fp8 attention spike gain {gain}: flag {flag}, rel-L2 {r:.4f}, spiked row error {spike_err:.4f}")
    assert r <= (0.02 if flag else 0.12)


@pytest.mark.parametrize("t,gains", [(512, (2.0,)), (512, (4.0,)), (2048, (8.0,)), (2048, (1.0, 2.5, 6.0)), (4096, (0.75, 1.5, 2.25, 3.0, 3.75, 4.5))])
def test_flash_attention_d512_spike(ctx, t, gains):
    """d = 512 kernel with the fixed softmax reference (attn_d512.hip): late keys dominate a query. Round 5: the reference MOVES IN PLACE (exact
    power-of-two rescaling of the 256 accumulators) once a score lies 2^48 above it, so none of these cases raises the overflow flag any more
    (before: gain 4 = 2^130 above the first tile's maximum -> flag -> the rescaling kernel recomputed the whole launch).
    gain 2: 2^65 above (one move); gain 4: 2^130 above, i.e. the stream's own probabilities of that tile overflow to inf and are recomputed from the
    scores; gain 8: 2^260; staircases of growing spikes in different tiles: one move per step, for three queries of different waves at once, one
    of them in the LAST tile (whose scores are made by the last full stream)."""
    g = torch.Generator().manual_seed(4)
    d = 512
    q = rb(torch.randn(1, t, 1, d, generator=g))
    k = rb(torch.randn(1, t, 1, d, generator=g))
    v = rb(torch.randn(1, t, 1, d, generator=g))
    rows = (7, 100, t - 1)                                                     # queries of three waves / workgroups
    last = {}
    for qi in rows:
        for i, gn in enumerate(gains):
            key = (t - 6 - 37 * (len(gains) - 1 - i) - 3 * rows.index(qi)) if len(gains) > 1 else t - 6 - 3 * rows.index(qi)   # growing spikes at increasing key positions
            k[0, key, 0] = q[0, qi, 0] * gn
            last[qi] = key
    k = rb(k)
    scale = d ** -0.5
    # Reference = the kernel's NUMBER SCHEME in fp64 (the long spike keys make every other query's row moderately peaky, where the scheme's own
    # roundings - q * scale * log2(e) rounded to bf16, probabilities rounded to bf16 for the second product, the denominator summed unrounded -
    # are worth 0.02-0.04 on small outputs; against an exact softmax that would hide a wrong rescaling behind a loose tolerance): a reference move
    # is an exact power of two on every term, so it must not show at all.
    qs = rb(q[0, :, 0] * (scale * 1.44269504088896340736)).double().cuda()
    s2 = qs @ k[0, :, 0].double().cuda().T
    p2 = torch.exp2(s2 - s2.max(-1, keepdim=True).values)
    ref = ((rb(p2.float().cpu()).double().cuda() @ v[0, :, 0].double().cuda()) / p2.sum(-1, keepdim=True)).float().cpu()[None, :, None]
    o = torch.empty(1, t, 1, d, dtype=torch.int16, device="cuda")
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), P(dev_bf16(q)), P(dev_bf16(k)), P(dev_bf16(v)), P(o), 1, 1, t, t, d,
                                      scale, None, P(ws), ws.numel()), "attention")
    got = L.from_bf16_bits(o).cpu()
    tkp = ((t + 63) & ~63) + 64
    flag = int(ws[((512 * tkp * 2 + 255) & ~255) + ((t * 512 * 2 + 255) & ~255):][:4].view(torch.int32)[0])   # layout of ir_op_attention's d = 512 form: old V^T | V^T tiles | flag
    assert flag == 0, "the in-place reference move must keep the fixed-reference kernel from flagging"
    if max(gains) >= 2.0:
        for qi in rows:
            assert (got[0, qi, 0] - v[0, last[qi], 0]).abs().max() <= 2 ** -6, f"query {qi} must return the value row of its largest spike (to one bf16 ulp)"
    close(got, ref, 2 ** -6, 6e-3, "d512 attention spike")


@pytest.mark.parametrize("h,w,shift", [(8, 8, 0), (16, 24, 0), (16, 24, 4), (64, 64, 4)])
def test_swin_window_attention(ctx, h, w, shift):
    heads, hd, ws_ = 6, 30, 8
    g = torch.Generator().manual_seed(h * w + shift)
    B = 2
    qkv = rb(torch.randn(B, h * w, 3, heads, hd, generator=g))
    table = torch.randn(225, heads, generator=g) * 0.5
    scale = hd ** -0.5
    # reference: roll, partition, attention with relative-position bias + shift mask, reverse (swinir.py:44-73,125-156,227-283)
    coords = torch.stack(torch.meshgrid(torch.arange(8), torch.arange(8), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0) + 7
    idx = rel[:, :, 0] * 15 + rel[:, :, 1]
    bias = table[idx.view(-1)].view(64, 64, heads).permute(2, 0, 1)  # heads, q, k
    x = qkv.view(B, h, w, 3 * heads * hd)
    if shift:
        x = torch.roll(x, (-shift, -shift), (1, 2))
    xw = x.view(B, h // 8, 8, w // 8, 8, -1).permute(0, 1, 3, 2, 4, 5).reshape(-1, 64, 3, heads, hd)
    qq, kk, vv = [xw[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
    attn = (qq * scale) @ kk.transpose(-2, -1) + bias[None]
    if shift:
        img = torch.zeros(1, h, w, 1)
        cnt = 0
        for hs in (slice(0, -8), slice(-8, -shift), slice(-shift, None)):
            for wsl in (slice(0, -8), slice(-8, -shift), slice(-shift, None)):
                img[:, hs, wsl, :] = cnt
                cnt += 1
        mw = img.view(1, h // 8, 8, w // 8, 8, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, 64)
        am = mw[:, None, :] - mw[:, :, None]
        am = am.masked_fill(am != 0, -100.0)
        nW = am.shape[0]
        attn = (attn.view(B, nW, heads, 64, 64) + am[None, :, None]).view(-1, heads, 64, 64)
    o = (attn.softmax(-1) @ vv).transpose(1, 2).reshape(-1, 8, 8, heads * hd)
    o = o.view(B, h // 8, w // 8, 8, 8, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, h, w, -1)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    ref = o.reshape(B, h * w, heads, hd)
    # device layout: [B][T][3][heads][32] zero padded
    qp = torch.zeros(B, h * w, 3, heads, 32)
    qp[..., :hd] = qkv
    biasT = (bias.permute(0, 2, 1).contiguous() * math.log2(math.e)).contiguous()  # [heads][key][query]
    out = torch.empty(B, h * w, heads, 32, dtype=torch.int16, device="cuda")
    ctx.check(ctx.lib.ir_op_swin_attention(ctx.h, ctx.stream(), P(dev_bf16(qp)), P(out), P(biasT.cuda()), B, h, w, heads, shift, scale),
              "swin attention")
    got = L.from_bf16_bits(out).cpu()
    assert got[..., hd:].abs().max() == 0
    close(got[..., :hd], ref, 2 ** -6, 6e-3, "swin attention")


@pytest.mark.parametrize("rows,cols", [(300, 4096), (64, 64), (5, 260), (3, 1028)])
def test_softmax_rows(ctx, rows, cols):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(rows, cols, generator=g) * 4
    y = torch.empty(rows, cols, dtype=torch.int16, device="cuda")
    ctx.check(ctx.lib.ir_op_softmax_rows(ctx.h, ctx.stream(), P(x.cuda()), P(y), rows, cols), "softmax")
    close(L.from_bf16_bits(y).cpu(), x.softmax(-1), 2 ** -7, 1e-6, "softmax rows")


@pytest.mark.parametrize("t,gain", [(512, 2.0), (512, 6.0)])
def test_flash_attention_d512_fp8_spike(ctx, t, gain):
    """flash_attn_d512_fp8_kernel with its fixed softmax reference: a late key dominates one query. gain 2: the spike is about 2^65 above the
    first tile's maximum - inside what the per-tile exponent bytes and the fp32 row sum carry (up to ~2^100); gain 6: about 2^195 above ->
    the overflow flag -> V^T is built and the bf16 rescaling kernel recomputes everything. Either way the spiked query returns the spiked
    key's value row (e4m3 rounding of V, 2^-4 relative, in the first case)."""
    g = torch.Generator().manual_seed(4)
    d = 512
    q = rb(torch.randn(1, t, d, generator=g))
    k = rb(torch.randn(1, t, d, generator=g))
    v = rb(torch.randn(1, t, d, generator=g))
    k[0, t - 6] = q[0, 7] * gain
    k = rb(k)
    scale = d ** -0.5
    ref = (torch.softmax(q[0].double() @ k[0].double().t() * scale, dim=-1) @ v[0].double()).float()
    o = torch.empty(1, t, d, dtype=torch.int16, device="cuda")
    tiles = (t // 64) * 66560
    ws = torch.zeros(((tiles + 255) & ~255) + 256 + (t + 64) * 512 * 2, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.ir_op_attention_d512_fp8(ctx.h, ctx.stream(), P(dev_bf16(q)), P(dev_bf16(k)), P(dev_bf16(v)), P(o), 1, t, scale, P(ws), ws.numel()),
              "attention_d512_fp8")
    torch.cuda.synchronize()
    got = L.from_bf16_bits(o).cpu()[0]
    flag = int(ws[(tiles + 255) & ~255:][:4].view(torch.int32)[0])   # layout of ir_op_attention_d512_fp8: tiles | flag | V^T
    assert (flag != 0) == (gain > 4), f"overflow flag {flag} at gain {gain}"
    spike_err = float((got[7] - v[0, t - 6]).abs().max())
    assert spike_err < (2 ** -6 if flag else 2 ** -2), f"the spiked query must return the spiked key's value row (max error {spike_err:.4f})"
    r = float((got - ref).norm() / ref.norm())
    print(f"fp8 d512 attention spike gain {gain}: flag {flag}, rel-L2 {r:.4f}, spiked row error {spike_err:.4f}")
    assert r <= (0.02 if flag else 0.12)


@pytest.mark.parametrize("n,h,w,cin,cout", [(1, 64, 64, 256, 256), (1, 24, 40, 128, 128), (2, 128, 96, 512, 512)])
def test_conv3x3_fp8_up(ctx, n, h, w, cin, cout):
    """The fp8 3x3 conv on the nearest-2x upsampled input (conv_halo_s1_fp8_kernel<1>, or conv_halo_kernel<.., UP, FP8> below 32 patch tiles):
    the VAE decoder's Upsample convs under IR_FLAG_FP8. Against a float64 convolution of the DEQUANTISED, upsampled operands."""
    g = torch.Generator().manual_seed(n + h + w + cin + cout + 1)
    x8 = (torch.randn(n, cin, h, w, generator=g) * 4).clamp(-448, 448).to(torch.float8_e4m3fn)
    w8 = (torch.randn(cout, cin, 3, 3, generator=g) * 64).clamp(-448, 448).to(torch.float8_e4m3fn)
    deq = torch.rand(cout, generator=g) * 1e-3 + 1e-4
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(F.interpolate(x8.double(), scale_factor=2, mode="nearest"), w8.double(), None, padding=1) * deq.double()[None, :, None, None] + bias.double()[None, :, None, None]
    xin = x8.permute(0, 2, 3, 1).contiguous().view(torch.uint8).cuda()
    wp = w8.permute(0, 2, 3, 1).contiguous().view(torch.uint8).cuda()
    out = torch.empty(n, 2 * h, 2 * w, cout, dtype=torch.int16, device="cuda")
    ctx.check(ctx.lib.ir_op_conv_fp8_up(ctx.h, ctx.stream(), P(xin), P(wp), P(deq.cuda()), P((bias / deq).cuda()), P(out), n, h, w, cin, cout), "conv_fp8_up")
    torch.cuda.synchronize()
    close(L.from_bf16_bits(out).cpu().permute(0, 3, 1, 2).double(), ref, 2 ** -7, 2e-3, "conv fp8 up")


@pytest.mark.parametrize("n,h,w,cin,cout,taps,res", [(1, 8, 8, 1280, 1280, 9, True), (1, 16, 16, 2560, 1280, 9, False), (2, 16, 16, 640, 640, 9, True),
                                                    (1, 16, 16, 5120, 1280, 1, True), (1, 32, 32, 1280, 640, 9, False), (1, 64, 64, 320, 320, 9, False)])
def test_conv_splitk(ctx, n, h, w, cin, cout, taps, res):
    """The split-K form of the generic implicit GEMM (igemm.hip: ir_igemm_splitk + splitk_finish_kernel), which the ControlLDM path's small-M
    convs / linears take: against F.conv2d in fp32 on the same bf16 operands, with bias, SiLU and an fp32 residual in the finishing kernel;
    bit-identical run to run (fixed summation order). The last shape has too many tiles and too few k-tiles to split."""
    import ctypes
    g = torch.Generator().manual_seed(cin + cout + h)
    x = rb(torch.randn(n, cin, h, w, generator=g))
    k = 3 if taps == 9 else 1
    wt = rb(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(taps * cin))
    b = torch.randn(cout, generator=g)
    r_ = torch.randn(n, cout, h, w, generator=g) if res else None
    ref = F.silu(F.conv2d(x, wt, b, padding=k // 2))
    if res:
        ref = ref + r_
    xin = dev_bf16(x.permute(0, 2, 3, 1).contiguous())
    wp = dev_bf16(wt.permute(0, 2, 3, 1).reshape(cout, taps * cin).contiguous())
    rd = r_.permute(0, 2, 3, 1).contiguous().cuda() if res else None
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    outs, splits = [], ctypes.c_int(-1)
    for _ in range(2):
        out = torch.empty(n, h, w, cout, dtype=torch.int16, device="cuda")
        ctx.check(ctx.lib.ir_op_conv_splitk(ctx.h, ctx.stream(), P(xin), P(wp), P(b.cuda()), P(out), n, h, w, cin, cout, taps, L.ACT_SILU, P(rd), 1, 0,
                                            P(ws), ws.numel(), ctypes.byref(splits)), "conv_splitk")
        torch.cuda.synchronize()
        outs.append(out)
    assert (splits.value > 1) == (h < 64), f"split count {splits.value}"
    assert torch.equal(outs[0], outs[1]), "split-K must be deterministic"
    # the four-tile ring these small launches take (igemm_kernel's NST = 4) accumulates in the order of the two-tile form: plain-kernel mode runs that
    ctx.check(ctx.lib.ir_set_plain_kernels(ctx.h, 1), "plain on")
    try:
        out2 = torch.empty(n, h, w, cout, dtype=torch.int16, device="cuda")
        ctx.check(ctx.lib.ir_op_conv_splitk(ctx.h, ctx.stream(), P(xin), P(wp), P(b.cuda()), P(out2), n, h, w, cin, cout, taps, L.ACT_SILU, P(rd), 1, 0,
                                            P(ws), ws.numel(), ctypes.byref(splits)), "conv_splitk plain")
        torch.cuda.synchronize()
    finally:
        ctx.check(ctx.lib.ir_set_plain_kernels(ctx.h, 0), "plain off")
    assert torch.equal(outs[0], out2), "ring form and two-tile form must agree bit for bit"
    close(L.from_bf16_bits(outs[0]).cpu().permute(0, 3, 1, 2), ref, 2 ** -7, 4e-3, f"conv split-K x{splits.value}")


# ---------------------------------------------------------------------------------------------------------------------
# csrc/vae_io.hip: the VAE's first / last convolution at full resolution as HBM-bound kernels of their own
@pytest.mark.parametrize("n,h,w", [(1, 64, 64), (2, 72, 192), (1, 256, 320)])
def test_vae_conv_in(ctx, n, h, w):
    """Encoder.conv_in (3 -> 128) straight from fp32 NCHW planes with `x * 2 - 1` folded in, against F.conv2d on the same bf16-rounded
    operands, plus the GroupNorm(32) partial sums of the STORED values per 8 x 64 tile (ragged tiles: 72 rows, 192 / 320 columns)."""
    import ctypes
    g = torch.Generator().manual_seed(h * 7 + w)
    x = torch.rand(n, 3, h, w, generator=g)
    wt = (torch.rand(128, 3, 3, 3, generator=g) - 0.5) * 0.6
    b = (torch.rand(128, generator=g) - 0.5) * 0.2
    wp = torch.zeros(128, 9, 32)
    wp[:, :, :3] = wt.permute(0, 2, 3, 1).reshape(128, 9, 3)
    out = torch.empty(n, h, w, 128, dtype=torch.int16, device="cuda")
    tiles = ctypes.c_int(0)
    per = ((h + 7) // 8) * ((w + 63) // 64)
    part = torch.full((n, per, 2, 32), float("nan"), device="cuda")
    xd = x.cuda().contiguous()
    ctx.check(ctx.lib.ir_op_vae_conv_in(ctx.h, ctx.stream(), L.ptr(xd), P(dev_bf16(wp)), P(b.cuda()), L.ptr(out), L.ptr(part), n, h, w, 2.0, -1.0,
                                        ctypes.byref(tiles)), "vae_conv_in")
    torch.cuda.synchronize()
    assert tiles.value == per
    xin = (xd * 2.0 - 1.0).to(torch.bfloat16).float()
    ref = F.conv2d(xin, rb(wt).cuda(), b.cuda(), padding=1).permute(0, 2, 3, 1)
    got = out.view(torch.bfloat16).float()
    err = (got - ref).abs().max().item()
    print(f"vae_conv_in ({n}, {h}x{w}): max abs err {err:.5f} (|ref| max {ref.abs().max().item():.2f})")
    assert err <= 2 ** -7 * ref.abs().max().item() + 1e-3       # one bf16 rounding of the result
    # statistics of the stored (bf16) values, per tile and group
    gv = got.double().reshape(n, h, w, 32, 4)
    want = torch.zeros(n, per, 2, 32, dtype=torch.float64, device="cuda")
    tx = (w + 63) // 64
    for ty in range((h + 7) // 8):
        for txi in range(tx):
            blk = gv[:, ty * 8:(ty + 1) * 8, txi * 64:(txi + 1) * 64]
            want[:, ty * tx + txi, 0] = blk.sum(dim=(1, 2, 4))
            want[:, ty * tx + txi, 1] = (blk * blk).sum(dim=(1, 2, 4))
    rel = ((part.double() - want).abs() / (want.abs() + 1.0)).max().item()
    assert rel <= 2e-4, rel


@pytest.mark.parametrize("n,h,w", [(1, 64, 64), (2, 72, 96), (1, 256, 320)])
def test_vae_norm_conv_out(ctx, n, h, w):
    """Decoder.norm_out (as finalised scale / shift) + SiLU + conv_out (128 -> 3) in one pass over the tensor, against
    F.conv2d(bf16(silu(x * scale + shift))) - exactly what the stand-alone GroupNorm pass stored for the generic conv to read."""
    g = torch.Generator().manual_seed(h * 5 + w)
    x = rb(torch.randn(n, h, w, 128, generator=g) * 1.5)
    sc = 0.5 + torch.rand(n, 128, generator=g)
    sh = (torch.rand(n, 128, generator=g) - 0.5)
    wt = (torch.rand(3, 128, 3, 3, generator=g) - 0.5) * 0.1
    b = (torch.rand(3, generator=g) - 0.5) * 0.2
    wp = torch.zeros(32, 9, 128)
    wp[:3] = wt.permute(0, 2, 3, 1).reshape(3, 9, 128)
    bp = torch.zeros(32)
    bp[:3] = b
    out = torch.full((n, h, w, 4), float("nan"), device="cuda")
    ctx.check(ctx.lib.ir_op_vae_norm_conv_out(ctx.h, ctx.stream(), P(dev_bf16(x)), P(sc.cuda()), P(sh.cuda()), P(dev_bf16(wp)), P(bp.cuda()),
                                              L.ptr(out), n, h, w), "vae_norm_conv_out")
    torch.cuda.synchronize()
    act = F.silu(x.cuda() * sc.cuda()[:, None, None, :] + sh.cuda()[:, None, None, :]).to(torch.bfloat16).float()
    ref = F.conv2d(act.permute(0, 3, 1, 2), rb(wt).cuda(), b.cuda(), padding=1).permute(0, 2, 3, 1)
    err = (out[..., :3] - ref).abs().max().item()
    print(f"vae_norm_conv_out ({n}, {h}x{w}): max abs err {err:.5f} (|ref| max {ref.abs().max().item():.2f})")
    # the kernel's SiLU is x * rcp(1 + exp2(-x log2 e)) (1 ulp each): an activation may round to the neighbouring bf16 value
    assert err <= 4e-3 * ref.abs().max().item() + 2e-3 and float(out[..., 3].abs().max()) == 0.0


@pytest.mark.parametrize("n,h,w,cin,cout", [(1, 64, 64, 512, 512), (2, 100, 72, 256, 256), (1, 256, 256, 256, 128),
                                             (1, 64, 64, 64, 64), (2, 50, 39, 64, 64), (1, 24, 40, 128, 64), (1, 8, 8, 128, 128), (1, 16, 24, 128, 128),
                                             (2, 8, 16, 256, 128)])   # 64-channel tiles: conv_halo_kernel<.., PH> (SwinIR's upsampler)
def test_conv_up2x2_phase_form(ctx, n, h, w, cin, cout):
    """nearest-2x upsample + 3x3 conv (the VAE decoder's Upsample, model.py:63-67) as four 2x2 convs on the low-resolution tensor
    (conv_halo_s1_kernel<0, 4>, weights.pack_conv_up2x2): (1) against the same decomposition in PyTorch on the bf16-rounded phase weights
    (tight: only accumulation order and the bf16 output rounding differ), (2) against F.conv2d(F.interpolate(x), W) with the fp32 weights -
    the identity the decomposition rests on (the phase weights are sums of up to four taps rounded to bf16 once). Ragged patches (100 x 72)."""
    from instarevive_amd.weights import pack_conv_up2x2
    g = torch.Generator().manual_seed(h * w + cin)
    x = rb(torch.randn(n, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    wup = pack_conv_up2x2(wt)                                   # int16 bits [4 * cout][4 * cin]
    out = torch.empty(n, 2 * h, 2 * w, cout, dtype=torch.int16, device="cuda")
    xin = dev_bf16(x.permute(0, 2, 3, 1).contiguous())
    ctx.check(ctx.lib.ir_op_conv_up2x2(ctx.h, ctx.stream(), P(xin), P(wup.cuda()), P(b.cuda()), P(out), n, h, w, cin, cout), "conv_up2x2")
    torch.cuda.synchronize()
    got = L.from_bf16_bits(out).cpu().permute(0, 3, 1, 2)       # [n, cout, 2h, 2w]
    # (1) the four phase convs in PyTorch: phase (dy, dx) is a 2x2 conv whose window starts at (y - 1 + dy, x - 1 + dx)
    wph = L.from_bf16_bits(wup).reshape(4, cout, 2, 2, cin).permute(0, 1, 4, 2, 3).contiguous()   # [phase][cout][cin][sy][sx]
    ref = torch.empty(n, cout, 2 * h, 2 * w)
    for dy in (0, 1):
        for dx in (0, 1):
            xp = F.pad(x, (1 - dx, dx, 1 - dy, dy))             # left / right / top / bottom zeros so that a "valid" 2x2 conv gives h x w outputs
            ref[:, :, dy::2, dx::2] = F.conv2d(xp, wph[2 * dy + dx], b)
    err = (got - ref).abs()
    assert not (err > 3e-3 + 2 ** -7 * ref.abs()).any(), f"vs phase convs: max abs err {float(err.max()):.4g}"
    # (2) the definition
    full = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, b, padding=1)
    e2 = (got - full).abs()
    rel = float((got - full).norm() / full.norm())
    print(f"conv_up2x2 ({n}, {h}x{w}, {cin}->{cout}): max abs err vs phase convs {float(err.max()):.4f}, vs upsample + 3x3 (fp32 weights) rel L2 {rel:.5f}")
    assert rel <= 4e-3 and float(e2.max()) <= 0.05 * float(full.abs().max())
