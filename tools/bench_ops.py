#!/usr/bin/env python3
"""Per-shape timing of the igemm / attention kernels through the C ABI (development aid; not part of bench.py)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instarevive_amd import _lib as L, Context

ctx = Context(0)


def timeit(fn, iters=5, warm=2):
    """IR_BENCH_ITERS=n: n timed launches back to back (default 5); IR_BENCH_REPS=r: the best of r such windows (default 1) - short ops need both to
    separate a few per cent on a box whose clock wanders."""
    iters = int(os.environ.get("IR_BENCH_ITERS", iters))
    reps = int(os.environ.get("IR_BENCH_REPS", 1))
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            fn()
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / iters
        best = ms if best is None else min(best, ms)
    return best


def conv(n, h, w, cin, cout, up=0, stride=1):
    x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16).view(torch.int16)
    wt = (torch.randn(cout, 9 * cin, device="cuda") / math.sqrt(9 * cin)).to(torch.bfloat16).view(torch.int16)
    b = torch.zeros(cout, device="cuda")
    ho, wo = (h // 2, w // 2) if stride == 2 else ((2 * h, 2 * w) if up else (h, w))
    out = torch.empty(n, ho, wo, cout, dtype=torch.int16, device="cuda")
    fn = lambda: ctx.check(ctx.lib.ir_op_conv(ctx.h, ctx.stream(), L.ptr(x), L.ptr(wt), L.ptr(b), L.ptr(out), n, h, w, cin, cout, cout, 9, stride,
                                              0 if stride == 2 else 1, up, 0, 0.0, None, 0, 0), "conv")
    ms = timeit(fn)
    fl = 2.0 * n * ho * wo * cout * 9 * cin
    print(f"conv {n}x{h}x{w} {cin}->{cout} up={up} s={stride}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")


def conv_up2(n, h, w, c):
    """Upsample conv (nearest 2x + 3x3, c -> c) in its four-phase 2x2 form (conv_halo_s1_kernel<0, 4>); h x w is the LOW-resolution size"""
    x = torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16).view(torch.int16)
    wup = (torch.randn(4 * c, 4 * c, device="cuda") / math.sqrt(4 * c)).to(torch.bfloat16).view(torch.int16)
    b = torch.zeros(c, device="cuda")
    out = torch.empty(n, 2 * h, 2 * w, c, dtype=torch.int16, device="cuda")
    fn = lambda: ctx.check(ctx.lib.ir_op_conv_up2x2(ctx.h, ctx.stream(), L.ptr(x), L.ptr(wup), L.ptr(b), L.ptr(out), n, h, w, c, c), "conv_up2x2")
    ms = timeit(fn)
    fl = 2.0 * n * 4 * h * w * c * 9 * c
    print(f"conv {n}x{h}x{w} {c}->{c} up, four 2x2 phase convs: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s on the 9-tap FLOPs ({fl * 4 / 9 / ms / 1e9:.1f} executed)")


def conv_norm(n, h, w, cin, cout):
    """3x3 conv with the GroupNorm apply + SiLU of its input inside (conv_halo_s1_kernel<0, 9, NORM>) against apply pass + conv"""
    x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16).view(torch.int16)
    wt = (torch.randn(cout, 9 * cin, device="cuda") / math.sqrt(9 * cin)).to(torch.bfloat16).view(torch.int16)
    b = torch.zeros(cout, device="cuda")
    sc, sh = torch.rand(n, cin, device="cuda") + 0.5, torch.randn(n, cin, device="cuda") * 0.3
    out = torch.empty(n, h, w, cout, dtype=torch.int16, device="cuda")
    fn = lambda: ctx.check(ctx.lib.ir_op_conv_norm(ctx.h, ctx.stream(), L.ptr(x), L.ptr(sc), L.ptr(sh), L.ptr(wt), L.ptr(b), None, L.ptr(out), n, h, w, cin, cout), "conv_norm")
    ms = timeit(fn)
    fl = 2.0 * n * h * w * cout * 9 * cin
    print(f"conv+norm-in {n}x{h}x{w} {cin}->{cout}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")


def conv8(n, h, w, cin, cout):
    """3x3 conv on e4m3 operands (conv_halo_s1_fp8_kernel when it takes the shape; IR_NO_CONV_S1_FP8=1 forces conv_halo_kernel<.., FP8>)"""
    x = torch.randint(0, 120, (n, h, w, cin), device="cuda", dtype=torch.uint8)
    wt = torch.randint(0, 120, (cout, 9 * cin), device="cuda", dtype=torch.uint8)
    g, b = torch.full((cout,), 1e-3, device="cuda"), torch.zeros(cout, device="cuda")
    out = torch.empty(n, h, w, cout, dtype=torch.int16, device="cuda")
    fn = lambda: ctx.check(ctx.lib.ir_op_conv_fp8(ctx.h, ctx.stream(), L.ptr(x), L.ptr(wt), L.ptr(g), L.ptr(b), L.ptr(out), n, h, w, cin, cout, None), "conv8")
    ms = timeit(fn)
    fl = 2.0 * n * h * w * cout * 9 * cin
    print(f"conv fp8 {n}x{h}x{w} {cin}->{cout} (route {ctx.lib.ir_op_conv_fp8_route(ctx.h, n, h, w, cin, cout, 0)}): {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")


def linear(m, k, n, out_f32=0, act=0, res=0):
    """res: 0 none, 1 bf16 residual, 2 fp32 residual + gate (the DiT residual stream)"""
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16).view(torch.int16)
    wt = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(torch.bfloat16).view(torch.int16)
    b = torch.zeros(n, device="cuda")
    out = torch.empty(m, n, dtype=torch.float32 if out_f32 else torch.int16, device="cuda")
    gate = torch.ones(n, device="cuda") if res == 2 else None
    r = None if res == 0 else (torch.randn(m, n, device="cuda") if res == 2 else torch.randn(m, n, device="cuda").to(torch.bfloat16).view(torch.int16))
    fn = lambda: ctx.check(ctx.lib.ir_op_linear(ctx.h, ctx.stream(), L.ptr(x), L.ptr(wt), L.ptr(b), L.ptr(out), m, k, n, n, act,
                                                L.ptr(gate) if gate is not None else None, L.ptr(r) if r is not None else None,
                                                1 if res == 2 else 0, out_f32, 1.0), "linear")
    ms = timeit(fn)
    print(f"linear {m}x{k}->{n} f32={out_f32} act={act} res={res}: {ms:8.3f} ms  {2.0 * m * k * n / ms / 1e9:8.1f} TFLOP/s")


def attn(b, heads, t, d):
    q = torch.randn(b, t, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    k = torch.randn(b, t, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    v = torch.randn(b, t, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    o = torch.empty_like(q)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    fn = lambda: ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), b, heads, t, t, d, d ** -0.5, None,
                                                   L.ptr(ws), ws.numel()), "attn")
    ms = timeit(fn)
    print(f"attn b{b} h{heads} T{t} d{d}: {ms:8.3f} ms  {4.0 * b * heads * t * t * d / ms / 1e9:8.1f} TFLOP/s (incl. V transpose)")


def attn8(b, heads, t):
    """DiT self-attention on e4m3 operands (flash_attn_fp8_kernel + its prep kernel)"""
    d = 72
    q = torch.randn(b, t, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    k = torch.randn(b, t, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    v = torch.randn(b, t, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    o = torch.empty_like(q)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    fn = lambda: ctx.check(ctx.lib.ir_op_attention_fp8(ctx.h, ctx.stream(), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), b, heads, t, d ** -0.5, L.ptr(ws), ws.numel()), "attn8")
    ms = timeit(fn)
    print(f"attn fp8 b{b} h{heads} T{t} d{d}: {ms:8.3f} ms  {4.0 * b * heads * t * t * d / ms / 1e9:8.1f} TFLOP/s (incl. K / V quantisation pass)")


def xattn(b, heads, tq, tk, d):
    """cross-attention shape: short key sequence with an additive key bias"""
    q = torch.randn(b, tq, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    k = torch.randn(b, tk, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    v = torch.randn(b, tk, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    kb = torch.zeros(b, tk, device="cuda")
    o = torch.empty_like(q)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    fn = lambda: ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), b, heads, tq, tk, d, d ** -0.5, L.ptr(kb),
                                                   L.ptr(ws), ws.numel()), "attn")
    ms = timeit(fn, iters=20)
    print(f"cross attn b{b} h{heads} Tq{tq} Tk{tk} d{d}: {ms * 1e3:8.1f} us  {4.0 * b * heads * tq * tk * d / ms / 1e9:8.1f} TFLOP/s (incl. V transpose)")


def gn(n, hw, c):
    x = torch.randn(n, hw, c, device="cuda").to(torch.bfloat16).view(torch.int16)
    y = torch.empty_like(x)
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    fn = lambda: ctx.check(ctx.lib.ir_op_groupnorm(ctx.h, ctx.stream(), L.ptr(x), L.ptr(y), L.ptr(g), L.ptr(b), n, hw, c, 32, 1e-6, 1, L.ptr(ws), ws.numel()), "gn")
    ms = timeit(fn)
    print(f"groupnorm+silu {n}x{hw}x{c}: {ms:8.3f} ms  {6.0 * n * hw * c / ms / 1e9:8.2f} TB/s (partial pass 2 B + apply 4 B per element)")


if __name__ == "__main__":
    which = sys.argv[1:] or ["conv", "linear", "attn"]
    if "one" in which:  # the three heaviest shapes of the 2048^2 path (profiling runs)
        conv(1, 2048, 2048, 128, 128)
        linear(16384, 1152, 4608)
        attn(1, 16, 16384, 72)
    if "conv" in which:
        conv(1, 2048, 2048, 128, 128)
        conv(1, 2048, 2048, 256, 256)
        conv(1, 2048, 2048, 256, 128)
        conv(1, 1024, 1024, 256, 256)
        conv(1, 1024, 1024, 512, 512)
        conv(1, 1024, 1024, 256, 256, up=1)
        conv_up2(1, 1024, 1024, 256)
        conv(1, 512, 512, 512, 512, up=1)
        conv_up2(1, 512, 512, 512)
        conv_up2(1, 256, 256, 512)
        conv(1, 512, 512, 512, 512)
        conv(1, 256, 256, 512, 512)
        conv(1, 2048, 2048, 64, 64)
        conv(1, 256, 256, 192, 192)
    if "swinconv" in which:   # SwinIR's 180 (padded 192)-channel convs at the headline's and the ControlLDM path's token grids
        conv(1, 256, 256, 192, 192)
        conv(1, 64, 64, 192, 192)
    if "convnorm" in which:
        for shp in ((1, 2048, 2048, 128, 128), (1, 2048, 2048, 256, 128), (1, 1024, 1024, 256, 256), (1, 1024, 1024, 512, 512), (1, 512, 512, 512, 512)):
            conv(*shp)
            conv_norm(*shp)
            gn(shp[0], shp[1] * shp[2], shp[3])
    if "linear" in which:
        linear(16384, 1152, 3456)
        linear(16384, 1152, 1152)
        linear(16384, 1152, 4608)
        linear(16384, 4608, 1152)
        linear(16384, 1152, 1152, out_f32=1, res=2)
        linear(16384, 4608, 1152, out_f32=1, res=2)
        linear(16384, 1152, 4608, act=2)
        linear(65536, 192, 192, res=1)
        linear(65536, 192, 384, act=1)
        linear(65536, 512, 65536, out_f32=1)
        linear(65536, 65536, 512)
        linear(65536, 192, 576)
        linear(4194304, 128, 128)
    if "xattn" in which:
        xattn(1, 16, 16384, 300, 72)
        xattn(1, 16, 4096, 300, 72)
        xattn(25, 16, 1024, 300, 72)
    if "gn" in which:
        gn(1, 2048 * 2048, 128)
        gn(1, 2048 * 2048, 256)
        gn(1, 1024 * 1024, 256)
        gn(1, 1024 * 1024, 512)
        gn(1, 512 * 512, 512)
    if "conv8" in which:
        conv8(1, 2048, 2048, 128, 128)
        conv8(1, 2048, 2048, 256, 128)
        conv8(1, 1024, 1024, 256, 256)
        conv8(1, 1024, 1024, 512, 512)
        conv8(1, 512, 512, 512, 512)
        conv8(1, 256, 256, 512, 512)
    if "attn8" in which:
        attn8(1, 16, 16384)
        attn8(1, 16, 1024)
        attn(1, 16, 16384, 72)
    if "attn" in which:
        attn(1, 16, 16384, 72)
        attn(1, 16, 1024, 72)
