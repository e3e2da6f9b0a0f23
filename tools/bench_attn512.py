"""A/B of the two d = 512 attention kernels through the C ABI (fast = attn_d512.hip, plain = flash_attn_d512_kernel<2>).
   python tools/bench_attn512.py [T ...]"""
import sys
sys.path.insert(0, ".")
import torch
from instarevive_amd import _lib as L
from instarevive_amd.models import get_context

ctx = get_context()


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def attn512(t, b=1):
    q = torch.randn(b, t, 1, 512, device="cuda").to(torch.bfloat16).view(torch.int16)
    k = torch.randn(b, t, 1, 512, device="cuda").to(torch.bfloat16).view(torch.int16)
    v = torch.randn(b, t, 1, 512, device="cuda").to(torch.bfloat16).view(torch.int16)
    o = torch.empty_like(q)
    ws = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
    fn = lambda: ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), b, 1, t, t, 512, 512 ** -0.5, None, L.ptr(ws), ws.numel()), "attn")
    res = {}
    for tag, plain in (("fast", 0), ("plain", 1), ("fast", 0), ("plain", 1)):
        ctx.lib.ir_set_plain_kernels(ctx.h, plain)
        res.setdefault(tag, []).append(timeit(fn))
    ctx.lib.ir_set_plain_kernels(ctx.h, 0)
    fl = 4.0 * b * t * t * 512
    print(f"attn512 b{b} T{t}: fast {min(res['fast']):8.3f} ms ({fl / min(res['fast']) / 1e9:7.1f} TFLOP/s)   plain {min(res['plain']):8.3f} ms "
          f"({fl / min(res['plain']) / 1e9:7.1f} TFLOP/s)   (incl. V transpose)", flush=True)


for t in [int(a) for a in sys.argv[1:]] or [4096, 16384, 65536]:
    attn512(t)
attn512(4096, 25)
