"""DiT self-attention (16 heads x 72) through the C ABI: time per launch.   python tools/bench_attn72.py [T ...]"""
import sys
sys.path.insert(0, ".")
import torch
from instarevive_amd import _lib as L
from instarevive_amd.models import get_context

ctx = get_context()
for t in [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384]:
    heads, d = 16, 72
    q = (torch.randn(1, t, heads, d, device="cuda")).to(torch.bfloat16).view(torch.int16)
    k = torch.randn(1, t, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    v = torch.randn(1, t, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    o = torch.empty_like(q)
    ws = torch.empty(heads * 96 * (t + 64) * 2 + 8192, dtype=torch.uint8, device="cuda")
    fn = lambda: ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), 1, heads, t, t, d, d ** -0.5, None, L.ptr(ws), ws.numel()), "attn")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"attn 16x72 T{t}: {ms:8.3f} ms  {4.0 * heads * t * t * d / ms / 1e9:8.1f} TFLOP/s useful (incl. V transpose + fallback launch)", flush=True)
