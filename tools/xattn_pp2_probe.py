import sys
sys.path.insert(0, ".")
import torch
from instarevive_amd import _lib as L
from instarevive_amd.models import get_context
ctx = get_context()
heads, d, tq = 16, 72, 16384
for tk in (320, 384, 1024):
    q = torch.randn(1, tq, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    k = torch.randn(1, tk, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    v = torch.randn(1, tk, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
    o = torch.empty_like(q)
    ws = torch.empty(heads * 96 * (tk + 64) * 2 + 8192, dtype=torch.uint8, device="cuda")
    fn = lambda: ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), 1, heads, tq, tk, d, d ** -0.5, None, L.ptr(ws), ws.numel()), "attn")
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"pp2-routed attention Tq {tq} Tk {tk}: {e0.elapsed_time(e1)/20*1e3:.1f} us (incl. V transpose + fallback launch)", flush=True)
