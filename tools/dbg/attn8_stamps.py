import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from instarevive_amd import _lib as L, Context
ctx = Context(0)
lib = ctx.lib
b, heads, t, d = 1, 16, 16384, 72
q = torch.randn(b, t, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
k = torch.randn(b, t, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
v = torch.randn(b, t, heads, d, device="cuda").to(torch.bfloat16).view(torch.int16)
o = torch.empty_like(q)
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
fn = lambda: ctx.check(lib.ir_op_attention_fp8(ctx.h, ctx.stream(), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), b, heads, t, d ** -0.5, L.ptr(ws), ws.numel()), "attn8")
for _ in range(3): fn()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.ir_f8_stamps(buf, 1)
n = 5
for _ in range(n): fn()
torch.cuda.synchronize()
lib.ir_f8_stamps(buf, 0)
waves = n * (t // 256) * heads * 4
tiles = t // 64
print("cycles per tile and wave: phase A %.0f, mid %.0f, phase B %.0f" % tuple(x / waves / tiles for x in buf[:3]))
