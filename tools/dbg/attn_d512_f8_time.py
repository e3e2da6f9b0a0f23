"""Time ir_op_attention_d512_fp8 at T = 65536 (the headline call), for A/B of library variants (INSTAREVIVE_HIP_LIB=...)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from instarevive_amd import Context
from instarevive_amd import _lib as L
ctx = Context(0)
t, d = 65536, 512
g = torch.Generator(device="cuda").manual_seed(1)
q, k, v = (torch.randn(1, t, d, generator=g, device="cuda").to(torch.bfloat16) for _ in range(3))
o = torch.empty(1, t, d, dtype=torch.int16, device="cuda")
ws = torch.zeros((t // 64) * 66560 + 4096 + (t + 64) * 512 * 2, dtype=torch.uint8, device="cuda")
fn = lambda: ctx.check(ctx.lib.ir_op_attention_d512_fp8(ctx.h, ctx.stream(), L.ptr(q.view(torch.int16)), L.ptr(k.view(torch.int16)), L.ptr(v.view(torch.int16)), L.ptr(o), 1, t, d ** -0.5, L.ptr(ws), ws.numel()), "x")
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("INSTAREVIVE_HIP_LIB", "in-tree"), f"{e0.elapsed_time(e1) / 10:.3f} ms per call (prep + kernel + fallback stubs)")
