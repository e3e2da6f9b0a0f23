import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from instarevive_amd import _lib as L, Context
ctx = Context(0)
t, heads, d = 256, 1, 72
def run(q, k, v):
    o = torch.zeros(1, t, heads, d, dtype=torch.int16, device="cuda")
    tiles = heads * (t // 64) * 10240
    ws = torch.zeros(tiles + heads * 96 * (t + 128) * 2 + 8192, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.ir_op_attention_fp8(ctx.h, ctx.stream(), L.ptr(q.view(torch.int16)), L.ptr(k.view(torch.int16)), L.ptr(v.view(torch.int16)),
                                          L.ptr(o), 1, heads, t, d ** -0.5, L.ptr(ws), ws.numel()), "attention_fp8")
    torch.cuda.synchronize()
    return o.view(torch.bfloat16).float()
res = []
for dim in (0, 40, 66):
    row = []
    for p0 in range(64):
        k = torch.zeros(1, t, heads, d, device="cuda"); k[0, torch.arange(t) % 64 == p0, 0, dim] = 8.0
        q = torch.zeros(1, t, heads, d, device="cuda"); q[..., dim] = 16.0
        v = torch.zeros(1, t, heads, d, device="cuda"); v[0, :, 0, 3] = (torch.arange(t, device="cuda") % 64) / 64.0
        got = run(q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16))
        row.append(round(float(got[0, 7, 0, 3]) * 64))
    print(f"dim {dim}: peak key p0 -> observed V key:", row)
