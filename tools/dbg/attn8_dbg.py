import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from instarevive_amd import _lib as L, Context
from tests.support.fp8_tiles import decode_tiles, quantise_q
ctx = Context(0)
torch.manual_seed(0)
def run(q, k, v, label):
    b, t, heads, d = q.shape
    o = torch.zeros(b, t, heads, d, dtype=torch.int16, device="cuda")
    tiles = heads * (t // 64) * 10240
    ws = torch.zeros(tiles + heads * 96 * (t + 128) * 2 + 8192, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.ir_op_attention_fp8(ctx.h, ctx.stream(), L.ptr(q.view(torch.int16)), L.ptr(k.view(torch.int16)), L.ptr(v.view(torch.int16)),
                                          L.ptr(o), b, heads, t, d ** -0.5, L.ptr(ws), ws.numel()), "attention_fp8")
    torch.cuda.synchronize()
    got = o.view(torch.bfloat16).float()
    ref = torch.softmax((q.double().permute(0, 2, 1, 3) @ k.double().permute(0, 2, 3, 1)) * d ** -0.5, -1) @ v.double().permute(0, 2, 1, 3)
    ref = ref.permute(0, 2, 1, 3).float()
    vt_bytes = ((b * heads * 96 * (((t + 63) & ~63) + 64) * 2 + 255) & ~255)
    flag = ws[((tiles + 255) & ~255) + vt_bytes:][:4].view(torch.int32)
    print(f"{label}: rel-L2 {float((got - ref).norm() / ref.norm()):.4f} flag {int(flag[0])} got[0,0,0,:6] {got[0,0,0,:6].tolist()} ref {ref[0,0,0,:6].tolist()}")
    print("   got[0,5,0,64:72]", got[0,5,0,64:72].tolist(), "ref", ref[0,5,0,64:72].tolist())
    print("   got[0,40,0,:4]", got[0,40,0,:4].tolist(), "ref", ref[0,40,0,:4].tolist(), " got[0,200,0,:4]", got[0,200,0,:4].tolist(), "ref", ref[0,200,0,:4].tolist())
for t in (256, 1024):
    q = torch.randn(1, t, 1, 72, device="cuda").to(torch.bfloat16)
    k = torch.randn(1, t, 1, 72, device="cuda").to(torch.bfloat16)
    v = torch.randn(1, t, 1, 72, device="cuda").to(torch.bfloat16)
    run(q, k, torch.ones_like(v), f"T={t} V=1")
    run(torch.zeros_like(q), k, v, f"T={t} q=0")
    vv = torch.zeros_like(v); vv[:, :, :, 3] = torch.arange(t, device="cuda").to(torch.bfloat16)[None, :, None] / t
    run(torch.zeros_like(q), k, vv, f"T={t} q=0 v=key index in d=3")
    run(q, k, v, f"T={t} random")
print("---- masks")
t = 256
q = torch.randn(1, t, 1, 72, device="cuda").to(torch.bfloat16) * 3
k = torch.randn(1, t, 1, 72, device="cuda").to(torch.bfloat16)
v = torch.randn(1, t, 1, 72, device="cuda").to(torch.bfloat16)
for lo, hi in ((0, 32), (32, 64), (64, 72), (0, 1), (5, 6), (33, 34)):
    qm = torch.zeros_like(q); qm[..., lo:hi] = q[..., lo:hi]
    run(qm, k, v, f"q only d {lo}:{hi}")
km = torch.zeros_like(k); km[..., 0] = 1.0
qm = torch.zeros_like(q); qm[..., 0] = torch.linspace(-8, 8, t, device="cuda").to(torch.bfloat16)[None, :, None]
run(qm, km, v, "k = e0, q = ramp e0 (uniform softmax expected)")
