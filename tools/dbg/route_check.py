import sys, os, math
sys.path.insert(0, os.getcwd())
import torch
from instarevive_amd import _lib as L, Context
ctx = Context(0)
n,h,w,c = 1,2048,2048,64
x = torch.randn(n,h,w,c, device="cuda").to(torch.bfloat16).view(torch.int16)
wt = (torch.randn(c, 9*c, device="cuda")/math.sqrt(9*c)).to(torch.bfloat16).view(torch.int16)
b = torch.zeros(c, device="cuda")
out = torch.empty(n,h,w,c, dtype=torch.int16, device="cuda")
def fn(): ctx.check(ctx.lib.ir_op_conv(ctx.h, ctx.stream(), L.ptr(x), L.ptr(wt), L.ptr(b), L.ptr(out), n,h,w,c,c,c,9,1,1,0,L.ACT_LRELU,0.2,None,0,0),"conv")
for _ in range(3): fn()
torch.cuda.synchronize()
ctx.profile_begin()
for _ in range(10): fn()
torch.cuda.synchronize()
for k,v in ctx.profile_end_kernels().items(): print(k, round(v["ms"]/10,4), v["launches"])
