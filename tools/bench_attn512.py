import sys; sys.path.insert(0,".")
sys.argv=["x","none"]
exec(open("tools/bench_ops.py").read().split("if __name__")[0])
def attn512(t):
    q = torch.randn(1, t, 1, 512, device="cuda").to(torch.bfloat16).view(torch.int16)
    k = torch.randn(1, t, 1, 512, device="cuda").to(torch.bfloat16).view(torch.int16)
    v = torch.randn(1, t, 1, 512, device="cuda").to(torch.bfloat16).view(torch.int16)
    o = torch.empty_like(q)
    ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    fn = lambda: ctx.check(ctx.lib.ir_op_attention(ctx.h, ctx.stream(), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), 1, 1, t, t, 512, 512 ** -0.5, None, L.ptr(ws), ws.numel()), "attn")
    ms = timeit(fn, iters=3, warm=1)
    print(f"attn512 T{t}: {ms:8.3f} ms  {4.0 * t * t * 512 / ms / 1e9:8.1f} TFLOP/s (incl. V transpose)")
attn512(16384); attn512(65536)
