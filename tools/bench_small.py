#!/usr/bin/env python3
"""Timing of the small-M convs / linears of the ControlLDM path (ir_op_conv_splitk: generic implicit GEMM, split-K when the heuristic says so) with COLD
weights: every launch of a window uses another copy of the weight matrix (the copies together exceed the 256 MB Infinity Cache), as in the network,
where 2.4 GB of weights pass once per step. Development aid for SURVEY.md section 8(f) N4; knobs: IR_SPLITK_*, IR_IGEMM_RING_MAX, IR_NO_SPLITK."""
import sys, os, math, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instarevive_amd import _lib as L, Context

ctx = Context(0)
SHAPES = [(1, 16, 16, 2560, 1280, 9), (1, 16, 16, 1280, 1280, 9), (1, 8, 8, 2560, 1280, 9), (1, 8, 8, 1280, 1280, 9), (1, 32, 32, 1280, 640, 9), (1, 32, 32, 640, 640, 9),
          (1, 16, 16, 1280, 1280, 1), (1, 16, 16, 1280, 10240, 1), (1, 16, 16, 5120, 1280, 1), (1, 32, 32, 640, 1920, 1), (1, 32, 32, 640, 5120, 1), (1, 32, 32, 2560, 640, 1),
          (1, 64, 64, 320, 960, 1), (1, 64, 64, 320, 2560, 1), (1, 64, 64, 1280, 320, 1), (1, 8, 8, 1280, 1280, 1)]


def run(n, h, w, cin, cout, taps):
    wbytes = cout * taps * cin * 2
    copies = max(2, min(24, (600 << 20) // wbytes))
    x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16).view(torch.int16)
    wts = [(torch.randn(cout, taps * cin, device="cuda") / math.sqrt(taps * cin)).to(torch.bfloat16).view(torch.int16) for _ in range(copies)]
    b = torch.zeros(cout, device="cuda")
    out = torch.empty(n, h, w, cout, dtype=torch.int16, device="cuda")
    ws = torch.empty(96 << 20, dtype=torch.uint8, device="cuda")
    splits = ctypes.c_int(-1)
    def fn(i):
        ctx.check(ctx.lib.ir_op_conv_splitk(ctx.h, ctx.stream(), L.ptr(x), L.ptr(wts[i % copies]), L.ptr(b), L.ptr(out), n, h, w, cin, cout, taps, L.ACT_NONE, None, 1, 0,
                                            L.ptr(ws), ws.numel(), ctypes.byref(splits)), "conv_splitk")
    for i in range(copies):
        fn(i)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for i in range(copies):
            fn(i)
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) / copies * 1e3
        best = us if best is None else min(best, us)
    fl = 2.0 * n * h * w * cout * taps * cin
    print(f"{'conv' if taps == 9 else 'lin '} M={n * h * w:5d} {cin:5d}->{cout:5d} split {splits.value:3d}: {best:7.1f} us  {fl / best / 1e6:7.1f} TFLOP/s  weights {wbytes / best / 1e6:6.2f} TB/s",
          flush=True)


VAE_SHAPES = [(1, 64, 64, 512, 512, 9), (1, 128, 128, 512, 512, 9), (1, 256, 256, 256, 256, 9), (1, 64, 64, 320, 320, 9), (1, 64, 64, 640, 320, 9), (1, 32, 32, 640, 640, 9),
              (1, 64, 64, 512, 512, 1), (1, 64, 64, 512, 1536, 1)]

if __name__ == "__main__":
    for s in (VAE_SHAPES if "vae" in sys.argv[1:] else SHAPES):
        run(*s)
