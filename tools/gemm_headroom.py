"""What do the DiT's GEMM shapes reach through the vendor library on this box? (torch.matmul = hipBLASLt; a yardstick for gemm_pp_kernel,
not a product path: the product GEMMs carry fused epilogues - bias, GELU, gate, fp32 residual - that a library call would split off.)
    python tools/gemm_headroom.py"""
import torch

def bench(m, k, n, reps=30):
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ b.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{m} x {k} -> {n}: {us:8.1f} us  {2.0 * m * k * n / us / 1e6:7.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    for k, n in ((1152, 1152), (1152, 3456), (1152, 4608), (4608, 1152), (4096, 4096), (8192, 8192)):
        bench(16384, k, n)
