#!/bin/bash
# Round 6: request schedule of the residual vectors in gemm_pp_kernel's fp32-residual form (library: 9 before the slab write + 9 behind; variants: round 5's
# 3 x 6, 6 + 12, 12 + 6) on the DiT's residual GEMM shapes, alternating on one box
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_gemm; mkdir -p $O
export IR_BENCH_ITERS=40 IR_BENCH_REPS=3
run() { if [ -z "$2" ]; then timeout -k 10 200 python tools/bench_ops.py linear 2>&1 | grep "^linear 16384.*res=2" | sed "s/^/$1 /"; else INSTAREVIVE_HIP_LIB=$PWD/$2 timeout -k 10 200 python tools/bench_ops.py linear 2>&1 | grep "^linear 16384.*res=2" | sed "s/^/$1 /"; fi; }
{ for rep in 1 2; do run na9 ""; run k10 tools/libir_k10.so; run na6 tools/libir_na6.so; run na12 tools/libir_na12.so; done; } | tee $O/gemm_k1.txt
