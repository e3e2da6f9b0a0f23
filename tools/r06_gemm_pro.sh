#!/bin/bash
# Round 6: gemm_pp_kernel's prologue that starts on the first tile pair (in-tree, IR_GPP_PRO = 1) against the all-at-once prologue (tools/libir_prev.so = -DIR_GPP_PRO=0),
# alternating on ONE box: correctness first (tests), then the DiT's GEMM shapes and the whole step.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_pro; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "linear or gemm or igemm" > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -1 $O/tests.log
ops() { if [ -z "$2" ]; then IR_BENCH_ITERS=20 IR_BENCH_REPS=3 timeout -k 10 200 python tools/bench_ops.py linear 2>&1 | grep "^linear 16384" | sed "s/^/$1 /";
  else INSTAREVIVE_HIP_LIB=$PWD/$2 IR_BENCH_ITERS=20 IR_BENCH_REPS=3 timeout -k 10 200 python tools/bench_ops.py linear 2>&1 | grep "^linear 16384" | sed "s/^/$1 /"; fi; }
step() { if [ -z "$2" ]; then timeout -k 10 400 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_host_rate --cli_files 0 2>&1 | grep "timed loop [0-9]\|gemm_pp_kernel  \|verify" | sed "s/^/$1 /";
  else INSTAREVIVE_HIP_LIB=$PWD/$2 timeout -k 10 400 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_host_rate --cli_files 0 2>&1 | grep "timed loop [0-9]\|gemm_pp_kernel  \|verify" | sed "s/^/$1 /"; fi; }
{
for rep in 1 2; do ops new ""; ops old tools/libir_prev.so; done
for rep in 1 2; do step new ""; step old tools/libir_prev.so; done
} > $O/gemm_pro.txt 2>&1
cat $O/gemm_pro.txt
