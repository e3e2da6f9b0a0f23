#!/bin/bash
# Round 6: L2 prefetch touches of gemm_pp_kernel's A operand (-DIR_GPP_PF=d variants, tools/build_variant.py) against the in-tree library, alternating on ONE box:
# the DiT's GEMM shapes (tools/bench_ops.py linear) and the whole step.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_pf; mkdir -p $O
ops() { # label, lib
  if [ -z "$2" ]; then IR_BENCH_ITERS=20 IR_BENCH_REPS=3 timeout -k 10 200 python tools/bench_ops.py linear 2>&1 | grep "^linear 16384" | sed "s/^/$1 /";
  else INSTAREVIVE_HIP_LIB=$PWD/$2 IR_BENCH_ITERS=20 IR_BENCH_REPS=3 timeout -k 10 200 python tools/bench_ops.py linear 2>&1 | grep "^linear 16384" | sed "s/^/$1 /"; fi
}
step() { # label, lib
  if [ -z "$2" ]; then timeout -k 10 400 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_verify --no_host_rate --cli_files 0 2>&1 | grep "timed loop [0-9]\|gemm_pp_kernel  " | sed "s/^/$1 /";
  else INSTAREVIVE_HIP_LIB=$PWD/$2 timeout -k 10 400 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_verify --no_host_rate --cli_files 0 2>&1 | grep "timed loop [0-9]\|gemm_pp_kernel  " | sed "s/^/$1 /"; fi
}
{
for rep in 1 2; do
  ops base ""
  for v in pf2 pf4 pf8; do [ -f tools/libir_$v.so ] && ops $v tools/libir_$v.so; done
done
for rep in 1 2; do
  step base ""
  for v in pf2 pf4 pf8; do [ -f tools/libir_$v.so ] && step $v tools/libir_$v.so; done
done
} > $O/gemm_pf.txt 2>&1
cat $O/gemm_pf.txt
