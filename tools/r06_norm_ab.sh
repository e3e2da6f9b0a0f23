#!/bin/bash
# Round 6: conv_halo_s1_kernel<0, 9, NORM> with the norm arithmetic as packed pairs (library) against scalar instructions (tools/libir_nscalar.so), alternating on one box
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_norm; mkdir -p $O
export IR_BENCH_ITERS=20 IR_BENCH_REPS=3
{
for rep in 1 2; do
  timeout -k 10 200 python tools/bench_ops.py convnorm 2>&1 | grep "norm-in" | sed "s/^/packed /"
  INSTAREVIVE_HIP_LIB=$PWD/tools/libir_nscalar.so timeout -k 10 200 python tools/bench_ops.py convnorm 2>&1 | grep "norm-in" | sed "s/^/scalar /"
done
} | tee $O/norm_ab.txt
