#!/bin/bash
# Round 6: gemm_pp_kernel's specialised epilogue forms against the library of the previous commit (tools/libir_prev.so, tools/build_prev.sh igemm.hip)
# and three knock-outs of the fp32-residual row phase, on the DiT's GEMM shapes (tools/bench_ops.py linear), alternating on ONE box.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_gemm; mkdir -p $O
run() { # label, lib
  if [ -z "$2" ]; then timeout -k 10 200 python tools/bench_ops.py linear 2>&1 | grep "^linear 16384" | sed "s/^/$1 /";
  else INSTAREVIVE_HIP_LIB=$PWD/$2 timeout -k 10 200 python tools/bench_ops.py linear 2>&1 | grep "^linear 16384" | sed "s/^/$1 /"; fi
}
{
for rep in 1 2; do
  run new ""
  run prev tools/libir_prev.so
done
IR_GEMM_PP_GENERIC=1 run generic ""
for v in gko_r1 gko_r2 gko_r3; do [ -f tools/libir_$v.so ] && run $v tools/libir_$v.so; done
} | tee $O/gemm_ab.txt
