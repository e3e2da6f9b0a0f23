#!/bin/bash
# knock-outs of gemm_pp_kernel's K loop on the DiT GEMM shapes (results of the variants are wrong by design; timing only):
# base | 1 no MFMAs | 2 no LDS-DMA | 3 no fragment reads | 4 no B (weight) pieces | 5 no A (activation) pieces
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_gko; mkdir -p $O
for v in base gko1 gko2 gko3 gko4 gko5; do
  if [ $v = base ]; then python tools/bench_ops.py linear 2>&1 | grep "^linear 16384" | sed "s/^/$v /"; else INSTAREVIVE_HIP_LIB=$PWD/tools/libir_$v.so python tools/bench_ops.py linear 2>&1 | grep "^linear 16384" | sed "s/^/$v /"; fi
done | tee $O/gemm_ko.txt
