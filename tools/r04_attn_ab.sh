#!/bin/bash
# A/B timing of the attention kernels' builds on ONE box: tools/libir_base.so (round 3) against variants
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04; mkdir -p $O
: > $O/attn_ab.txt
for rep in 1 2; do
for v in base v3 v4 v4b; do
  echo "== $v" >> $O/attn_ab.txt
  INSTAREVIVE_HIP_LIB=$PWD/tools/libir_$v.so timeout -k 5 120 python tools/bench_attn72.py 16384 >> $O/attn_ab.txt 2>&1 || exit 1
  INSTAREVIVE_HIP_LIB=$PWD/tools/libir_$v.so timeout -k 5 120 python tools/bench_attn512.py 65536 >> $O/attn_ab.txt 2>&1 || exit 1
done
done
grep -v amdgpu.ids $O/attn_ab.txt
