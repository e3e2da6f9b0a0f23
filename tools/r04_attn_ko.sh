#!/bin/bash
# knock-out timings of flash_attn_pp2_kernel (results wrong by design): python tools/build_variant.py attn_d512.hip tools/libir_pp2koN.so -DIR_KO_PP2=N first
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04; mkdir -p $O
echo "base" > $O/attn_ko.txt
python tools/bench_attn72.py 16384 >> $O/attn_ko.txt 2>&1 || exit 1
for k in 1 2 3 4 5; do
  echo "IR_KO_PP2=$k (1 no wait+barrier, 2 no LDS-DMA, 3 no exp, 4 no fragment reads, 5 no MFMA)" >> $O/attn_ko.txt
  INSTAREVIVE_HIP_LIB=$PWD/tools/libir_pp2ko$k.so timeout -k 5 120 python tools/bench_attn72.py 16384 >> $O/attn_ko.txt 2>&1 || exit 1
done
cat $O/attn_ko.txt | grep -v amdgpu.ids
