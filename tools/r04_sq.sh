#!/bin/bash
# SQ counters of one bench pass (separate --pmc run, no trace domains) + the fp8 line with its parity_512 block
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
python bench.py --fp8 --lq 1024 --sr_scale 2 --cpu_small > $O/bench_fp8.json 2> $O/bench_fp8.err || exit 1
echo "[1] fp8 line done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -o sq -- python3 bench.py --steps 1 --warmup 0 --no_profile --no_cpu_baseline --no_verify --no_host_rate > $O/pmc_sq.log 2>&1 || exit 1
python tools/pmc_sum.py $O/pmc_sq > $O/pmc_sq_summary.txt 2>&1
rm -rf $O/pmc_sq
grep -E "conv_halo_s1_kernel|flash_attn_pp2|flash_attn_d512_v2|gemm_pp" $O/pmc_sq_summary.txt | grep -E "MFMA_BUSY|BUSY_CU" 
python bench.py --fp8 --fp8_parts attention --lq 1024 --sr_scale 2 --cpu_small > $O/bench_fp8_attention.json 2> $O/bench_fp8_attention.err || exit 1
python bench.py --fp8 --fp8_parts no_encoder_convs --lq 1024 --sr_scale 2 --cpu_small > $O/bench_fp8_no_encoder_convs.json 2> $O/bench_fp8_no_encoder_convs.err || exit 1
echo "[3] fp8 operand-subset lines done"
