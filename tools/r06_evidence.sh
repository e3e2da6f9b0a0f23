#!/bin/bash
# One gpurun call that produces the round's measurement files under gpurun_out/r06 (copied into profiles/r06_* afterwards).
#   /usr/local/graft/bin/gpurun --timeout 1150 -- 'bash tools/r06_evidence.sh'
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
IR_KEEP_POWER_TRACE=$O/power_trace.txt python bench.py --cli_files 0 > $O/bench_final.json 2> $O/bench_final.err || exit 1
echo "[1] bench done: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_final.json | head -1)"
python bench.py --gpus 1 --steps 20 --warmup 5 --cpu_small > $O/bench_driver_form.json 2> $O/bench_driver_form.err || exit 1
echo "[1b] driver form done: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_driver_form.json | head -1)"
python bench.py --fp8 --lq 1024 --sr_scale 2 --cpu_small > $O/bench_fp8.json 2> $O/bench_fp8.err || exit 1
python bench.py --fp8 --fp8_parts all --lq 1024 --sr_scale 2 --cpu_small > $O/bench_fp8_all.json 2> $O/bench_fp8_all.err || exit 1
echo "[2] fp8 done (calibrated set / every part)"
python bench.py --tiled --net_hw 2176x3840 --no_cpu_baseline --no_host_rate > $O/bench_4k_tiled.json 2> $O/bench_4k_tiled.err || exit 1
python bench.py --tiled --net_hw 2176x3840 --graph --steps 5 --warmup 2 > $O/bench_4k_tiled_graph.json 2> $O/bench_4k_tiled_graph.err || exit 1
echo "[3] 4k tiled done"
python bench.py --batch 8 --steps 3 --warmup 1 --no_cpu_baseline --no_host_rate > $O/bench_b8.json 2> $O/bench_b8.err || exit 1
echo "[4] batch 8 done"
IR_NO_POWER_TRACE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_verify --no_host_rate --cli_files 0 > $O/prof.log 2>&1 || exit 1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
echo "[5] kernel trace done"
IR_NO_POWER_TRACE=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no_profile --no_cpu_baseline --no_verify --no_host_rate --cli_files 0 > $O/pmc_fetch.log 2>&1 || exit 1
IR_NO_POWER_TRACE=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no_profile --no_cpu_baseline --no_verify --no_host_rate --cli_files 0 > $O/pmc_write.log 2>&1 || exit 1
python tools/pmc_kernels.py $O/pmc_fetch $O/pmc_write $O/r06_pmc_kernels.json 1 > $O/pmc_kernels.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write
echo "[6] pmc done"
IR_NO_POWER_TRACE=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -o sq -- python3 bench.py --steps 1 --warmup 0 --no_profile --no_cpu_baseline --no_verify --no_host_rate --cli_files 0 > $O/pmc_sq.log 2>&1 || exit 1
python tools/pmc_sum.py $O/pmc_sq > $O/pmc_sq_summary.txt 2>&1
rm -rf $O/pmc_sq
echo "[7] sq counters done"
python tools/bench_cldm.py --steps 20 --warmup 3 > $O/bench_cldm.log 2>&1 || echo "cldm bench failed"
python tools/bench_cldm.py --steps 40 --warmup 5 --graph > $O/bench_cldm_graph.log 2>&1 || echo "cldm graph bench failed"
echo "[8] cldm done"
ls -la $O
