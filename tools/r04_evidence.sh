#!/bin/bash
# One gpurun call that produces the round's measurement files under gpurun_out/r04 (copied into profiles/ afterwards).
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/r04_evidence.sh'
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench_final.json 2> $O/bench_final.err || exit 1
echo "[1] bench done"
python bench.py --fp8 --lq 1024 --sr_scale 2 --no_cpu_baseline > $O/bench_fp8.json 2> $O/bench_fp8.err || exit 1
echo "[2] fp8 done"
python bench.py --tiled --net_hw 2176x3840 --no_cpu_baseline --no_host_rate > $O/bench_4k_tiled.json 2> $O/bench_4k_tiled.err || exit 1
python bench.py --tiled --net_hw 2176x3840 --graph --steps 5 --warmup 2 > $O/bench_4k_tiled_graph.json 2> $O/bench_4k_tiled_graph.err || exit 1
python bench.py --tiled --net_hw 2176x3840 --no_profile --steps 5 --warmup 2 > $O/bench_4k_tiled_plain.json 2> $O/bench_4k_tiled_plain.err || exit 1
echo "[3] 4k tiled done"
IR_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 --no_cpu_baseline --no_host_rate > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err || exit 1
echo "[4] n2 rehearsal done"
python bench.py --batch 8 --steps 3 --warmup 1 --no_cpu_baseline --no_host_rate > $O/bench_b8.json 2> $O/bench_b8.err || exit 1
echo "[4b] batch 8 (cfg-4 per-GPU workload) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_verify --no_host_rate > $O/prof.log 2>&1 || exit 1
echo "[5] kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no_profile --no_cpu_baseline --no_verify --no_host_rate > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no_profile --no_cpu_baseline --no_verify --no_host_rate > $O/pmc_write.log 2>&1 || exit 1
python tools/pmc_kernels.py $O/pmc_fetch $O/pmc_write $O/pmc_kernels.json 1 > $O/pmc_kernels.txt 2>&1
echo "[6] pmc done"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
# the big traces are not needed back
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
ls -la $O
