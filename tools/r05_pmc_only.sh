#!/bin/bash
# re-collect the PMC traffic file after a kernel-source change (roofline.traffic is refused unless its csrc_sha16 equals the tree's)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05p; mkdir -p $O
export TMPDIR=/tmp
IR_NO_POWER_TRACE=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no_profile --no_cpu_baseline --no_verify --no_host_rate --cli_files 0 > $O/pmc_fetch.log 2>&1 || exit 1
IR_NO_POWER_TRACE=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no_profile --no_cpu_baseline --no_verify --no_host_rate --cli_files 0 > $O/pmc_write.log 2>&1 || exit 1
python tools/pmc_kernels.py $O/pmc_fetch $O/pmc_write $O/r05_pmc_kernels.json 1 > $O/pmc_kernels.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write
cp $O/r05_pmc_kernels.json profiles/r05_pmc_kernels.json
python bench.py --gpus 1 --steps 20 --warmup 5 --cpu_small > $O/bench.json 2> $O/bench.err
grep -o '"traffic": [0-9.a-z]*' $O/bench.json | head -1; grep "timed loop" $O/bench.err | cut -c1-200
