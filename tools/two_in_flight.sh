#!/bin/bash
# Estimate of what two images in flight (two processes, one GPU, hardware queue arbitration) would buy over one: combined images/s against a single run
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_host_rate --no_verify --no_profile > $O/tif_single.json 2> $O/tif_single.log || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_host_rate --no_verify --no_profile > $O/tif_a.json 2> $O/tif_a.log &
PA=$!
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_host_rate --no_verify --no_profile > $O/tif_b.json 2> $O/tif_b.log &
PB=$!
wait $PA || exit 1
wait $PB || exit 1
python - <<'PY'
import json
s = json.load(open("gpurun_out/r04/tif_single.json")); a = json.load(open("gpurun_out/r04/tif_a.json")); b = json.load(open("gpurun_out/r04/tif_b.json"))
print(f"single: {s['ms_per_step']:.2f} ms/step = {s['value']:.3f} images/s")
print(f"two processes: {a['ms_per_step']:.2f} and {b['ms_per_step']:.2f} ms/step = {a['value'] + b['value']:.3f} images/s together (if fully overlapped in time)")
PY
