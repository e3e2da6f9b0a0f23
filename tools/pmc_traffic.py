#!/usr/bin/env python3
"""HBM traffic of the conv/GEMM kernels from two rocprofv3 PMC passes -> profiles/rNN_pmc_igemm.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no_cpu_baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no_cpu_baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_igemm.json

FETCH_SIZE / WRITE_SIZE are in KB. On gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM
section; re-checked here on gn_apply_kernel, whose traffic is known exactly), so reads are doubled; WRITE_SIZE is exact.
passes = pipeline passes in the profiled command (steps + warmup).
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def load(root, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return tot, cnt


fetch, fc = load(sys.argv[1], "FETCH_SIZE")
write, wc = load(sys.argv[2], "WRITE_SIZE")
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 1
# optional: launches of the family inside ONE pipeline pass as bench.py counts them (the profiled process also runs the prompt's
# 30 caption / K-V projections once, outside the step; their traffic is < 0.3 % of the total but they would dilute the average)
step_launches = float(sys.argv[5]) if len(sys.argv) > 5 else None
fam = [k for k in fetch if k.startswith("igemm_kernel") or k.startswith("conv_halo") or k.startswith("gemm_pp") or k.startswith("swin_mlp")]
f_kb = sum(fetch[k] for k in fam) / passes
w_kb = sum(write[k] for k in fam) / passes
launches = step_launches or sum(fc[k] for k in fam) / passes
hbm = (2.0 * f_kb + w_kb) * 1024.0
calib = {k: {"fetch_kb": fetch[k] / fc[k], "write_kb": write[k] / wc[k], "launches": fc[k]} for k in fetch if k.startswith("gn_apply")}
out = {
    "workload": "1x2048x2048 untiled", "kernel": "conv_halo_s1_kernel + conv_halo_pp_kernel + conv_halo_kernel + igemm_kernel + gemm_pp_kernel + swin_mlp_kernel", "round": int(re.search(r"r(\d+)_", os.path.basename(sys.argv[3])).group(1)) if re.search(r"r(\d+)_", os.path.basename(sys.argv[3])) else None,
    "fetch_size_kb_per_step": f_kb, "write_size_kb_per_step": w_kb, "launches_per_step": launches,
    "hbm_bytes_per_step": hbm, "hbm_bytes_per_launch": hbm / max(launches, 1),
    "per_kernel_kb": {k: {"fetch_kb_x2": 2 * fetch[k] / passes, "write_kb": write[k] / passes, "launches": fc[k] / passes} for k in sorted(fam)},
    "calibration_gn_apply_per_launch": calib,
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 1 --warmup 0); FETCH_SIZE doubled "
            "(gfx950 reports half the bytes of 16 B/lane reads; checked with tools/fetch_calib.hip for plain loads and for LDS-DMA pieces of "
            "1 KB, 16 rows x 64 B and 8 rows x 128 B: 0.5000 each; WRITE_SIZE exact). The counters sit on the L2 -> fabric side: "
            "a weight matrix is fetched once per XCD (8 L2s), and Infinity-Cache hits are included",
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("fetch_size_kb_per_step", "write_size_kb_per_step", "launches_per_step", "hbm_bytes_per_launch")}))
