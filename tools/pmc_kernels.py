#!/usr/bin/env python3
"""HBM-side traffic of EVERY kernel of one pipeline pass from two rocprofv3 PMC passes -> profiles/rNN_pmc_kernels.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_verify --no_host_rate
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_verify --no_host_rate
    python tools/pmc_kernels.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r03_pmc_kernels.json [passes]

FETCH_SIZE / WRITE_SIZE are in KB. On gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM
section; profiles/r02_fetch_calibration.txt re-checks it for plain loads and LDS-DMA pieces), so reads are doubled; WRITE_SIZE is exact.
bench.py reads `per_kernel[<dominant kernel>].hbm_bytes_per_launch` as roofline.traffic (labelled static).
"""
import csv
import glob
import json
import os
import re
import sys
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


def main():
    fetch, fc = load(sys.argv[1], "FETCH_SIZE")
    write, wc = load(sys.argv[2], "WRITE_SIZE")
    passes = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    per = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
        launches = max(fc.get(k, 0), wc.get(k, 0)) / passes
        if launches <= 0:
            continue
        rd, wr = 2.0 * fetch.get(k, 0.0) * 1024 / passes, write.get(k, 0.0) * 1024 / passes
        per[k] = dict(launches_per_step=launches, read_bytes_per_step=rd, write_bytes_per_step=wr, hbm_bytes_per_launch=(rd + wr) / launches)
    m = re.search(r"r(\d+)_", os.path.basename(sys.argv[3]))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from instarevive_amd.build import source_hash
    # csrc_sha16: the kernel sources these counters were collected on - bench.py refuses the file as roofline.traffic when it differs from the tree's
    out = dict(workload="1x2048x2048 untiled", round=int(m.group(1)) if m else None, csrc_sha16=source_hash(), per_kernel=per,
               hbm_bytes_per_step=sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in per.values()),
               note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 1 --warmup 0); FETCH_SIZE doubled "
                    "(gfx950 reports half the bytes of 16 B/lane reads: profiles/r02_fetch_calibration.txt), WRITE_SIZE exact. The counters sit on the "
                    "L2 -> fabric side: a weight matrix is fetched once per XCD (8 L2s), and Infinity-Cache hits are included")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in list(per.items())[:12]:
        print(f"{k[:70]:70s} {v['launches_per_step']:6.0f} launches  {(v['read_bytes_per_step'] + v['write_bytes_per_step']) / 1e9:8.2f} GB/step")


if __name__ == "__main__":
    main()
