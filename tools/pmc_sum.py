#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter values per (kernel, counter): python tools/pmc_sum.py DIR [name-substring]."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
tot, cnt, dur = defaultdict(float), defaultdict(int), defaultdict(float)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        if pat and pat not in name:
            continue
        key = (name[:70], row["Counter_Name"])
        tot[key] += float(row["Counter_Value"])
        cnt[key] += 1
        dur[key] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
for key in sorted(tot):
    print(f"{key[0]:70s} {key[1]:28s} n={cnt[key]:5d} sum={tot[key]:.4e} per_launch={tot[key] / cnt[key]:.4e} avg_us={dur[key] / cnt[key]:.1f}")
