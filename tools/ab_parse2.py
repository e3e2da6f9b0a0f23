"""Summarise a tools/ab_lib.sh / ab_env.sh file of bench.py lines: ms per step per arm and the per-kernel rows whose names match the given substrings.
usage: python tools/ab_parse2.py <file> [substr ...]"""
import sys, json
tag = None
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("=="):
        tag = line
    elif line.startswith("{"):
        d = json.loads(line)
        pk = (d.get("roofline") or {}).get("per_kernel") or {}
        rows = []
        for sub in sys.argv[2:]:
            for k, v in pk.items():
                if sub in k:
                    rows.append(f"{k.split('/')[-1][:28]}={v['ms_per_step']:.2f}")
        print(tag, "ms", d["ms_per_step"], "clock", d.get("clock_mhz"), "parity2048", (d.get("parity_2048") or {}).get("psnr_db"), "verified", d.get("verified"), " ".join(rows))
