#!/bin/bash
# rocprofv3 kernel trace of one bench step, kept per DISPATCH (not aggregated): gpurun_out/<tag>_seq.txt lists every dispatch of the LAST pipeline pass
# in launch order with its duration, so that a kernel's launches can be told apart by layer.   usage: tools/ktrace_seq.sh <tag> python3 bench.py ...
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
TAG=$1; shift
rm -rf $O/kts_$TAG
rocprofv3 --kernel-trace --output-format csv -d $O/kts_$TAG -o kt -- "$@" > $O/${TAG}_kts.log 2>&1 || { tail -5 $O/${TAG}_kts.log; exit 1; }
F=$(find $O/kts_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$F" "$O/${TAG}_seq.txt" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "")
# the last pass: from the last u8_to_nchw / swin_prep dispatch on
names = [short(r["Kernel_Name"]) for r in rows]
starts = [i for i, n in enumerate(names) if n.startswith(__import__("os").environ.get("KTS_ANCHOR", "vae_conv_in_kernel"))]
i0 = starts[-1] if starts else 0
# back up to the beginning of that image's SwinIR stage (first dispatch after the previous image's last kernel is unknowable: take 700 before)
with open(sys.argv[2], "w") as f:
    t0 = int(rows[i0]["Start_Timestamp"])
    for r, n in zip(rows[i0:], names[i0:]):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        f.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} us  {d:9.1f} us  grid {r.get('Grid_Size', '?'):>9}  {n}\n")
print("wrote", sys.argv[2], len(rows) - i0, "dispatches")
PY
rm -rf $O/kts_$TAG
