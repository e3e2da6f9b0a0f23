#!/bin/bash
# rocprofv3 kernel trace of one command; the per-kernel stats land in gpurun_out/r04/<tag>_stats.csv.  usage: tools/ktrace.sh <tag> python3 <script> [args]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
TAG=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$TAG -o kt -- "$@" > $O/${TAG}.log 2>&1 || { tail -5 $O/${TAG}.log; exit 1; }
find $O/kt_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_stats.csv
rm -rf $O/kt_$TAG
head -${KT_ROWS:-10} $O/${TAG}_stats.csv | cut -c1-170
