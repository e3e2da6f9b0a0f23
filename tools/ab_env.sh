#!/bin/bash
# A/B of one environment knob on ONE box with the in-tree library, alternating: tools/ab_env.sh <out file> <KNOB=value> <command...>
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$1; KNOB=$2; shift 2
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for rep in 1 2; do
  echo "== default" >> "$OUT"
  timeout -k 10 300 "$@" >> "$OUT" 2>&1 || exit 1
  echo "== $KNOB" >> "$OUT"
  env "$KNOB" timeout -k 10 300 "$@" >> "$OUT" 2>&1 || exit 1
done
