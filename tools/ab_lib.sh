#!/bin/bash
# A/B of two builds of the library on ONE box: tools/libir_prev.so against the in-tree one, alternating; usage: tools/ab_lib.sh <out file> <command...>
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for rep in 1 2; do
  echo "== new" >> "$OUT"
  timeout -k 10 300 "$@" >> "$OUT" 2>&1 || exit 1
  echo "== prev" >> "$OUT"
  INSTAREVIVE_HIP_LIB=$PWD/tools/libir_prev.so timeout -k 10 300 "$@" >> "$OUT" 2>&1 || exit 1
done
grep -v amdgpu.ids "$OUT"
