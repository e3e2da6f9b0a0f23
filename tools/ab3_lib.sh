#!/bin/bash
# A/B/C of library builds on ONE box, alternating: tools/ab3_lib.sh <out file> <lib1> <lib2> [<lib3> ...] -- <command...>
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$1; shift
LIBS=()
while [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
shift
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for rep in 1 2; do
  for L in "${LIBS[@]}"; do
    echo "== $L" >> "$OUT"
    INSTAREVIVE_HIP_LIB=$PWD/$L timeout -k 10 300 "$@" >> "$OUT" 2>&1 || exit 1
  done
done
