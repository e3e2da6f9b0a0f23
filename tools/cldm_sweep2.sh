#!/bin/bash
# Round 6: split-K / ring heuristics of the small launches on the ControlLDM path after the tile-order fix (tools/bench_cldm.py, per-launch events)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06b; mkdir -p $O
run() { name=$1; shift
  env "$@" python tools/bench_cldm.py --steps 6 --warmup 2 > $O/cldm_$name.log 2>&1
  echo "$name $(grep -o '"ms_per_step": [0-9.]*' $O/cldm_$name.log) $(grep -h 'igemm_kernel<taps' $O/cldm_$name.log | awk '{printf "%s=%s ", $2, $3}')"
}
run base
run xcd_order IR_NO_TILE_LIN=1 IR_IGEMM_RING_MAX=0
run noring IR_IGEMM_RING_MAX=0
run t256_p12 IR_SPLITK_TARGET=256 IR_SPLITK_PER=12
run t256_p12_noring IR_SPLITK_TARGET=256 IR_SPLITK_PER=12 IR_IGEMM_RING_MAX=0
run t256_p8 IR_SPLITK_TARGET=256 IR_SPLITK_PER=8
run t256_p12_kt12 IR_SPLITK_TARGET=256 IR_SPLITK_PER=12 IR_SPLITK_KT=12
run t256_p12_tiles96 IR_SPLITK_TARGET=256 IR_SPLITK_PER=12 IR_SPLITK_TILES=96
