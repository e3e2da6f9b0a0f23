#!/bin/bash
# Sweep of the split-K heuristic of the small-M launches on the ControlLDM path (tools/bench_cldm.py --graph: the whole step as one hipGraph)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_cldm; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" python tools/bench_cldm.py --graph --steps 10 --warmup 3 2> $O/$name.err | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$name /"
}
run base
run kt10_per3 IR_SPLITK_KT=10 IR_SPLITK_PER=3
run kt10_per2 IR_SPLITK_KT=10 IR_SPLITK_PER=2
run kt5_per2 IR_SPLITK_KT=5 IR_SPLITK_PER=2
run kt5_per1 IR_SPLITK_KT=5 IR_SPLITK_PER=1
run kt5_per2_t96 IR_SPLITK_KT=5 IR_SPLITK_PER=2 IR_SPLITK_TILES=96 IR_SPLITK_TARGET=512
run kt10_per3_t96 IR_SPLITK_KT=10 IR_SPLITK_PER=3 IR_SPLITK_TILES=96 IR_SPLITK_TARGET=512
run nosplit IR_NO_SPLITK=1
