#!/bin/bash
# round-5 GPU call 11: FETCH_SIZE / WRITE_SIZE of the persistent launches, product build vs the pair-fused timing probe (the counters next to r05_a's A/B lines)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out
for v in BASE PAIR5; do
  lib=""; [ $v = BASE ] || lib=$R/tools/abl_so/libpwv_$v.so
  for c in FETCH_SIZE WRITE_SIZE; do
    bash tools/pmc.sh r05_s_${v}_$c $c PWV_LIB=$lib
    echo "== $v $c"; grep -A1 -i "stack_persist" gpurun_out/r05_s_${v}_${c}_pmc.txt | head -4
  done
done
