#!/bin/bash
# round-6 GPU call 38: protocol stress -- random shapes through the persistent launch against the per-layer launches, idle and under load
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z9; mkdir -p $O
timeout 1500 python tools/persist_fuzz.py --cases 300 --seed 1 > $O/fuzz_idle.txt 2>&1; tail -4 $O/fuzz_idle.txt
timeout 1500 python tools/persist_fuzz.py --cases 200 --seed 2 --load > $O/fuzz_load.txt 2>&1; tail -4 $O/fuzz_load.txt
timeout 900 python tools/persist_fuzz.py --cases 100 --seed 3 --max-rows 200000 > $O/fuzz_long.txt 2>&1; tail -3 $O/fuzz_long.txt
