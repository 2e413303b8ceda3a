#!/bin/bash
# round-6 GPU call 43: the protocol's stress test at volume (4000 random shapes, half of them under load)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z14; mkdir -p $O
for s in 11 12 13 14; do timeout 1200 python tools/persist_fuzz.py --cases 500 --seed $s 2>&1 | grep -v amdgpu.ids | tail -3; done | tee $O/fuzz_idle.txt
for s in 21 22 23 24; do timeout 1200 python tools/persist_fuzz.py --cases 500 --seed $s --load 2>&1 | grep -v amdgpu.ids | tail -3; done | tee $O/fuzz_load.txt
