#!/bin/bash
# round-6 GPU call 11: short inputs -- what would it be worth if the units a neighbour workgroup reads were stored PLAIN (valid when that neighbour
# shares the producer's XCD: one L2) instead of write-through?  Timing-only upper bound (tools/probes/plain_probe.patch: ALL ring stores plain while
# a layer fits the L2s) at 1 x 16000 / C1 / 3 x 64000 / C3, and the bitwise suite with that library to see whether the XCD boundaries show.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_i; mkdir -p $O
ab() { PWV_LIB=$2 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-f32-exact $3 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],4), round(d['value']/1e6,2))"; }
for k in 1 2 3; do
  ab "16k base " "" "--length 16000"
  ab "16k plain" tools/abl_so/libpwv_ALLPLAIN.so "--length 16000"
  ab "c1 base " "" "--case bench/c1"
  ab "c1 plain" tools/abl_so/libpwv_ALLPLAIN.so "--case bench/c1"
  ab "3x64k base " "" "--length 64000 --utts 3"
  ab "3x64k plain" tools/abl_so/libpwv_ALLPLAIN.so "--length 64000 --utts 3"
done | tee $O/ab_plain.txt
PWV_LIB=tools/abl_so/libpwv_ALLPLAIN.so timeout 900 python -m pytest tests/test_gpu_persist.py -m gpu -q 2>&1 | tail -4 | tee $O/pytest_plain.txt
