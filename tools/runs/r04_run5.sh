#!/bin/bash
# round-4 GPU call: per-unit progress bytes for the left-neighbour look-back -- bitwise tests, then A/B against the previous build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r04_h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_persist.py -m gpu -x -q > $O/pytest_persist.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_persist.log
ab() {  # label, lib, bench args
  PWV_LIB=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact $3 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],4), round(d['value']/1e6,2))"
}
for k in 1 2 3; do
  ab "c3 new" "" "--case bench/c3"; ab "c3 prev" tools/abl_so/libpwv_PREV.so "--case bench/c3"
  ab "c2 new" "" "--case bench/c2"; ab "c2 prev" tools/abl_so/libpwv_PREV.so "--case bench/c2"
  ab "c3@80000 new" "" "--case bench/c3 --length 80000"; ab "c3@80000 prev" tools/abl_so/libpwv_PREV.so "--case bench/c3 --length 80000"
  ab "c3@48000 new" "" "--case bench/c3 --length 48000"; ab "c3@48000 prev" tools/abl_so/libpwv_PREV.so "--case bench/c3 --length 48000"
done > $O/ab_unit_progress.txt 2>&1
cat $O/ab_unit_progress.txt
