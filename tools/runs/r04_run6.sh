#!/bin/bash
# round-4 GPU call: software-pipelined fp16-mode residual layer kernel -- its tests, then A/B against the previous build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r04_j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_f16.py -m gpu -x -q > $O/pytest_f16.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_f16.log
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c5" > $O/pytest_fullsize.log 2>&1; echo "fullsize rc $?"; tail -2 $O/pytest_fullsize.log
ab() {  # label, lib, bench args
  PWV_LIB=$2 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-f32-exact $3 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],4), round(d['value']/1e6,2))"
}
for k in 1 2 3; do
  ab "c5f16 new" "" "--case bench/c5 --precision f16"; ab "c5f16 prev" tools/abl_so/libpwv_PREV.so "--case bench/c5 --precision f16"
  ab "c3f16 new" "" "--case bench/c3 --precision f16"; ab "c3f16 prev" tools/abl_so/libpwv_PREV.so "--case bench/c3 --precision f16"
  ab "c4f16 new" "" "--case bench/c4 --precision f16"; ab "c4f16 prev" tools/abl_so/libpwv_PREV.so "--case bench/c4 --precision f16"
done > $O/ab_h16_pipe.txt 2>&1
cat $O/ab_h16_pipe.txt
