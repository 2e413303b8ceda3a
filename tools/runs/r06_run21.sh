#!/bin/bash
# round-6 GPU call 31: units per workgroup in the SHORT instantiation (4000 ... 16000 samples), SHORT up to 7 units per workgroup
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z3; mkdir -p $O
for len in 4000 8000 12000 16000; do echo "== length $len"; python tools/min_units_sweep.py --length $len 2>/dev/null | tail -7; done | tee $O/min_units.txt
( time timeout 1200 python -m pytest tests/test_gpu_persist.py -m gpu -q -x ) > $O/pytest_persist.log 2>&1; head -2 $O/pytest_persist.log | tail -1
