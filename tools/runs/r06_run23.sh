#!/bin/bash
# round-6 GPU call 33: unit words on the GENERAL kernel for medium inputs (8 ... 24 units per workgroup)?  bit identity, then lengths 32000 ... 96000
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z5; mkdir -p $O
( time PWV_LIB=tools/abl_so/libpwv_MED24.so timeout 1200 python -m pytest tests/test_gpu_persist.py -m gpu -q -x ) > $O/pytest_persist_med.log 2>&1; grep -n "passed\|failed" $O/pytest_persist_med.log | tail -2
for len in 32000 40000 48000 64000 80000 96000 128000; do for k in 1 2; do for v in BASE MED24; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact --length $len 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $len', round(d['ms_per_step'],4), round(d['value']/1e6,2))"; done; done; done | tee $O/ab_lengths.txt
