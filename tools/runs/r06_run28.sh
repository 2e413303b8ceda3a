#!/bin/bash
# round-6 GPU call 40: poll granularity of the SHORT wait loop (s_sleep 1 instead of 4)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z11; mkdir -p $O
for k in 1 2 3 4; do for v in BASE SL1; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact --length 16000 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 16k', round(d['ms_per_step'],4))"; done; done | tee $O/ab_16k.txt
for k in 1 2 3; do for v in BASE SL1; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact --case bench/c1 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c1', round(d['ms_per_step'],4), round(d['value']/1e6,1))"; done; done | tee $O/ab_c1.txt
