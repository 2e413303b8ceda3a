#!/bin/bash
# round-6 GPU call 21: packed K order x[t] first (all f16 kernels) + the short-input protocol: the whole -m gpu suite, then A/B against HEAD
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_q; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; grep -n "passed\|failed\|error" $O/pytest.log | tail -5
tools/ab.sh 3 HEAD0 BASE | tee $O/ab_c3.txt
BENCH_ARGS="--length 16000" ; for k in 1 2 3; do for v in HEAD0 BASE; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact --length 16000 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 16k', round(d['ms_per_step'],4))"; done; done | tee $O/ab_16k.txt
for k in 1 2; do for v in HEAD0 BASE; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact --case bench/c5 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c5', round(d['ms_per_step'],4))"; done; done | tee $O/ab_c5.txt
for k in 1 2; do for v in HEAD0 BASE; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact --case bench/c4 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c4', round(d['ms_per_step'],4))"; done; done | tee $O/ab_c4.txt
