#!/bin/bash
# round-6 GPU call 34: the P row requested and waited for in asm (the early half for real): bit identity, then 16k / C1 against the product
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z6; mkdir -p $O
( time PWV_LIB=tools/abl_so/libpwv_ASMP.so timeout 1200 python -m pytest tests/test_gpu_persist.py -m gpu -q -x ) > $O/pytest_persist_asmp.log 2>&1; grep -n "passed\|failed" $O/pytest_persist_asmp.log | tail -2
for k in 1 2 3; do for v in BASE ASMP; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact --length 16000 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 16k', round(d['ms_per_step'],4))"; done; done | tee $O/ab_16k.txt
for k in 1 2 3; do for v in BASE ASMP; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact --case bench/c1 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c1', round(d['ms_per_step'],4), round(d['value']/1e6,1))"; done; done | tee $O/ab_c1.txt
for k in 1 2; do for v in BASE ASMP; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact --length 24000 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 24k', round(d['ms_per_step'],4))"; done; done | tee $O/ab_24k.txt
