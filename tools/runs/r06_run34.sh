#!/bin/bash
# round-6 GPU call 46: the operand split with v_fma_mix_f32 (x - float(hi) reading the fp16 halves in place: -60 VALU per unit): parity, then C3 / C4 / 16k against the product
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z17; mkdir -p $O
( PWV_LIB=tools/abl_so/libpwv_MIX.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_persist.py -m gpu -q -x ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log | head -1
tools/ab.sh 4 BASE MIX | tee $O/ab_c3.txt
for k in 1 2; do for v in BASE MIX; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-f32-exact --case bench/c4 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c4', round(d['ms_per_step'],4))"; done; done | tee $O/ab_c4.txt
for k in 1 2; do for v in BASE MIX; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-f32-exact --case bench/c5 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c5', round(d['ms_per_step'],4))"; done; done | tee $O/ab_c5.txt
for k in 1 2 3; do for v in BASE MIX; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact --length 16000 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 16k', round(d['ms_per_step'],4))"; done; done | tee $O/ab_16k.txt
