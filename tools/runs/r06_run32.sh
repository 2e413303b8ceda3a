#!/bin/bash
# round-6 GPU call 44: GEMM1 of the short-input instantiation without the sched_barrier pins (free scheduling for a lone wave)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z15; mkdir -p $O
( PWV_LIB=tools/abl_so/libpwv_NOPIN.so timeout 900 python -m pytest tests/test_gpu_persist.py -m gpu -q -x -k "bit_identical_to_per_layer" ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log | head -1
for k in 1 2 3 4; do for v in BASE NOPIN; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact --length 16000 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 16k', round(d['ms_per_step'],4))"; done; done | tee $O/ab_16k.txt
