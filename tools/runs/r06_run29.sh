#!/bin/bash
# round-6 GPU call 41: cond_proj_kernel with its staging loop unrolled: parity, then C1 against the build before
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z12; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "prologue or cond_proj or project" ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for k in 1 2 3 4; do for v in PREVC BASE; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-f32-exact --case bench/c1 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c1', round(d['ms_per_step'],4), round(d['value']/1e6,1))"; done; done | tee $O/ab_c1.txt
