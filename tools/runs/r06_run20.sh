#!/bin/bash
# round-6 GPU call 30: how far does the SHORT instantiation pay?  5 / 6 / 7 units per workgroup (lengths 20000 / 24000 / 28000) against the general kernel
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z2; mkdir -p $O
for len in 8000 12000 16000 20000 24000 28000 32000; do for k in 1 2; do for v in BASE SU7; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact --length $len 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $len', round(d['ms_per_step'],4), round(d['value']/1e6,2))"; done; done; done | tee $O/ab_lengths.txt
