#!/bin/bash
# round-6 GPU call 50: the operand split with v_fma_mix_f32 alone (gates scalar as before) against the build before: C3, C4, C5, 16k
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z20; mkdir -p $O
tools/ab.sh 6 PREV BASE | tee $O/ab_c3.txt
for c in "c4:--case bench/c4" "c5:--case bench/c5" "16k:--length 16000"; do n=${c%%:*}; a=${c#*:}; for k in 1 2 3; do for v in PREV BASE; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""; PWV_LIB=$lib python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-f32-exact $a 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $n', round(d['ms_per_step'],4))"; done; done; done | tee $O/ab_rest.txt
