#!/bin/bash
# round-4 GPU call 3: fp16-mode persistent launch at 12 waves per workgroup; where the persistent launch stops paying (rows per launch)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r04_c; mkdir -p $O
ab() {  # label, lib, env, bench args
  PWV_LIB=$2 env $3 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-f32-exact $4 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],4), round(d['value']/1e6,2))"
}
for k in 1 2; do
  ab "c5f16 perlayer" "" PWV_PERSIST=0 "--case bench/c5 --precision f16"
  ab "c5f16 persist8" "" PWV_PERSIST=1 "--case bench/c5 --precision f16"
  ab "c5f16 persist12" tools/abl_so/libpwv_H16W12.so PWV_PERSIST=1 "--case bench/c5 --precision f16"
  ab "c3f16 perlayer" "" PWV_PERSIST=0 "--case bench/c3 --precision f16"
  ab "c3f16 persist8" "" PWV_PERSIST=1 "--case bench/c3 --precision f16"
  ab "c3f16 persist12" tools/abl_so/libpwv_H16W12.so PWV_PERSIST=1 "--case bench/c3 --precision f16"
done > $O/ab_f16_waves.txt 2>&1
for k in 1 2; do for len in 320000 480000 640000 960000; do
  ab "c3 $len perlayer" tools/abl_so/libpwv_PREV.so PWV_PERSIST=0 "--case bench/c3 --length $len"
  ab "c3 $len persist" tools/abl_so/libpwv_PREV.so PWV_PERSIST=1 "--case bench/c3 --length $len"
done; done > $O/ab_rows_crossover.txt 2>&1
cat $O/*.txt
