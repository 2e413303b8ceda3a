#!/bin/bash
# round-5 GPU call 8: write-through ring stores when a layer's rows exceed the L2s (size rule inside the launcher) vs the previous build;
# units per workgroup at short inputs; where the persistent launch stops paying now that the tail rides in it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r05_i; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_persist.py -m gpu -q -x 2>&1 | tail -2
ab() {  # label, lib, env, bench args
  PWV_LIB=$2 env $3 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact $4 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],4), round(d['value']/1e6,2))"
}
for k in 1 2 3; do
  ab "c3 rule" "" PWV_X=0 ""
  ab "c3 prev" tools/abl_so/libpwv_PREV5.so PWV_X=0 ""
  ab "c2 rule" "" PWV_X=0 "--case bench/c2"
  ab "c2 prev" tools/abl_so/libpwv_PREV5.so PWV_X=0 "--case bench/c2"
done > $O/ab_wt_rule.txt 2>&1
cat $O/ab_wt_rule.txt
python tools/min_units_sweep.py > $O/min_units.txt 2>/dev/null; cat $O/min_units.txt
for k in 1 2; do for len in 480000 640000 960000; do
  ab "c3 $len perlayer" "" PWV_PERSIST=0 "--case bench/c3 --length $len --steps 10"
  ab "c3 $len persist " "" PWV_PERSIST=1 "--case bench/c3 --length $len --steps 10"
done; done > $O/ab_rows_crossover.txt 2>&1
cat $O/ab_rows_crossover.txt
