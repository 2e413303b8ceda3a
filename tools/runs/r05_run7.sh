#!/bin/bash
# round-5 GPU call 7: does the dirty ring in L2 at the end of a persistent launch cost the 5.8 us between two flows?  (all ring stores write-through)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r05_h; mkdir -p $O
ab() {  # label, lib, bench args
  PWV_LIB=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact $3 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],4), round(d['value']/1e6,2))"
}
for k in 1 2 3; do
  ab "c3 base" "" ""
  ab "c3 wt  " tools/abl_so/libpwv_WT.so ""
  ab "16k base" "" "--length 16000"
  ab "16k wt  " tools/abl_so/libpwv_WT.so "--length 16000"
  ab "c1 base" "" "--case bench/c1"
  ab "c1 wt  " tools/abl_so/libpwv_WT.so "--case bench/c1"
  ab "c4 base" "" "--case bench/c4"
  ab "c4 wt  " tools/abl_so/libpwv_WT.so "--case bench/c4"
done > $O/ab_wt.txt 2>&1
cat $O/ab_wt.txt
BENCH_ARGS="--length 16000" NLAST=30 PWV_LIB=tools/abl_so/libpwv_WT.so bash tools/timeline.sh r05_h/short_wt
tail -14 gpurun_out/r05_h/short_wt_timeline.txt
