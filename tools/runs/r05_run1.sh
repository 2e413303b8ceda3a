#!/bin/bash
# round-5 GPU call 1: (a) layer-stationary hand-over probe, (b) the PAIR-FUSED schedule priced with a timing-only
# build (tools/probes/pair_probe.patch: what it would not load / not store AND the 32 selects + 32 ds_bpermute it would add),
# variants alternating on one box, (c) the -m gpu suite with the new repairs-are-failures fixture
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r05_a; mkdir -p $O
timeout 120 tools/probes/stage_handover > $O/stage_handover.txt 2>&1; echo "handover rc $?"; cat $O/stage_handover.txt
ab() {  # label, lib, env, bench args
  PWV_LIB=$2 env $3 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact $4 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],4), round(d['value']/1e6,2))"
}
for k in 1 2 3; do
  for v in BASE PAIR5 PAIR5ST PAIR5LD PAIR3; do
    lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""
    ab "c3 $v" "$lib" PWV_X=0 ""
  done
done > $O/ab_pair_c3.txt 2>&1
cat $O/ab_pair_c3.txt
for k in 1 2; do
  for v in BASE PAIR5; do
    lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""
    ab "c4 $v" "$lib" PWV_X=0 "--case bench/c4"
  done
done > $O/ab_pair_c4.txt 2>&1
cat $O/ab_pair_c4.txt
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
