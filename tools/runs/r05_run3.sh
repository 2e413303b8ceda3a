#!/bin/bash
# round-5 GPU call 3: the one-launch prologue (bit-identity, A/B), the whole -m gpu suite, the short-input timeline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r05_c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed\|error" $O/pytest.log | tail -5; grep -n "Error\|assert" $O/pytest.log | head -20
ab() {  # label, env, bench args
  env $2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact $3 2>$O/err.txt < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],4), round(d['value']/1e6,2))"
}
for k in 1 2 3; do
  ab "c3 fused" PWV_FUSE_PROLOGUE=1 ""
  ab "c3 sep  " PWV_FUSE_PROLOGUE=0 ""
  ab "16k fused" PWV_FUSE_PROLOGUE=1 "--length 16000"
  ab "16k sep  " PWV_FUSE_PROLOGUE=0 "--length 16000"
  ab "c1 fused" PWV_FUSE_PROLOGUE=1 "--case bench/c1"
  ab "c1 sep  " PWV_FUSE_PROLOGUE=0 "--case bench/c1"
done > $O/ab_prologue.txt 2>&1
cat $O/ab_prologue.txt; tail -3 $O/err.txt
for k in 1 2; do
  ab "c4 fused" PWV_FUSE_PROLOGUE=1 "--case bench/c4"
  ab "c4 sep  " PWV_FUSE_PROLOGUE=0 "--case bench/c4"
done > $O/ab_prologue_c4.txt 2>&1
cat $O/ab_prologue_c4.txt
# kernel timeline of the short case (rocprofv3 kernel trace of one bench run; the last 40 kernels = the eager event-timing forwards)
BENCH_ARGS="--length 16000" NLAST=60 bash tools/timeline.sh r05_c/short
tail -45 gpurun_out/r05_c/short_timeline.txt
