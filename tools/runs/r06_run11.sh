#!/bin/bash
# round-6 GPU call 15: event timeline of a short persistent launch (where the hand-over between two layers goes)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_k; mkdir -p $O
PWV_LIB=tools/libpwv_ptrace.so timeout 300 python tools/persist_timeline.py 16000 10 2 60 > $O/timeline_16000_10.txt 2>&1
PWV_LIB=tools/libpwv_ptrace.so timeout 300 python tools/persist_timeline.py 16000 30 2 60 > $O/timeline_16000_30.txt 2>&1
PWV_LIB=tools/libpwv_ptrace.so timeout 300 python tools/persist_trace.py 16000 10 2 > $O/ptrace_16000_10.txt 2>&1
tail -40 $O/timeline_16000_10.txt
for i in 1 2; do python bench.py --no-cpu-baseline --no-f32-exact --length 16000 --steps 50 --warmup 10 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('16k', d['ms_per_step'], d['value'])"; done
