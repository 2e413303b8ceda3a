#!/bin/bash
# round-5 GPU call 9: where does a unit's time go -- per-phase cycle accounting of the general loop (-DPWV_PTRACE build), short and long input
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r05_j; mkdir -p $O
for a in "16000 10 2" "16000 30 2" "160000 10 2"; do
  echo "== rows layers nets: $a"; PWV_LIB=tools/libpwv_ptrace.so timeout 200 python tools/persist_trace.py $a 2>/dev/null | grep -v "ranges"
done > $O/ptrace_phases.txt 2>&1
cat $O/ptrace_phases.txt
