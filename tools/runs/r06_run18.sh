#!/bin/bash
# round-6 GPU call 26: finer stamps at the top of a SHORT unit
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_v; mkdir -p $O
PWV_LIB=tools/libpwv_ptrace.so timeout 300 python tools/persist_timeline.py 16000 10 2 60 > $O/timeline_16000_10.txt 2>&1
sed -n 60,120p $O/timeline_16000_10.txt
