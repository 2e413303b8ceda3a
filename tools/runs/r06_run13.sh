#!/bin/bash
# round-6 GPU call 20: stationary units + quick WAR look on short inputs (bit identity, timeline, A/B)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_p; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_persist.py -m gpu -q -x ) > $O/pytest_persist.log 2>&1; tail -3 $O/pytest_persist.log
PWV_LIB=tools/libpwv_ptrace.so timeout 300 python tools/persist_timeline.py 16000 10 2 60 > $O/timeline_16000_10.txt 2>&1
tail -22 $O/timeline_16000_10.txt
BENCH_ARGS="--length 16000" tools/ab_env.sh 3 "PWV_PERSIST_UNITWORDS=0" "PWV_PERSIST_STATIONARY=0" "PWV_PERSIST_UNITWORDS=1" | tee $O/ab_16k.txt
BENCH_ARGS="--case bench/c1" tools/ab_env.sh 3 "PWV_PERSIST_UNITWORDS=0" "PWV_PERSIST_STATIONARY=0" "PWV_PERSIST_UNITWORDS=1" | tee $O/ab_c1.txt
BENCH_ARGS="" tools/ab_env.sh 2 "PWV_PERSIST_UNITWORDS=0" "PWV_PERSIST_UNITWORDS=1" | tee $O/ab_c3.txt
