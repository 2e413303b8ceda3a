#!/bin/bash
# round-4 GPU call 2: the fp16-mode persistent launch (bitwise vs per-layer, A/B), the refactored split-fp16 instantiation vs the
# previous build, what a persistent launch is worth at C5's size without the condition, the tests call 1 did not reach
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r04_b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_f16.py tests/test_safe_call.py tests/test_gpu_unfused_and_e2e.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 600 tools/ab.sh 3 BASE PREV > $O/ab_prev.txt 2>&1
BENCH_ARGS="--case bench/c5 --precision f16" timeout 600 tools/ab_env.sh 3 PWV_PERSIST=0 PWV_PERSIST=1 > $O/ab_c5f16_persist.txt 2>&1
BENCH_ARGS="--case bench/c3 --precision f16" timeout 600 tools/ab_env.sh 2 PWV_PERSIST=0 PWV_PERSIST=1 > $O/ab_c3f16_persist.txt 2>&1
BENCH_ARGS="--case bench/c3 --length 960000" timeout 600 tools/ab_env.sh 2 PWV_PERSIST=0 PWV_PERSIST=1 > $O/ab_c3_960k_persist.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c5" > $O/pytest_fullsize_c5.log 2>&1; echo "fullsize rc $?"; tail -3 $O/pytest_fullsize_c5.log
for f in $O/ab_*.txt; do echo "== $f"; cat $f; done
