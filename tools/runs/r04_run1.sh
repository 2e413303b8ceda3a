#!/bin/bash
# round-4 GPU call 1: the whole -m gpu suite, the driver's bench line, A/B pairs (look-back upper bound, stream forms of the
# per-layer path), the kernel timeline of one C3 step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r04_a; mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
tools/ab.sh 3 BASE ABL_NOXB16 > $O/ab_noxb16.txt 2>&1
BENCH_ARGS="--case bench/c5" tools/ab_env.sh 2 PWV_TWO_STREAMS=1 PWV_TWO_STREAMS=0 > $O/ab_c5_streams.txt 2>&1
BENCH_ARGS="--case bench/c5 --precision f16" tools/ab_env.sh 2 PWV_TWO_STREAMS=1 PWV_TWO_STREAMS=0 > $O/ab_c5f16_streams.txt 2>&1
tools/ab_env.sh 2 "PWV_PERSIST=0 PWV_TWO_STREAMS=1" "PWV_PERSIST=0 PWV_TWO_STREAMS=0" > $O/ab_c3_perlayer_streams.txt 2>&1
NLAST=24 tools/timeline.sh r04_a_c3; mv gpurun_out/r04_a_c3_timeline.txt $O/ 2>/dev/null; rm -f gpurun_out/r04_a_c3_timeline.log
cat $O/ab_*.txt; tail -30 $O/r04_a_c3_timeline.txt
