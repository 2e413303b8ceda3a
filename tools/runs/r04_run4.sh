#!/bin/bash
# round-4 GPU call 4: where a SHORT forward's time goes (default model at 1 x 16000, C1): kernel timelines of one step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r04_d; mkdir -p $O
BENCH_ARGS="--case bench/c3 --length 16000" NLAST=40 tools/timeline.sh r04_d_short; mv gpurun_out/r04_d_short_timeline.txt $O/; rm -f gpurun_out/r04_d_short_timeline.log
BENCH_ARGS="--case bench/c1" NLAST=20 tools/timeline.sh r04_d_c1; mv gpurun_out/r04_d_c1_timeline.txt $O/; rm -f gpurun_out/r04_d_c1_timeline.log
BENCH_ARGS="--case bench/c3 --length 16000" tools/ab_env.sh 2 PWV_PERSIST=1 > $O/short_bench.txt 2>&1
BENCH_ARGS="--case bench/c1" tools/ab_env.sh 2 PWV_PERSIST=1 > $O/c1_bench.txt 2>&1
cat $O/*.txt
