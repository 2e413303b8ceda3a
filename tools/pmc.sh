#!/bin/bash
# usage: tools/pmc.sh TAG "COUNTER ..." [ENV...] -- one rocprofv3 --pmc pass (kernel trace only) over a short eager bench run;
# per-kernel means -> gpurun_out/TAG_pmc.txt
set -u; BENCH_ARGS=${BENCH_ARGS:-}
export TMPDIR=/tmp
tag=$1; ctr=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$tag -- python $R/bench.py --no-cpu-baseline --no-f32-exact --no-graph --steps 2 --warmup 1 $BENCH_ARGS > $R/gpurun_out/${tag}_pmc.log 2>&1 < /dev/null
cd $R
python tools/pmc_summary.py "$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)" > gpurun_out/${tag}_pmc.txt
rm -rf /tmp/pmc_$tag
