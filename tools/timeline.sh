#!/bin/bash
# usage: tools/timeline.sh TAG [ENV...] -- kernel timeline of the last bench step (rocprofv3 kernel trace) -> gpurun_out/TAG_timeline.txt
set -u; BENCH_ARGS=${BENCH_ARGS:-}
export TMPDIR=/tmp
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python $R/bench.py --no-cpu-baseline --no-f32-exact --steps 4 --warmup 2 $BENCH_ARGS > $R/gpurun_out/${tag}_timeline.log 2>&1 < /dev/null
cd $R
python tools/step_timeline.py /tmp/tl_$tag ${NLAST:-300} > gpurun_out/${tag}_timeline.txt
rm -rf /tmp/tl_$tag
