#!/bin/bash
# usage: tools/kstats.sh TAG [ENV...] -- rocprofv3 kernel stats of a short bench run under ENV; writes gpurun_out/TAG_kstats.md
set -u; BENCH_ARGS=${BENCH_ARGS:-}
export TMPDIR=/tmp
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- python $R/bench.py --no-cpu-baseline --no-f32-exact --steps 10 --warmup 2 $BENCH_ARGS > $R/gpurun_out/${tag}_kstats.log 2>&1 < /dev/null
cd $R
python tools/summarize_rocprof.py "$(find /tmp/ks_$tag -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kstats.md "kernel stats: $* bench.py --steps 10 --warmup 2 $BENCH_ARGS"
[ -n "${KPAT:-}" ] && python tools/kernel_durations.py "$(find /tmp/ks_$tag -name '*kernel_trace.csv' | head -1)" "$KPAT" 48 >> gpurun_out/${tag}_kstats.md
rm -rf /tmp/ks_$tag
