#!/bin/bash
# round-6 GPU call 13: instance norm with its finalize kernel -- the whole -m gpu suite, then bench/in and its kernel stats
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_j; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -3
BENCH_ARGS="--case bench/in" tools/ab_env.sh 3 "PWV_FUSE_TAIL=1" | tee $O/bench_in.txt
python bench.py --no-cpu-baseline --no-f32-exact --case bench/in --steps 20 --warmup 3 2>/dev/null | grep "^{" > $O/r06_z_in_bench.json
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p6_in -- python $R/bench.py --no-cpu-baseline --no-f32-exact --case bench/in --steps 10 --warmup 2 > /tmp/p6_in.log 2>&1 < /dev/null
cd $R; python tools/summarize_rocprof.py "$(find /tmp/p6_in -name '*kernel_stats.csv' | head -1)" $O/r06_z_in_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-f32-exact --steps 10 --warmup 2 --case bench/in"
head -12 $O/r06_z_in_kernel_stats.md | cut -c1-180
