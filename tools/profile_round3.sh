#!/bin/bash
# Round-3 profile set (run on the GPU box through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'tools/profile_round3.sh r03_e'
# Raw rocprofv3 output stays in /tmp on the box; the summaries land in gpurun_out/<tag>_profiles/ (the only part that travels
# back) -- copy them into profiles/.  Counter passes are separate runs with --kernel-trace only (never with other trace domains).
set -u
export TMPDIR=/tmp
tag=${1:-r03_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${tag}_profiles
mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-f32-exact"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE"
run_stats() {   # name, env..., -- bench args
    local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    cd /tmp; env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3_$name -- $B --steps 10 --warmup 2 "$@" > $O/${name}_stats.log 2>&1 < /dev/null
    cd $R; python tools/summarize_rocprof.py "$(find /tmp/p3_$name -name '*kernel_stats.csv' | head -1)" $O/${name}_kernel_stats.md \
        "rocprofv3 --kernel-trace --stats -- ${envs[*]} python bench.py --no-cpu-baseline --no-f32-exact --steps 10 --warmup 2 $*"
    cp "$(find /tmp/p3_$name -name '*kernel_stats.csv' | head -1)" /tmp/p3_${name}_stats.csv; rm -rf /tmp/p3_$name
}
run_pmc() {     # name, "counters", env..., -- bench args   -> /tmp/p3_<name>.csv
    local name=$1 ctr=$2; shift 2; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    cd /tmp; env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/p3_$name -- $B --no-graph --steps 2 --warmup 1 "$@" > $O/${name}.log 2>&1 < /dev/null
    cd $R; cp "$(find /tmp/p3_$name -name '*counter_collection.csv' | head -1)" /tmp/p3_$name.csv; rm -rf /tmp/p3_$name
}
for cfg in "c3:--case bench/c3" "c4:--case bench/c4" "c5:--case bench/c5" "c5f16:--case bench/c5 --precision f16"; do
    n=${cfg%%:*}; a=${cfg#*:}
    run_stats $n -- $a
    run_pmc ${n}_fetch FETCH_SIZE -- $a
    run_pmc ${n}_write WRITE_SIZE -- $a
    run_pmc ${n}_sq "$SQ" -- $a
    cd $R; $B $a --steps 20 --warmup 3 > $O/${n}_bench.json 2> $O/${n}_bench.err < /dev/null
done
run_stats c3_perlayer PWV_PERSIST=0 -- --case bench/c3
run_stats c3_f32 -- --case bench/c3 --precision f32
cd $R; PWV_PERSIST=0 $B --steps 20 --warmup 3 > $O/c3_perlayer_bench.json 2> /dev/null < /dev/null
python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null
python tools/profile_round3_summarize.py $O /tmp > $O/configs.md
STEPS=9000 tools/power_sample.sh > $O/power_clock.txt 2>&1
rm -f $O/*.log
ls -la $O
