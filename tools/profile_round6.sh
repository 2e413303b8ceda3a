#!/bin/bash
# Round-6 profile set (profile_round4.sh + the f-4 variants, the short-input and exact-fp32 rows) (run on the GPU box through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'tools/profile_round6.sh r06_x'
# Per benched configuration: the bench line, rocprofv3 --kernel-trace --stats of the same command, and three PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ counters; --kernel-trace only, never with other trace domains).  Raw rocprofv3 output stays in
# /tmp on the box; tools/profile_round4_summarize.py turns it into gpurun_out/<tag>_profiles/ (the only part that travels
# back): <tag>_<cfg>_kernel_stats.md, <tag>_<cfg>_hbm_traffic.json, <tag>_<cfg>_bench.json, <tag>_configs.md -- copy into profiles/.
set -u
export TMPDIR=/tmp
tag=${1:-r04_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${tag}_profiles
mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-f32-exact"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE"
run_stats() {   # name, env..., -- bench args
    local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    cd /tmp; env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4_$name -- $B --steps 10 --warmup 2 "$@" > /tmp/p4_${name}_stats.log 2>&1 < /dev/null
    cd $R; cp "$(find /tmp/p4_$name -name '*kernel_stats.csv' | head -1)" /tmp/p4_${name}_stats.csv
    python tools/summarize_rocprof.py /tmp/p4_${name}_stats.csv $O/${tag}_${name}_kernel_stats.md \
        "rocprofv3 --kernel-trace --stats -- ${envs[*]} python bench.py --no-cpu-baseline --no-f32-exact --steps 10 --warmup 2 $*"
    rm -rf /tmp/p4_$name
}
run_pmc() {     # name, "counters", -- bench args   -> /tmp/p4_<name>.csv
    local name=$1 ctr=$2; shift 3
    cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/p4_$name -- $B --no-graph --steps 2 --warmup 1 "$@" > /tmp/p4_${name}.log 2>&1 < /dev/null
    cd $R; cp "$(find /tmp/p4_$name -name '*counter_collection.csv' | head -1)" /tmp/p4_$name.csv; rm -rf /tmp/p4_$name
}
for cfg in "c3:--case bench/c3" "c4:--case bench/c4" "c5:--case bench/c5" "c5_f16:--case bench/c5 --precision f16" "skip:--case bench/skip"; do
    n=${cfg%%:*}; a=${cfg#*:}
    cd $R; $B $a --steps 20 --warmup 3 > $O/${tag}_${n}_bench.json 2> /tmp/p4_${n}_bench.err < /dev/null
    run_stats $n -- $a
    run_pmc ${n}_fetch FETCH_SIZE -- $a
    run_pmc ${n}_write WRITE_SIZE -- $a
    run_pmc ${n}_sq "$SQ" -- $a
done
for cfg in "in:--case bench/in" "16k:--length 16000" "16k_f32:--length 16000 --precision f32" "c3_f32:--precision f32" "c1:--case bench/c1" "c2:--case bench/c2"; do
    n=${cfg%%:*}; a=${cfg#*:}
    cd $R; $B $a --steps 20 --warmup 3 > $O/${tag}_${n}_bench.json 2> /tmp/p4_${n}_bench.err < /dev/null
done
run_stats in -- --case bench/in
run_stats 16k -- --length 16000
run_stats c3_perlayer PWV_PERSIST=0 -- --case bench/c3
run_stats c3_f32 -- --case bench/c3 --precision f32
cd $R; PWV_PERSIST=0 $B --steps 20 --warmup 3 > $O/${tag}_c3_perlayer_bench.json 2> /dev/null < /dev/null
python bench.py > $O/${tag}_bench_default.json 2> /tmp/p4_bench_default.err < /dev/null
python tools/profile_round4_summarize.py $O /tmp $tag > $O/${tag}_configs.md
STEPS=9000 tools/power_sample.sh > $O/${tag}_power_clock.txt 2>&1
ls -la $O
