#!/bin/bash
# gpurun_out/<tag>_* (made by tools/profile_round2.sh on the GPU box) -> profiles/<tag>_*
set -eu
tag=${1:-r02_x}
out=${2:-profiles}
cd "$(dirname "$0")/.."
mkdir -p $out
f() { find gpurun_out/$1 -name "*$2" | head -1; }
python tools/summarize_rocprof.py "$(f ${tag}_stats kernel_stats.csv)" $out/${tag}_kernel_stats.md \
    "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-exact (default: per-layer launches, two streams, HIP-graph replay)"
python tools/summarize_rocprof.py "$(f ${tag}_stats_persist kernel_stats.csv)" $out/${tag}_kernel_stats_persist.md \
    "the same command with PWV_PERSIST=1 (persistent dataflow launch for the residual layers 1..L-2 of every stack)"
python tools/summarize_rocprof.py "$(f ${tag}_stats_f32 kernel_stats.csv)" $out/${tag}_kernel_stats_f32.md \
    "rocprofv3 --kernel-trace --stats -- python bench.py --precision f32 --steps 5 --warmup 2 --no-cpu-baseline --no-f32-exact (exact-fp32 MFMA kernels, FIRST / HEAD fused)"
python tools/hbm_traffic.py "$(f ${tag}_pmc_FETCH_SIZE counter_collection.csv)" "$(f ${tag}_pmc_WRITE_SIZE counter_collection.csv)" \
    "layer_f16x3_kernel<false, false, false, false, false>" 81920000 $out/${tag}_hbm_traffic.json \
    "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-exact --no-graph" \
    "bench/c3, 1 x 160000 samples" "$(f ${tag}_stats kernel_stats.csv)"
{ echo "# SQ counters per launch (rocprofv3 --pmc, bench.py --no-graph --steps 2): per-layer launches (default)"; echo '```';
  python tools/pmc_summary.py "$(f ${tag}_sq_1 counter_collection.csv)" "$(f ${tag}_sq_2 counter_collection.csv)" | grep -A20 "layer_f16x3_kernel<false, false, false, false, false>\|stack_persist" | head -60;
  echo '```'; echo; echo "# the same with PWV_PERSIST=1 (stack_persist_kernel)"; echo '```';
  python tools/pmc_summary.py "$(f ${tag}_sqp_1 counter_collection.csv)" "$(f ${tag}_sqp_2 counter_collection.csv)" | grep -A20 "stack_persist" | head -60; echo '```'; } > $out/${tag}_sq_counters.md
cp gpurun_out/${tag}_bench_default.json $out/${tag}_bench_default.json
cp gpurun_out/${tag}_bench_persist.json $out/${tag}_bench_persist.json
ls -la $out | grep ${tag}
