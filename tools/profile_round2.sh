#!/bin/bash
# Round-2 profile set (run on the GPU box through gpurun from the repo root):
#   gpurun --timeout 900 -- 'tools/profile_round2.sh r02_e'
# Raw rocprofv3 output goes to gpurun_out/<tag>_* on the box and is summarised there into gpurun_out/<tag>_profiles/ (the
# only part that travels back); copy those files into profiles/.  Counter passes are separate runs with --kernel-trace only.
set -u
export TMPDIR=/tmp
tag=${1:-r02_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-f32-exact"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- $B --steps 10 --warmup 2 \
    > $R/gpurun_out/${tag}_stats.log 2>&1 < /dev/null
PWV_PERSIST=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats_persist -- $B --steps 10 --warmup 2 \
    > $R/gpurun_out/${tag}_stats_persist.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats_f32 -- $B --precision f32 --steps 5 --warmup 2 \
    > $R/gpurun_out/${tag}_stats_f32.log 2>&1 < /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${tag}_pmc_$c -- $B --steps 3 --warmup 1 --no-graph \
        > $R/gpurun_out/${tag}_pmc_$c.log 2>&1 < /dev/null
done
i=0
for grp in \
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_sq_$i -- $B --steps 2 --warmup 1 --no-graph \
        > $R/gpurun_out/${tag}_sq_$i.log 2>&1 < /dev/null
    PWV_PERSIST=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_sqp_$i -- $B --steps 2 --warmup 1 --no-graph \
        > $R/gpurun_out/${tag}_sqp_$i.log 2>&1 < /dev/null
done
cd $R
python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err < /dev/null
PWV_PERSIST=1 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_persist.json 2> gpurun_out/${tag}_bench_persist.err < /dev/null
# summarise here (the raw kernel traces are too big to travel back), keep only the summaries
tools/profile_round2_summarize.sh ${tag} gpurun_out/${tag}_profiles
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_stats_persist gpurun_out/${tag}_stats_f32 gpurun_out/${tag}_pmc_* gpurun_out/${tag}_sq_* gpurun_out/${tag}_sqp_*
ls gpurun_out/${tag}_profiles
