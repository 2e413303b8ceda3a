#!/bin/bash
# Reproduce the profiles/ files of a round on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'tools/profile_round.sh r01_d'
# Writes raw rocprofv3 output under gpurun_out/ and prints the commands that turn it into profiles/<tag>_*.
# Counter passes are separate runs with --kernel-trace only (never combined with sys/hip/hsa traces).
set -u
export TMPDIR=/tmp
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- \
    python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${tag}_pmc_$c -- \
        python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $R/gpurun_out/${tag}_pmc_$c.log 2>&1
done
i=0
for grp in \
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC" \
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_sq_$i -- \
        python $R/tools/kbench.py --precision 1 --G 2 --rows 160000 --iters 20 > $R/gpurun_out/${tag}_sq_$i.log 2>&1
done
cd $R
python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
cat <<MSG
then, locally:
  python tools/summarize_rocprof.py \$(find gpurun_out/${tag}_stats -name '*kernel_stats.csv') profiles/${tag}_kernel_stats.md "<title>"
  python tools/hbm_traffic.py \$(find gpurun_out/${tag}_pmc_FETCH_SIZE -name '*counter_collection.csv') \\
         \$(find gpurun_out/${tag}_pmc_WRITE_SIZE -name '*counter_collection.csv') "layer_f16x3_kernel<false, false, false>" \\
         81920000 profiles/${tag}_hbm_traffic.json "<command>" "<workload>"
  for i in 1 2 3 4; do python tools/pmc_summary.py \$(find gpurun_out/${tag}_sq_\$i -name '*counter_collection.csv'); done   # -> profiles/${tag}_sq_counters.md
  cp gpurun_out/${tag}_bench_default.json profiles/
MSG
