#!/bin/bash
# round-6 GPU call 5: steady-state unit rate of the register-stationary layer kernel (960000 rows: 234 units per workgroup, the 64 KB-per-wave
# weight prologue amortised) against the 8-wave LDS-fed kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_d; mkdir -p $O
for k in 1 2; do for v in 0 1; do for rows in 160000 960000; do
  echo -n "REGW=$v rows=$rows d=64: "; PWV_REGW=$v python tools/kbench.py --precision 1 --dilation 64 --rows $rows --iters 100 2>/dev/null | grep layer_residual
done; done; done | tee $O/kbench_rows.txt
