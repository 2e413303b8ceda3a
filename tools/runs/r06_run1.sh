#!/bin/bash
# round-6 GPU call 1: the register-stationary unit body (csrc/pwv_layer_regw.hip: 4 waves x 512 registers, filter|gate fragments in AGPRs)
# priced as a per-layer launch against the 8-wave LDS-fed kernel: bit-identity, kernel micro-benchmark, per-layer bench step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_a; mkdir -p $O
PWV_PERSIST=0 PWV_REGW=0 python tools/probes/regw/regw_check.py /tmp/a.npy 2>&1 | tail -2
PWV_PERSIST=0 PWV_REGW=1 python tools/probes/regw/regw_check.py /tmp/b.npy 2>&1 | tail -2
python tools/probes/regw/regw_check.py --cmp /tmp/a.npy /tmp/b.npy | tee $O/regw_bits.txt
for k in 1 2; do for v in 0 1; do for d in 1 64 512; do
  echo -n "REGW=$v d=$d: "; PWV_REGW=$v python tools/kbench.py --precision 1 --dilation $d --iters 200 2>/dev/null | grep layer_residual
done; done; done | tee $O/kbench.txt
BENCH_ARGS="" tools/ab_env.sh 3 "PWV_PERSIST=0 PWV_REGW=0" "PWV_PERSIST=0 PWV_REGW=1" "PWV_PERSIST=1" | tee $O/ab_perlayer.txt
# the `lo` A-fragment planes of the persistent kernel read from global memory (L1 / L2) instead of LDS -- would free 80 KB of LDS
tools/ab.sh 3 BASE LOGLOBAL | tee $O/ab_loglobal.txt
