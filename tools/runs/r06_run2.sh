#!/bin/bash
# round-6 GPU call 2: (1) the register-stationary layer kernel with its P row prefetched through LDS (one wave per SIMD: nobody covers a load
# issued at the top of a unit); (2) which of the persistent kernel's weight planes are better read from global memory (L1 / L2) than from LDS:
# PWV_ABL_WGLOBAL mask 1 = filter|gate lo, 3 = + dense lo (= call 1's LOGLOBAL), 11 = + dense hi, 15 = everything
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_b; mkdir -p $O
PWV_PERSIST=0 PWV_REGW=0 python tools/probes/regw/regw_check.py /tmp/a.npy 2>&1 | tail -2
PWV_PERSIST=0 PWV_REGW=1 python tools/probes/regw/regw_check.py /tmp/b.npy 2>&1 | tail -2
python tools/probes/regw/regw_check.py --cmp /tmp/a.npy /tmp/b.npy | tee $O/regw_bits.txt
for k in 1 2; do for v in 0 1; do for d in 1 64 512; do
  echo -n "REGW=$v d=$d: "; PWV_REGW=$v python tools/kbench.py --precision 1 --dilation $d --iters 200 2>/dev/null | grep layer_residual
done; done; done | tee $O/kbench.txt
tools/ab.sh 3 BASE WG1 WG3 WG11 WG15 | tee $O/ab_wglobal.txt
