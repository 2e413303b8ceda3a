#!/bin/bash
# round-6 GPU call 3 (redone as call 4: the first w_global.patch addressed the global copy through the LDS pointer + a distance, which the compiler
# turned back into out-of-range ds_read -- zeros: its "-10 %" measured zero-operand MFMAs, not a memory path; tests/test_gpu_persist.py caught it).
# Now real global_load_dwordx4: mask 1 = filter|gate lo, 3 = + dense lo, 10 = dense hi + lo, 15 = everything
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_c; mkdir -p $O
for v in WG3 WG15; do
PWV_LIB=tools/abl_so/libpwv_$v.so timeout 900 python -m pytest tests/test_gpu_persist.py -m gpu -q -x 2>&1 | tail -2 | tee $O/pytest_$v.txt
done
tools/ab.sh 3 BASE WG1 WG3 WG10 WG15 | tee $O/ab_wglobal2.txt
