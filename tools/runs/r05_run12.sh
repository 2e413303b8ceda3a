#!/bin/bash
# round-5 GPU call 12: frame-rate GEMM kernels with all staging loads in flight at once vs the build before (PREV6); parity of the GEMM paths
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r05_u; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -q > $O/pytest.log 2>&1; grep -h "passed\|failed" $O/pytest.log | tail -1
ab() { PWV_LIB=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact $3 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],4), round(d['value']/1e6,2))"; }
for k in 1 2 3; do
  ab "c1 new " "" "--case bench/c1"
  ab "c1 prev" tools/abl_so/libpwv_PREV6.so "--case bench/c1"
  ab "16k new " "" "--length 16000"
  ab "16k prev" tools/abl_so/libpwv_PREV6.so "--length 16000"
  ab "c3 new " "" ""
  ab "c3 prev" tools/abl_so/libpwv_PREV6.so ""
done | tee $O/ab_staging.txt
