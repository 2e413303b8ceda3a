#!/bin/bash
# round-6 GPU call 9: the exact-fp32 tail (last layer + head + affine inside the persistent launch for PWV_PREC_F32 too), the MFMA route of the
# composed path's convolutions and the wider in_stats grid -- whole -m gpu suite, then the numbers: f32 step at C3 / 16000 samples with and
# without the tail (PWV_FUSE_TAIL=0), bench/in again
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_h; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -3
BENCH_ARGS="--precision f32" tools/ab_env.sh 3 "PWV_FUSE_TAIL=0" "PWV_FUSE_TAIL=1" | tee $O/ab_f32_tail_c3.txt
BENCH_ARGS="--precision f32 --length 16000" tools/ab_env.sh 3 "PWV_FUSE_TAIL=0" "PWV_FUSE_TAIL=1" | tee $O/ab_f32_tail_16k.txt
BENCH_ARGS="--length 16000" tools/ab_env.sh 2 "PWV_FUSE_TAIL=1" | tee $O/f16x3_16k.txt
BENCH_ARGS="--case bench/in" tools/ab_env.sh 2 "PWV_FUSE_TAIL=1" | tee $O/bench_in.txt
