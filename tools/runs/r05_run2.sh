#!/bin/bash
# round-5 GPU call 2: the tail (last layer + head + affine inside the persistent launch): bit-identity, then what it is worth
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r05_b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_persist.py tests/test_safe_call.py -m gpu -q -x > $O/pytest_persist.log 2>&1; echo "pytest persist rc $?"; tail -15 $O/pytest_persist.log
ab() {  # label, env, bench args
  env $2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact $3 2>$O/err.txt < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],4), round(d['value']/1e6,2))"
}
for k in 1 2 3; do
  ab "c3 tail" PWV_FUSE_TAIL=1 ""
  ab "c3 sep " PWV_FUSE_TAIL=0 ""
  ab "16k tail" PWV_FUSE_TAIL=1 "--length 16000"
  ab "16k sep " PWV_FUSE_TAIL=0 "--length 16000"
  ab "c1 tail" PWV_FUSE_TAIL=1 "--case bench/c1"
  ab "c1 sep " PWV_FUSE_TAIL=0 "--case bench/c1"
done > $O/ab_tail.txt 2>&1
cat $O/ab_tail.txt; tail -3 $O/err.txt
for k in 1 2; do
  ab "c4 tail" PWV_FUSE_TAIL=1 "--case bench/c4"
  ab "c4 sep " PWV_FUSE_TAIL=0 "--case bench/c4"
  ab "c2 tail" PWV_FUSE_TAIL=1 "--case bench/c2"
  ab "c2 sep " PWV_FUSE_TAIL=0 "--case bench/c2"
done > $O/ab_tail2.txt 2>&1
cat $O/ab_tail2.txt
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed\|error" $O/pytest.log | tail -5
