#!/bin/bash
# round-6 GPU call 7: the whole -m gpu suite (no -x) + the range-guard table + the default bench line with the new cpu_baseline sweep
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_f; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 900 python tools/precision_report.py --range 16000 --json $O/range_table.json 2>&1 | grep -v amdgpu.ids | tee $O/range_table.txt
( time python bench.py ) > $O/bench_default.log 2>&1; grep "^{" $O/bench_default.log > $O/bench_default.json; tail -4 $O/bench_default.log | cut -c1-600
