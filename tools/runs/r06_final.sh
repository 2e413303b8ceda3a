#!/bin/bash
# round-6 final GPU call: the whole -m gpu suite, the smoke entry, then the round's profile set (tools/profile_round6.sh r06_z) on the final build, one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
tools/profile_round6.sh r06_z > $O/profile.log 2>&1; tail -3 $O/profile.log
for k in 1 2; do python bench.py 2>/dev/null | grep "^{" > $O/bench_default_$k.json; python -c "
import json; d=json.load(open('$O/bench_default_$k.json')); print('bench default', round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), round(d['f32_exact']['value']/1e6,2), round(d['cpu_baseline']['value']), d['cpu_baseline']['threads_swept'])"; done | tee $O/bench_default.txt
