#!/bin/bash
# round-4 final check on the GPU box: the whole -m gpu suite, smoke(), the driver's bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r04_f; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['model']['hbm_frac_of_8TBs'], d['roofline']['frac'], d['roofline'].get('frac_rocprof'), d['roofline'].get('traffic'), d['f32_exact']['mfma_frac'], d['cpu_baseline']['value'])"
