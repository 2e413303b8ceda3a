#!/bin/bash
# round-5 last check: suite, smoke, bench lines (default, C1, 16k) with the mel resident in the graph's input buffer
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r05_n; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed\|error" $O/pytest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
for c in "c3:" "c1:--case bench/c1" "16k:--length 16000" "c2:--case bench/c2" "c4:--case bench/c4"; do
  n=${c%%:*}; a=${c#*:}
  for k in 1 2; do python bench.py --no-cpu-baseline --no-f32-exact --steps 30 --warmup 5 $a 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$n', round(d['ms_per_step'],4), round(d['value']/1e6,2), round(d['model']['hbm_frac_of_8TBs'],3))"; done
done | tee $O/configs.txt
python -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['model']['hbm_frac_of_8TBs'], d['roofline']['frac'], d['roofline']['committed_profile']['source'], d['f32_exact']['mfma_frac'], d['cpu_baseline']['value_1thread'], d['cpu_baseline']['value_best'])"
