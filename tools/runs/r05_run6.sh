#!/bin/bash
# round-5 GPU call 6: thread-local variable scopes (the two-thread test, three times), the whole suite, smoke, the recovery timeline
# with a poked give-up, the composed path priced, the driver-style bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r05_g; mkdir -p $O
for k in 1 2 3; do timeout 300 python -m pytest tests/test_safe_call.py -m gpu -q -x -k two_threads 2>&1 | tail -1; done
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed\|error" $O/pytest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout 300 python tools/co_tenant_recovery.py --poke 20 > $O/recovery_poke.txt 2>$O/recovery_poke.err; echo "recovery rc $?"; cat $O/recovery_poke.txt
timeout 300 python tools/unfused_bench.py > $O/unfused_bench.txt 2>$O/unfused_bench.err; echo "unfused rc $?"; cat $O/unfused_bench.txt; tail -3 $O/unfused_bench.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['model']['hbm_frac_of_8TBs'], d['roofline']['frac'], d['roofline']['committed_profile']['source'], d['roofline']['committed_profile']['frac_rocprof'], d['roofline']['traffic'], d['f32_exact']['mfma_frac'], d['cpu_baseline']['value_1thread'], d['cpu_baseline']['value_best'], d['cpu_baseline']['threads_swept'])"
