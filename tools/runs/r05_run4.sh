#!/bin/bash
# round-5 GPU call 4: the profile set of the round (tools/profile_round4.sh, tag r05_e), the other configurations' bench lines, the
# co-tenant recovery run (a give-up suspends the persistent launches; they come back), the whole -m gpu suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r05_d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed\|error" $O/pytest.log | tail -3
timeout 300 python tools/co_tenant_recovery.py > $O/co_tenant_recovery.txt 2>$O/co_tenant_recovery.err; echo "co-tenant rc $?"; cat $O/co_tenant_recovery.txt; grep -v amdgpu.ids $O/co_tenant_recovery.err | tail -5
for c in "c1:--case bench/c1" "c2:--case bench/c2" "16k:--length 16000" "64k3:--length 64000 --utts 3"; do
  n=${c%%:*}; a=${c#*:}
  python bench.py --no-cpu-baseline --no-f32-exact --steps 30 --warmup 5 $a > $O/bench_$n.json 2>/dev/null < /dev/null
  python -c "
import json
d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1]); print('$n', round(d['ms_per_step'],4), round(d['value']/1e6,2), round(d['model']['hbm_frac_of_8TBs'],3))"
done | tee $O/other_configs.txt
timeout 1500 bash tools/profile_round4.sh r05_e > $O/profile.log 2>&1; echo "profile rc $?"; tail -5 $O/profile.log
cat gpurun_out/r05_e_profiles/r05_e_configs.md
