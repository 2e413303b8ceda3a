#!/bin/bash
# round-6 GPU call 36: bench.py with its short_input leg; two processes on one GPU (give-up and recovery) with the short-input instantiation
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z7; mkdir -p $O
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; python -c "
import json
for l in open('$O/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), d.get('short_input'))"
timeout 600 bash tools/co_tenant_check.sh > $O/co_tenant.txt 2>&1; tail -12 $O/co_tenant.txt
timeout 600 python tools/co_tenant_recovery.py > $O/co_tenant_recovery.txt 2>&1; tail -8 $O/co_tenant_recovery.txt
