#!/bin/bash
# round-5 GPU call 5: the two-thread test with diagnostics, the co-tenant recovery run with a queue-flooding neighbour
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r05_f; mkdir -p $O
for k in 1 2 3; do timeout 300 python -m pytest tests/test_safe_call.py -m gpu -q -x -k two_threads 2>&1 | grep -n "passed\|failed\|AssertionError\|assert \|E  " | head -12; done
timeout 400 python tools/co_tenant_recovery.py --tenant-seconds 5 > $O/co_tenant_recovery.txt 2>$O/co_tenant_recovery.err; echo "co-tenant rc $?"; cat $O/co_tenant_recovery.txt; grep -v amdgpu.ids $O/co_tenant_recovery.err | tail -5
