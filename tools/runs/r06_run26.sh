#!/bin/bash
# round-6 GPU call 37: two processes on one GPU with SHORT inputs (16000 samples): give-up, repair and recovery of the short-input instantiation
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z8; mkdir -p $O
timeout 600 python tools/co_tenant_recovery.py --length 16000 > $O/co_tenant_recovery_16k.txt 2>&1; tail -12 $O/co_tenant_recovery_16k.txt
