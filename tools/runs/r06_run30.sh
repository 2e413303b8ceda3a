#!/bin/bash
# round-6 GPU call 42: error of both arithmetics against the fp64 oracle after the change of the packed K order
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z13; mkdir -p $O
timeout 900 python tools/precision_report.py > $O/precision.txt 2>&1; tail -15 $O/precision.txt
