#!/bin/bash
# round-6 GPU call 24: the whole -m gpu suite on the SHORT instantiation
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_t; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -n "passed\|failed\|FAILED" $O/pytest.log | tail -15
