#!/bin/bash
# round-6 GPU call 39: the -m gpu suite with the short-input instantiation up to 7 units per workgroup + its new test shapes; smoke; C client
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_z10; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -n "passed\|failed\|FAILED" $O/pytest.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
