#!/bin/bash
# round-6 GPU call 22: where do the 1.6 % on C3 come from -- HEAD vs the short-input protocol alone (NOSWAP) vs + packed K order x[t] first (BASE)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_r; mkdir -p $O
tools/ab.sh 4 HEAD0 NOSWAP BASE | tee $O/ab_c3.txt
