#!/bin/bash
# round-6 GPU call 6: the whole -m gpu suite on the build with struct_size / PWV_HIP_VERSION 300, the tail-aware probe, the noise rewind, the pooled status words
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_e; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; tail -15 $O/pytest.log
