#!/bin/bash
# the -DPWV_PTRACE build of the library (per-wave cycle accounting in the persistent stack kernel), next to the real one
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950:xnack- -O3 -std=c++17 -shared -fPIC -DPWV_PTRACE -Iinclude \
  -Iparallel-wavenet-vocoder_amd/csrc -o tools/libpwv_ptrace.so parallel-wavenet-vocoder_amd/csrc/*.hip
