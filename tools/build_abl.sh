#!/bin/bash
# Builds timing-only variants of the library for the A/B runs of tools/ab.sh:
#   tools/build_abl.sh ABL_NOXB16 ABL_BLOCK                       (round 4: tools/probes/persist_probes.patch)
#   PATCH=pair_probe.patch tools/build_abl.sh PAIR5:ABL_PAIR=5 PAIR5ST:"ABL_PAIR=5 ABL_PAIR_ONLYST"    (round 5)
# Each NAME[:DEFINES] becomes tools/abl_so/libpwv_NAME.so, compiled with -DPWV_<define> for every define (default: NAME itself)
# from a scratch copy of csrc/ with the patch applied (the probes are kept out of the product sources).  Results are WRONG by
# design for the ABL_* variants; only the step time is of interest.  (Rounds 1-2 used perturb.patch against the per-layer kernel
# of that time -- ADD_LDS / ADD_MFMA / ADD_VALU / ADD_P2 / ADD_X2 / ABL_NOP / ABL_PMFMA, HISTORY.md section 4, K1 item 6; git history has it.)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
patchfile=${PATCH:-persist_probes.patch}
tmp=$(mktemp -d)
cp -r "$root/parallel-wavenet-vocoder_amd/csrc" "$tmp/csrc"
[ "$patchfile" = none ] || (cd "$tmp/csrc" && patch -s -p1 < "$root/tools/probes/$patchfile")
mkdir -p "$root/tools/abl_so"
for v in "$@"; do
    name=${v%%:*}; defs=${v#*:}; [ "$defs" = "$v" ] && defs=$name
    flags=""; for d in $defs; do flags="$flags -DPWV_$d"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950:xnack- -O3 -std=c++17 -shared -fPIC $flags -I"$root/include" -I"$tmp/csrc" \
        -o "$root/tools/abl_so/libpwv_$name.so" "$tmp"/csrc/*.hip
    echo "built tools/abl_so/libpwv_$name.so ($flags)"
done
rm -rf "$tmp"
