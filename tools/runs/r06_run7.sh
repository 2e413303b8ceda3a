#!/bin/bash
# round-6 GPU call 8: the f-4 variants -- parity of the re-scheduled SKIP kernels (two-pass skip GEMM, late skip-sum load),
# bench lines + rocprofv3 kernel stats of bench/skip (use_skip_connection: True) and bench/in (normalize_wavenet: 'in'), A/B of the SKIP kernels against the previous build
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r06_g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_unfused_and_e2e.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-f32-exact"
for c in skip in; do
  $B --case bench/$c --steps 10 --warmup 2 > $O/r06_${c}_bench.json 2> $O/${c}_bench.err < /dev/null
  python - <<PY
import json
d=json.loads([l for l in open('$O/r06_${c}_bench.json') if l.startswith('{')][-1])
print('bench/$c', round(d['value']/1e6,3), 'M samples/s', round(d['ms_per_step'],3), 'ms', d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'), d['model']['hbm_frac_of_8TBs'])
PY
  cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p6_$c -- python $R/bench.py --no-cpu-baseline --no-f32-exact --case bench/$c --steps 5 --warmup 2 > /tmp/p6_$c.log 2>&1 < /dev/null
  cd $R; python tools/summarize_rocprof.py "$(find /tmp/p6_$c -name '*kernel_stats.csv' | head -1)" $O/r06_${c}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-f32-exact --case bench/$c --steps 5 --warmup 2"
  head -12 $O/r06_${c}_kernel_stats.md
done | tee $O/f4.txt
