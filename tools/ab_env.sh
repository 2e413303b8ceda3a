#!/bin/bash
# usage: tools/ab_env.sh ROUNDS "ENV1" "ENV2" ...   -- alternates bench.py runs under different environments on ONE box
# (e.g. tools/ab_env.sh 3 "PWV_PERSIST=0" "PWV_PERSIST=1"); prints ms/step per run.  Extra bench flags: BENCH_ARGS.
rounds=$1; shift
for k in $(seq $rounds); do for v in "$@"; do
env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact $BENCH_ARGS 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', round(d['ms_per_step'],4), round(d['value']/1e6,2))"
done; done
