#!/bin/bash
# usage: tools/ab.sh ROUNDS VARIANT...   (VARIANT = BASE or a name under tools/abl_so/libpwv_NAME.so); prints ms/step of bench.py per run
rounds=$1; shift
for k in $(seq $rounds); do for v in "$@"; do lib=tools/abl_so/libpwv_$v.so; [ $v = BASE ] && lib=""
PWV_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', round(d['ms_per_step'],4))"
done; done
