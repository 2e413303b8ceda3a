b() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact "$@" 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$LBL', round(d['ms_per_step'],4), round(d['value']/1e6,2))"; }
for i in 1 2; do
LBL=c3_base b
LBL=c3_pwt PWV_LIB=tools/abl_so/libpwv_PWT.so b
LBL=c2_base b --case bench/c2
LBL=c2_pwt PWV_LIB=tools/abl_so/libpwv_PWT.so b --case bench/c2
done
