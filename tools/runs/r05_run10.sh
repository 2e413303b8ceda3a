cd ${GRAFT_REPO_ROOT:-.}
ab() { PWV_LIB=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-exact $3 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],4), round(d['value']/1e6,2))"; }
for k in 1 2 3; do
  ab "16k base " "" "--length 16000"
  ab "16k plain" tools/abl_so/libpwv_PLAINSHARED.so "--length 16000"
  ab "c1 base " "" "--case bench/c1"
  ab "c1 plain" tools/abl_so/libpwv_PLAINSHARED.so "--case bench/c1"
done
