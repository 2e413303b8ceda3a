#!/bin/bash
# one bench.py run per BASELINE configuration (and the labelled fp16 extension) -> a markdown table on stdout
#   gpurun -- 'tools/config_table.sh > gpurun_out/configs.md'
echo "| config (per GPU) | precision | samples/s | x real-time @22.05 kHz | ms/step | K1 roofline frac | whole-model frac of 8 TB/s |"
echo "|---|---|---|---|---|---|---|"
run() {
python bench.py --no-cpu-baseline --no-f32-exact "$@" 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{}); m=d.get('model',{})
        print('| %s | %s | %.1f M | %.0f | %.3f | %s | %s |' % (d['config']['workload'].split(':')[0], d['dtype'].split(' ')[0], d['value']/1e6, d['value']/22050.0, d['ms_per_step'],
              ('%.3f' % r['frac']) if r.get('frac') else '-', ('%.3f' % m['hbm_frac_of_8TBs']) if m.get('hbm_frac_of_8TBs') else '-'))"
}
run --case bench/c1 --steps 50 --warmup 5
run --case bench/c2 --steps 20 --warmup 3
run --case bench/c3 --steps 20 --warmup 3
run --case bench/c3 --steps 10 --warmup 3 --precision f32
run --case bench/c4 --steps 6 --warmup 2
run --case bench/c5 --steps 4 --warmup 2
run --case bench/c3 --steps 20 --warmup 3 --precision f16
run --case bench/c4 --steps 6 --warmup 2 --precision f16
run --case bench/c5 --steps 4 --warmup 2 --precision f16
