#!/bin/bash
# (environment of the caller is inherited: e.g. PWV_PERSIST=1 tools/power_sample.sh; BENCH_ARGS adds bench flags)
BENCH_ARGS=${BENCH_ARGS:-}
# package power / clocks under the headline workload (profiles/rNN_power_clock.md): bench in the background, rocm-smi sampled
python bench.py --steps ${STEPS:-2500} --warmup 5 --no-cpu-baseline --no-f32-exact $BENCH_ARGS > /tmp/bench_bg.log 2>&1 &
pid=$!
sleep 16
for i in 1 2 3 4 5; do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Package Power|sclk|mclk|fclk|junction" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'
    echo
    sleep 0.7
done
wait $pid
grep -o '"ms_per_step": [0-9.]*' /tmp/bench_bg.log | head -1
