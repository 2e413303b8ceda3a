for k in 1 2 3; do s=$SECONDS; timeout 300 python -m pytest tests/test_gpu_unfused_and_e2e.py -x -q -k "multi_rank" 2>&1 | tail -1; echo "  took $((SECONDS-s)) s"; done
echo two independent benches on one GPU:
s=$SECONDS
python bench.py --no-cpu-baseline --no-f32-exact --steps 10 --warmup 2 > gpurun_out/co_a.json 2> gpurun_out/co_a.err &
python bench.py --no-cpu-baseline --no-f32-exact --steps 10 --warmup 2 > gpurun_out/co_b.json 2> gpurun_out/co_b.err
wait
echo "  took $((SECONDS-s)) s"
for f in a b; do python -c "
import json
for l in open('gpurun_out/co_$f.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$f', round(d['ms_per_step'],3), d['roofline']['kernel'][:40])"; grep -v amdgpu.ids gpurun_out/co_$f.err | tail -3; done
