#!/bin/bash
# round-5 final check on the GPU box: the whole -m gpu suite, smoke(), the driver's bench line, the RCCL-at-world-1 launcher form
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r05_k; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed\|error" $O/pytest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
(time python bench.py > $O/bench_default.json 2> $O/bench_default.err) 2>&1 | grep real; echo "bench rc $?"
python -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['model']['hbm_frac_of_8TBs'], d['roofline']['frac'], d['roofline']['committed_profile']['frac_rocprof'], d['roofline']['traffic'], d['f32_exact']['mfma_frac'], d['cpu_baseline']['value_1thread'], d['cpu_baseline']['value_best'])"
PWV_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --case bench/c4 --no-cpu-baseline --no-f32-exact > $O/bench_force_dist_c4.json 2> $O/bench_force_dist_c4.err; echo "force-dist rc $?"
python -c "
import json
d=json.loads([l for l in open('$O/bench_force_dist_c4.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['rccl_ranks'], d['backend'], d['sharded_generate']); print(json.dumps(d['job']))"
