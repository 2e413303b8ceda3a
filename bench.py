#!/usr/bin/env python
"""bench.py -- throughput of the IAF-WaveNet student generation path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1: one process per GPU over RCCL.  Started under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`
    the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; started plainly (`python bench.py
    --gpus N`) the script re-executes itself under torch.distributed.run with N ranks on 127.0.0.1.

One "step" = one full forward of the reference-default model (hparams/default.yaml as
models.py:36-65 builds it: 4 IAF flows, separate scalar + shifter WaveNets = 8 nets, 120
dilated layers, 'repeat' conditioning) on one batch of synthetic input: mel already in HBM
-> logistic noise sampled on the device -> waveform in HBM, weights resident.  Workload
("C3"): 1 utterance x 160000 samples (10 s @ 16 kHz, 2001 mel frames x 80
mels, hop 80) per GPU; utterances shard across GPUs with no data-path collective (weak scaling).

The timed loop replays a HIP graph of the forward (pwv_amd/graph.py; --no-graph enqueues every launch from the
host); the logistic sampler is a node of that graph and draws a fresh counter range on every replay.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline      -- the dominant kernel (fused gated-residual layer): algorithmic bytes (512 B per sample, net and
                   layer) per launch x concurrent launches / its mean launch duration, measured live with HIP events
                   the library records on the launch streams around each chain's run of residual-layer launches
                   (pwv_stack_args.ev_begin / ev_end, production launch path) against 8 TB/s; `traffic` = HBM bytes per
                   launch from the committed PMC passes (profiles/); the matrix-pipe view is beside it as roofline_mfma
                   (with --precision f32 the roles swap: fp32 MFMA roofline, HBM view beside it)
  short_input   -- (N=1, default workload) the same model on ONE utterance of 16000 samples under graph replay: the latency-bound end of the
                   path, where the persistent launch takes its short-input instantiation; reported beside `value`, never part of it
  cpu_baseline  -- the oracle's torch-CPU fp32 port of the same model timed on this host's cores on a bounded sample
                   (rank 0, N=1 only), cache-blocked in time and swept over {1, 2, 4, 8, 16, 32, physical cores} workers:
                   `value_1thread`, `value_best` (= `value`), `threads_swept`; a reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, = fp32 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense fp16/bf16 MFMA
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# algorithmic work of ONE net-layer on ONE sample (hoisted conditioning):
# 2*(W*R*2D + D*R) FLOP = 2*(2*64*128 + 64*64); residual stream read once + written once.
LAYER_FLOP_PER_SAMPLE = 2 * (2 * 64 * 128 + 64 * 64)      # 40960
LAYER_BYTES_PER_SAMPLE = 2 * 64 * 4                       # 512

# what holds the split-fp16 unit body below both nominal roofs (history in HISTORY.md section 4, K1 / K1p; the files are in profiles/ and tools/probes/)
LIMITER_K1P = ('neither nominal roof binds: by arithmetic intensity (3 x 80 = 240 fp16-FLOP/B < 312) the HBM roof is the nominal one, but the unit body is '
               'bound by ISSUE and CLOCK.  (1) issue: MFMA and VALU instructions of the two waves of a SIMD issue serially (tools/probes/coissue.hip: '
               '32 cycles per MFMA + 2-2.7 per VALU instruction, same counts on 8 workgroups at 2.4 GHz), so 120 MFMA + ~790 VALU + 96 transcendentals per '
               '32-row unit put the matrix pipe at 49-52 % busy inside the persistent launch (SQ_VALU_MFMA_BUSY_CYCLES, per-configuration table of the '
               'newest profiles/rNN_*_configs.md); (2) clock: rocm-smi reads the 1400 W package cap with sclk at 1.7-1.9 of 2.4 GHz under this bench '
               '(profiles/rNN_*_power_clock.txt), so added work costs time in proportion and idle time comes back as clock (tools/idle_probe.py).  '
               'The 3-term split cannot shed MFMAs at fp32 accuracy; HBM traffic is 1.04-1.10 x algorithmic; configurations whose working set leaves the '
               '256 MB Infinity Cache (C4, C5) run the unit at the same rate as C3, so HBM residency is not the limit either')


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--case', default='bench/c3', help='hparams case (bench/c1..c5, default, test/tran ...)')
    ap.add_argument('--length', type=int, default=0, help='override samples per utterance')
    ap.add_argument('--utts', type=int, default=0, help='override utterances per GPU')
    ap.add_argument('--precision', default='f16x3', choices=['f16x3', 'f32', 'f16'],
                    help="GEMM arithmetic: f16x3 = 3-term split-fp16 MFMA, fp32 accumulate (default, fp32 parity); f32 = exact "
                         "fp32 MFMA; f16 = REDUCED PRECISION build extension (fp16 residual stream, BASELINE config 5) -- "
                         "never the headline number")
    ap.add_argument('--shard', default='utterance', choices=['utterance', 'time'],
                    help="N > 1: 'utterance' = every GPU its own utterance(s) (weak scaling, the default); 'time' = ONE set of utterances cut along "
                         "the time axis with overlap-and-discard (exact, pwv_amd/timeshard.py): strong scaling of a long utterance (bench/c5)")
    ap.add_argument('--no-graph', action='store_true', help='enqueue every launch from the host instead of replaying a HIP graph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='target CPU time of the cpu_baseline sample')
    ap.add_argument('--no-f32-exact', action='store_true', help="skip the second timing of the same workload in exact fp32 ('f32_exact')")
    ap.add_argument('--cpu-baseline-child', action='store_true', help=argparse.SUPPRESS)      # (internal: the cpu_baseline leg in a clean process)
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def physical_cores():
    """Physical cores of this host (lscpu-style: distinct (package, core) pairs of /proc/cpuinfo; SMT siblings counted once)."""
    try:
        pairs, phys, core = set(), None, None
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('physical id'):
                    phys = line.split(':', 1)[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':', 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if pairs:
            return min(len(pairs), os.cpu_count() or len(pairs))
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_baseline_measure(case_cfg, target_s, chunk=4000, mode='process', per=4):
    """Oracle port (torch-CPU fp32 restatement of modules.py:11-259 / models.py:23-136) timed on the host cores on a bounded
    sample of the same model (k = 1 and k = all physical cores; generate.py:47-50 is the reference's own
    CPU / GPU switch).  Evaluated flow by flow in time chunks of `chunk` samples with each flow's look-back recomputed
    (oracle/torch_cpu.py ChunkedForward: the same function; a chunk's working set stays in cache instead of streaming 41 MB
    tensors through DRAM) and with one chunk per core at a time (k workers x single-threaded ops) instead of every small conv
    split over all cores -- the form in which this math scales on a many-core host.

    Every point of the sweep {1, 2, 4, 8, 16, 32, physical cores} is measured the same way (beyond 16
    workers the sweep used to time ONE chunk per worker, i.e. the pool's start-up): the pool is started and WARMED (one untimed
    chunk per worker: processes forked, op primitives created, pages touched), then a forward over `per` = 4 chunks per worker is
    timed.  The reported value is the best k measured AGAIN on that very sample, repeated until `target_s` seconds are filled --
    so `value_best` and its own sweep point are the same experiment."""
    from oracle import iaf_oracle as O
    from oracle.torch_cpu import ChunkedForward
    w = O.init_weights(case_cfg, seed=2)
    hop = case_cfg.hop_length
    chunk = max(hop, chunk // hop * hop)
    logical, phys = os.cpu_count() or 1, physical_cores()
    sweep = sorted({k for k in (1, 2, 4, 8, 16, 32, phys) if k <= logical})

    def point(k, seconds=0.0, per_k=per):
        f = ChunkedForward(w, case_cfg, chunk, k, mode)
        try:
            f(*O.synthetic_inputs(1, chunk * k, case_cfg))                      # warm: one chunk per worker, untimed
            mel, z = O.synthetic_inputs(1, chunk * per_k * k, case_cfg)
            n, t0 = 0, time.perf_counter()
            while True:
                y = f(mel, z)
                n += 1
                dt = time.perf_counter() - t0
                if dt >= seconds:
                    break
            assert np.isfinite(y).all()
            return chunk * per_k * k * n / dt, chunk * per_k * k * n, dt
        finally:
            f.close()

    rates, pers, prev = {}, {}, None
    for k in sweep:
        # `per` chunks per worker -- fewer only where the point would run beyond ~12 s at the rate of the point before it (the
        # physical-core point of a 128-core host past the memory-bound knee: 2 M samples at a third of the best rate)
        pers[k] = per if prev is None else int(max(1, min(per, 12.0 * prev // (k * chunk))))
        rates[k] = prev = point(k, per_k=pers[k])[0]
    best = max(rates, key=lambda k: rates[k])
    first_pass = rates[best]
    value, length, dt = point(best, target_s, pers[best])      # the best k again, on the same sample, until target_s seconds are filled ...
    rates[best] = value                                        # ... and THAT is its point of the sweep (the first pass is kept beside it)
    cpu_model = 'unknown CPU'
    try:
        with open('/proc/cpuinfo') as fh:
            cpu_model = next(l.split(':', 1)[1].strip() for l in fh if l.startswith('model name'))
    except Exception:
        pass
    return {'value': value, 'unit': 'samples/s', 'cores': best, 'kind': 'port',
            'value_1thread': rates[1], 'value_best': value, 'value_best_first_pass': first_pass, 'value_physical_cores': rates.get(phys),
            'threads_swept': {str(k): round(v, 1) for k, v in rates.items()}, 'chunks_per_worker_by_k': {str(k): v for k, v in pers.items()},
            'physical_cores': phys, 'logical_cpus': logical, 'cpu': cpu_model, 'chunk_samples': chunk, 'chunks_per_worker': per, 'workers': mode,
            'sample': 'same model, 1 utterance x %d samples per forward (%d chunks of %d per worker), %d samples in %.1f s wall on %d workers; torch-CPU fp32 '
                      'restatement of the reference math (oracle/torch_cpu.py, conditioning per sample as written) evaluated in time chunks with the '
                      'flow\'s look-back recomputed, one chunk per core at a time (%s pool, started and warmed before the clock); sweep over k in %s '
                      'on the same kind of sample (%d chunks per worker), best = %d; value_1thread is the k = 1 point'
                      % (chunk * per * best, per, chunk, length, dt, best, mode, sorted(rates), per, best)}


def cpu_baseline(case, case_cfg, target_s):
    """The measurement above in a CLEAN child process (no HIP runtime, no OpenMP team: its fork()ed worker pool is safe there and
    free of the GIL); if the child cannot be had, the same measurement in this process on a thread pool."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-child', '--case', case, '--cpu-seconds', str(target_s)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=max(180.0, 12 * target_s),
                           env=dict(os.environ, HIP_VISIBLE_DEVICES='', OMP_NUM_THREADS='1'))
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        sys.stderr.write('cpu_baseline child failed (rc %d): %s\n' % (r.returncode, r.stderr[-400:]))
    except Exception as e:
        sys.stderr.write('cpu_baseline child failed (%s: %s)\n' % (type(e).__name__, e))
    return cpu_baseline_measure(case_cfg, target_s, mode='thread')


def model_algorithmic_work(hp, elem_bytes):
    """SURVEY.md section 8d, the whole-model "layer-streaming" figures: every dilated layer reads its residual stream once
    and writes it once (halo re-reads, weights and anything a fusion keeps on chip not counted); FLOPs with the
    conditioning projection hoisted to frame rate in 'repeat' mode; only live ops (skip on the last layer only, no dense
    on the last layer -- what `use_skip_connection: False` executes)."""
    m = hp.model
    W, R, D, S, C = m.filter_width, m.residual_channels, m.dilation_channels, m.skip_channels, m.condition_channels
    hop, n_mels = hp.signal.hop_length, hp.signal.n_mels
    shared = bool(m.get('shared_nets', False))
    Q = 2 if shared else 1
    nets = []
    for i in range(m.n_iaf):
        nets += [len(m.dilations[i])] * (1 if shared else 2)
    method = m.cond_upsample_method
    b = elem_bytes
    cond_b = {'repeat': 2.0 * D * b / hop, 'transposed_conv': float(C * b)}.get(method, 0.0)
    nbytes = sum((L - 1) * 2 * R * b + 2 * b + L * cond_b for L in nets)
    per_sample_cond = 2 * C * D if method == 'transposed_conv' else 0
    mac = 0
    for L in nets:
        skip_all = bool(m.use_skip_connection)
        layers = sum(2 * W * R * D + per_sample_cond + (D * S if (j == L - 1 or skip_all) else 0) + (D * R if j < L - 1 or skip_all else 0)
                     for j in range(L))
        mac += W * 1 * R + layers + S * S + S * Q
    mac += {'repeat': n_mels * C / hop, 'transposed_conv': 8000.0}.get(method, 0.0)
    return nbytes, 2.0 * mac


def merged_length(intervals):
    """Total length of the union of [begin, end] intervals."""
    total, cur_b, cur_e = 0.0, None, None
    for b, e in sorted(intervals):
        if cur_e is None or b > cur_e:
            if cur_e is not None:
                total += cur_e - cur_b
            cur_b, cur_e = b, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        total += cur_e - cur_b
    return total


def latest_profile_json(suffix):
    """Newest profiles/rNN_*<suffix> (the rocprofv3 summaries tools/profile_round.sh commits)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_*' + suffix)))
    return c[-1] if c else None


def attach_profile(roof, args, rows):
    """`traffic` (HBM bytes per launch from the PMC passes) and the rocprofv3-timed launch duration of the dominant kernel come
    from the COMMITTED profile set of this configuration (profiles/rNN_x_<case>[_<precision>]_hbm_traffic.json, written by
    tools/profile_round4_summarize.py from the same command under rocprofv3 on the builder's box) -- PMC passes cannot run inside
    the timed region.  Everything that is not measured by THIS run sits under `committed_profile` and says so
    (`measured_in_run: false`); `traffic` itself, which the contract wants in the roofline object, is tagged the same way."""
    name = args.case.replace('bench/', '') + ('' if args.precision == 'f16x3' else '_' + args.precision)
    tpath = latest_profile_json('_%s_hbm_traffic.json' % name)
    if not tpath:
        return
    with open(tpath) as f:
        tj = json.load(f)
    if tj.get('rows') != rows or tj.get('kernel_pattern', '').split('<')[0] not in roof['kernel']:
        return
    roof['traffic'] = tj['traffic_bytes_per_launch']
    roof['traffic_measured_in_run'] = False
    cp = {'measured_in_run': False, 'source': os.path.relpath(tpath, ROOT),
          'box': tj.get('box', "the builder's gpurun box of that profile round (one MI355X), not the box of this run"),
          'command': tj.get('command'), 'traffic_bytes_per_launch': tj['traffic_bytes_per_launch'],
          'algorithmic_bytes_per_launch': tj.get('algorithmic_bytes_per_launch'), 'traffic_over_algorithmic': tj['ratio']}
    if 'frac_rocprof' in tj:
        cp['rocprof_kernel_us'] = tj['rocprof_kernel_us']
        cp['frac_rocprof'] = tj['frac_rocprof']      # (concurrent launches x) algorithmic bytes per launch / the rocprofv3 average / 8 TB/s
    roof['committed_profile'] = cp


def main():
    args = parse_args()
    if args.cpu_baseline_child:
        from oracle.iaf_oracle import ModelConfig
        from pwv_amd.hparam import hparam as hp_
        hp_.set_hparam_yaml(args.case)
        print(json.dumps(cpu_baseline_measure(ModelConfig.from_hparam(hp_), args.cpu_seconds, mode='process')))
        return
    # PWV_BENCH_DRYRUN=control (test hook, tests/test_host_logic.py): no GPU work at all -- the launcher / rendezvous / barrier /
    # max-over-ranks / rank-0 JSON control flow with a no-op step, over gloo on CPU.
    # PWV_BENCH_DRYRUN_ONE_GPU=1 (test hook, tests/test_gpu_unfused_and_e2e.py): the real step, but all ranks share GPU 0 and
    # rendezvous over gloo, to exercise the multi-rank path on a 1-GPU box.
    # PWV_BENCH_FORCE_DIST=1 (de-risks the driver's multi-GPU run on a 1-GPU box): `--gpus 1` initialises the `nccl` backend
    # (= RCCL) at world size 1 and takes EVERY distributed branch below -- barriers, the MAX all-reduce of the elapsed time on
    # a device tensor, the give-up vote, generate_sharded / generate_time_sharded_ranks over nccl scatter / gather.
    control = os.environ.get('PWV_BENCH_DRYRUN') == 'control'
    dryrun = control or os.environ.get('PWV_BENCH_DRYRUN_ONE_GPU') == '1'
    force_dist = os.environ.get('PWV_BENCH_FORCE_DIST') == '1'
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d\n' % (args.gpus, world, world))
    if not control and not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback exists in the product path)')
    if control:
        dev = torch.device('cpu')
    else:
        dev_index = 0 if dryrun else local_rank
        torch.cuda.set_device(dev_index)
        dev = torch.device('cuda', dev_index)
    dist = None
    backend = None
    if world > 1 or force_dist:
        if world == 1 and 'MASTER_PORT' not in os.environ:      # (no launcher: rendezvous with ourselves)
            import socket
            sk = socket.socket()
            sk.bind(('127.0.0.1', 0))
            os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
            sk.close()
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        # keep stdout to the one JSON line.  RCCL writes its debug stream to STDOUT by default: the version banner
        # ("RCCL version : ..." at NCCL_DEBUG >= VERSION) and NCCL WARN lines (seen on the GPU box: "alt_rsmi.cc NCCL WARN Could
        # not read node # 9" at init AND at teardown, glued to whatever was printed last).  Leave NCCL_DEBUG as the caller set it
        # (unset = silent) and send the stream to stderr; the JSON line is printed after the process group is gone (below).
        # (Only when stderr is a tty / pipe: RCCL opens the file for writing, which would truncate a regular file that `2> log`
        # points at and clobber what the ranks wrote before -- there the JSON line printed last is the protection.)
        try:
            import stat
            stderr_is_file = stat.S_ISREG(os.fstat(2).st_mode)
        except OSError:
            stderr_is_file = True
        if not stderr_is_file:
            os.environ.setdefault('NCCL_DEBUG_FILE', '/dev/stderr')
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = 'gloo' if dryrun else 'nccl'                              # 'nccl' IS RCCL on ROCm (xGMI)
        if dryrun:
            dist.init_process_group(backend='gloo')
        else:
            dist.init_process_group(backend='nccl', device_id=dev)
    n_gpus = world

    from pwv_amd.hparam import hparam as hp
    hp.set_hparam_yaml(args.case)
    length = args.length or hp.generate.length
    utts = args.utts or hp.generate.batch_size
    if args.case == 'bench/c4' and not args.utts:
        utts = max(1, hp.generate.batch_size // 8)       # 64 utterances over 8 GPUs: 8 per GPU
    hop, n_mels = hp.signal.hop_length, hp.signal.n_mels
    job_length = length                  # samples per utterance of the JOB
    time_shard = None
    if args.shard == 'time':
        # every rank computes its slice [a, b) of the time axis of the SAME utterance(s), preceded by the flow chain's look-back
        # (recomputed and discarded): exact, no data-path collective; the job's work is fixed, so this is STRONG scaling
        from pwv_amd.timeshard import chain_halo, shard_plan
        halo = chain_halo(hp.model.dilations, hp.model.filter_width, hp.model.n_iaf, hop)
        plan = shard_plan(length, world, halo, hop)
        c0, a, b = plan[min(rank, len(plan) - 1)]
        time_shard = {'halo': halo, 'window': b - c0, 'share': b - a, 'shards': len(plan)}
        length = b - c0                  # what this rank's forward runs on
    t_mel = 1 + length // hop
    rows = utts * length

    def sync_all():
        if not control:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not control:
            torch.cuda.synchronize()

    g = torch.Generator(device='cpu').manual_seed(1000 + rank)
    mel = (torch.rand((utts, t_mel, n_mels), generator=g) * 2 - 1).to(dev)
    if control:
        def make_step(precision):
            return (lambda: torch.zeros((utts, length, 1))), None, False
    else:
        from pwv_amd import _lib, engine
        from pwv_amd.models import IAFVocoder
        from pwv_amd.variables import VariableStore
        if local_rank == 0:
            _lib.build_library()          # one builder per node; the others wait (no concurrent hipcc on one .so)
        if dist is not None:
            dist.barrier()
        # synthetic, seeded per rank: mel ~ U(-1,1); random-init weights (glorot + N(0,0.1) biases) so bias paths run
        store = VariableStore(device=dev, seed=2)
        model0 = IAFVocoder(batch_size=utts, length=length, store=store, precision=args.precision)
        model0.noise_seed = 1 + (0 if time_shard else rank)
        model0(None, mel, is_training=False, verify=False)            # creates the variables
        for name in list(store.vars):
            if store.vars[name].dim() == 1:
                store.vars[name].normal_(0, 0.1, generator=None)
        store.version += 1
        engine.clear_plan_cache()

        def make_step(precision):
            """(step, eager_step, graphed?) for the model in `precision`; one step = fresh logistic noise + the forward"""
            model = model0 if precision == args.precision else IAFVocoder(batch_size=utts, length=length, store=store, precision=precision)
            model.noise_seed = 1 + rank

            def eager_step():      # enqueue-only, like the C ABI: the timed loop must not synchronise; model0.verify() follows it
                return model(None, mel, is_training=False, verify=False)

            if args.no_graph:
                return eager_step, eager_step, False
            # the same launches, captured once into a HIP graph (pwv_amd/graph.py); each step = ONE graph replay: the sampler is a
            # node of the graph (fresh logistic noise every step, pwv_logistic_noise_stream_f32), the mel sits in the graph's input buffer
            from pwv_amd.graph import GraphedVocoder
            try:
                graphed = GraphedVocoder(model)
                # self-check (untimed): a replay must reproduce the host-enqueued forward bit for bit on the same noise
                zc = engine.logistic_noise_op((utts, length, 1), dev, seed=12345)
                want = model(None, mel, is_training=False, z=zc, verify=False).clone()
                got = graphed(mel, z=zc)
                torch.cuda.synchronize()
                if not torch.equal(want, got):
                    raise RuntimeError('HIP-graph replay differs from the host-enqueued forward')
                del want, got, zc
                # the mel is resident in HBM before the timed region starts: it sits in the graph's own input buffer, so a step is
                # the noise kernel + the replay (no device-to-device copy of an input that does not change)
                graphed.mel.copy_(mel)
                torch.cuda.synchronize()
                return (lambda: graphed(graphed.mel)), eager_step, True
            except Exception as e:      # never lose the measurement to a capture problem: same launches, host-enqueued
                sys.stderr.write('graph capture / replay check failed (%s: %s); falling back to host-enqueued launches\n' % (type(e).__name__, e))
                torch.cuda.synchronize()
                return eager_step, eager_step, False

    def timed(step, warmup, steps):
        out = None
        for _ in range(warmup):
            out = step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        sync_all()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if dryrun else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, out

    def timed_verified(precision, warmup, steps):
        """make_step + timed + the give-up / range check of what was timed.  A persistent launch that gave up (its workgroups
        were not all resident: another process on this GPU, e.g. the two-ranks-on-one-GPU test hook) has switched the engine to
        the per-layer launches -- never lose the measurement to it: rebuild the step and time again.  With several ranks the
        decision is taken TOGETHER (one MAX all-reduce), and a give-up never leaves the timed loop early, so that the ranks'
        barriers stay paired."""
        for attempt in (0, 1):
            failed = []
            try:
                made = make_step(precision)
            except _lib.PwvPersistError as e:
                failed.append(e)
                made = make_step(precision)          # (the engine is on the per-layer path now)
            last = [None]

            def guarded():
                try:
                    last[0] = made[0]()
                except _lib.PwvPersistError as e:
                    failed.append(e)
                return last[0]

            elapsed_, out_ = timed(guarded, warmup, steps)
            if not control:
                try:
                    model0.verify()      # range guard of the split-fp16 arithmetic / give-up word of the persistent launch
                except _lib.PwvPersistError as e:
                    failed.append(e)
            again = 1 if failed else 0
            if dist is not None:
                t = torch.tensor([again], dtype=torch.int32, device='cpu' if dryrun else dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                again = int(t.item())
            if not again:
                return made, elapsed_, out_
            if not control:
                # bench policy: once any rank has seen a give-up, the whole measurement runs on per-layer launches (the engine's own
                # policy -- suspend, retry after 16 forwards -- would switch launch paths in the middle of the timed loop)
                engine.PERSIST = False
            if attempt:
                raise failed[0] if failed else _lib.PwvPersistError('a persistent launch gave up on another rank, twice')
            sys.stderr.write('%s\nre-timing with per-layer launches\n' % (failed[0] if failed else 'a persistent launch gave up on another rank'))

    (step, eager_step, graphed), elapsed, out = timed_verified(args.precision, args.warmup, args.steps)
    assert torch.isfinite(out).all(), 'non-finite output'

    # ---- the same workload in the reference's own arithmetic (exact fp32 MFMA), timed the same way ------------------
    f32_exact = None
    if args.precision == 'f16x3' and not args.no_f32_exact and not control:
        n32 = max(3, min(args.steps, 10))
        (step32, _, graphed32), e32, out32 = timed_verified('f32', 2, n32)
        assert torch.isfinite(out32).all()
        f32_exact = (e32, n32, graphed32)
        del step32, out32

    # ---- the JOB as north_star words it, over the job's backend (N > 1, or PWV_BENCH_FORCE_DIST=1): all mels on rank 0 -> scatter
    # (utterance shards) / broadcast of slices (time shards) -> every rank's verified forward -> gather of the waveforms on rank 0.
    # One untimed call, then the median of 5 timed ones, each between barriers -- `job.job_samples_per_s` next to the weak-scaling
    # `value` (whose timed loop is forward-only: every rank on its own resident mel)
    sharded, job = None, None
    JOB_CALLS = 5
    job_errors = []          # a forward that fails on ONE rank must not strand the others in the gather: zeros out, collectives stay paired,
                             # and the failure bit below stops every rank (generate._generate_over_ranks does the same)

    def guarded_fwd(f, shape_of):
        def g(*a):
            try:
                return f(*a)
            except Exception as e:      # noqa: BLE001
                job_errors.append(e)
                return torch.zeros(shape_of(*a), dtype=torch.float32, device=a[0].device)
        return g

    def raise_if_any_rank_failed():
        bad = 1 if job_errors else 0
        if dist is not None:
            t = torch.tensor([bad], dtype=torch.int32, device='cpu' if dryrun else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bad = int(t.item())
        if bad:
            raise job_errors[0] if job_errors else RuntimeError('bench.py: the sharded forward failed on another rank')

    def timed_job(call):
        call()                                   # (untimed: model objects for the shard shapes, plans, RCCL channels)
        times = []
        for _ in range(JOB_CALLS):
            sync_all()
            t0 = time.perf_counter()
            w_ = call()
            sync_all()
            times.append(time.perf_counter() - t0)
        if dist is not None:
            t = torch.tensor(times, dtype=torch.float64, device='cpu' if dryrun else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            times = [float(v) for v in t.tolist()]
        return w_, sorted(times)[len(times) // 2], times

    if dist is not None and time_shard is not None:
        from pwv_amd.distributed import generate_time_sharded_ranks
        coll_dev = torch.device('cpu') if dryrun else dev
        jt = 1 + job_length // hop
        full_mel = (torch.rand((utts, jt, n_mels), generator=torch.Generator().manual_seed(7)) * 2 - 1).to(coll_dev) if rank == 0 else None
        if control:
            fwd = lambda m, zz, t0: torch.zeros((m.shape[0], (m.shape[1] - 1) * hop, 1))
        else:
            nets_by_window = {}

            def fwd(m, zz, t0):
                win = (m.shape[1] - 1) * hop
                if win not in nets_by_window:
                    nets_by_window[win] = model0 if win == length else IAFVocoder(batch_size=m.shape[0], length=win, store=store, precision=args.precision)
                zw = engine.logistic_noise_window(m.shape[0], job_length, t0, win, dev, 4242)
                return nets_by_window[win](None, m.to(dev), is_training=False, z=zw, verify=True).to(coll_dev)      # (a verified call: rerun inside if a sticky word is raised)
        fwd = guarded_fwd(fwd, lambda m, zz, t0: (m.shape[0], (m.shape[1] - 1) * hop, 1))
        wav, job_s, job_times = timed_job(lambda: generate_time_sharded_ranks(fwd, full_mel, n_mels, job_length, hop, time_shard['halo'], coll_dev))
        raise_if_any_rank_failed()
        job_samples = utts * job_length
        if rank == 0:
            assert tuple(wav.shape) == (utts, job_length, 1) and bool(torch.isfinite(wav).all())
            sharded = '%d utterance(s) x %d samples cut into %d time shards (look-back %d samples), generated, gathered on rank 0: ok' % (
                utts, job_length, time_shard['shards'], time_shard['halo'])
            job = {'what': 'distributed.generate_time_sharded_ranks: mel on rank 0 -> slices to the ranks -> verified forward per rank -> waveform gathered on rank 0',
                   'samples': job_samples}
    elif dist is not None:
        from pwv_amd.distributed import generate_sharded
        total = utts * world
        coll_dev = torch.device('cpu') if dryrun else dev        # gloo (test hooks) scatters / gathers host tensors only
        full_mel = (torch.rand((total, t_mel, n_mels), generator=torch.Generator().manual_seed(7)) * 2 - 1).to(coll_dev) if rank == 0 else None
        if control:
            fwd = lambda m, zz: torch.zeros((m.shape[0], length, 1))
        else:
            nets_by_batch = {}

            def fwd(m, zz):
                nb = m.shape[0]
                if nb not in nets_by_batch:
                    nets_by_batch[nb] = model0 if nb == utts else IAFVocoder(batch_size=nb, length=length, store=store, precision=args.precision)
                return nets_by_batch[nb](None, m.to(dev), is_training=False, z=zz, verify=True).to(coll_dev)      # (a verified call)
        fwd = guarded_fwd(fwd, lambda m, zz: (m.shape[0], length, 1))
        wav, job_s, job_times = timed_job(lambda: generate_sharded(fwd, full_mel, (t_mel, n_mels), length, coll_dev))
        raise_if_any_rank_failed()
        job_samples = total * length
        if rank == 0:
            assert tuple(wav.shape) == (total, length, 1) and bool(torch.isfinite(wav).all())
            sharded = '%d utterances scattered over %d ranks, generated, gathered on rank 0: ok' % (total, world)
            job = {'what': 'distributed.generate_sharded: all mels on rank 0 -> scatter -> verified forward per rank (noise sampled on the device) -> waveforms gathered on rank 0',
                   'samples': job_samples}
    if job is not None:
        job.update({'job_samples_per_s': job_samples / job_s, 'job_ms_median': job_s * 1e3, 'job_ms_all': [round(v * 1e3, 4) for v in job_times],
                    'calls': JOB_CALLS, 'launch': 'host-enqueued, verified calls (each waits for its launches and reads the sticky words); max over ranks per call',
                    'note': 'the job-level number: inputs start and results end on rank 0; `value` above is the forward-only weak-scaling metric of SURVEY.md section 8d'})

    # ---- live kernel timing of the dominant kernel (HIP events on the launch streams) --------------------------------
    timing = None
    if not control:
        for attempt in (0, 1):
            engine.EVENT_LOG = []
            ref = torch.cuda.Event(enable_timing=True)
            ref.record()
            try:
                n_event_fwd = max(1, min(args.steps, 5))
                for _ in range(n_event_fwd):
                    eager_step()
                torch.cuda.synchronize()
                engine.raise_if_persist_failed()
                break
            except _lib.PwvPersistError as e:      # (see timed_verified)
                if attempt:
                    raise
                sys.stderr.write('%s\nre-timing the launches on the per-layer path\n' % e)
        log, engine.EVENT_LOG = engine.EVENT_LOG, None
        # entries: ('layer_residual', ...) one chain's run of `cnt` back-to-back per-layer launches between two HIP events;
        #          ('persist', ...) ONE persistent launch covering `cnt` layers of `gnets` nets
        pers = [(ref.elapsed_time(en[1]), ref.elapsed_time(en[2]), en[4], en[3]) for en in log if en[0] == 'persist']
        first_runs = sum(en[3] for en in log if en[0] == 'persist' and len(en) > 5 and en[5])      # net-layers that read 4 B instead of 256 B per sample
        tail_runs = sum(en[3] for en in log if en[0] == 'persist' and len(en) > 6 and en[6])       # nets whose LAST layer + head ran inside the launch
        chains = [(ref.elapsed_time(en[1]), ref.elapsed_time(en[2]), en[4], en[3]) for en in log if en[0] == 'layer_residual']
        if pers:
            # a forward has a few distinct launches (9- and 29-layer runs); take the MEDIAN duration of each kind over the timed
            # forwards (the events also span the host's enqueue of the control-word kernel: a late enqueue is not kernel time)
            kinds = {}
            for b, e, cnt, gnets in pers:
                kinds.setdefault((cnt, gnets), []).append(e - b)
            med = {k: sorted(v)[len(v) // 2] for k, v in kinds.items()}
            total_ms = sum(med[(cnt, gnets)] for _, _, cnt, gnets in pers)
            timing = {'kind': 'persist', 'forwards': n_event_fwd, 'launches': len(pers), 'total_ms': total_ms, 'net_layers': sum(cnt * g for _, _, cnt, g in pers),
                      'launch_ms': [min(med.values()), max(med.values())],
                      'layers_per_launch': sorted(set(cnt for _, _, cnt, _ in pers)), 'nets_per_launch': pers[0][3], 'first_net_layers': first_runs,
                      'tail_net_layers': tail_runs}
        elif chains:
            busy = merged_length([(b, e) for b, e, _, _ in chains])
            total_ms = sum(e - b for b, e, _, _ in chains)
            launches = sum(cnt for _, _, cnt, _ in chains)
            timing = {'kind': 'per-layer', 'forwards': n_event_fwd, 'layer_ms': total_ms / launches, 'launches': launches, 'nets_per_launch': chains[0][3],
                      'overlap': total_ms / busy if busy > 0 else 1.0}

    if rank == 0:
        nets_per_flow = 1 if bool(hp.model.get('shared_nets', False)) else 2
        # utterance shards: every GPU its own rows (weak scaling); time shards: ONE job of utts x job_length samples (strong scaling)
        total_samples = (utts * job_length if time_shard else rows * n_gpus) * args.steps
        value = total_samples / elapsed
        n_layers = sum(len(d) for d in hp.model.dilations[:hp.model.n_iaf])
        n_nets = hp.model.n_iaf * nets_per_flow
        result = {
            'metric': 'audio samples/sec, 4-flow IAF generation',
            'value': value,
            'unit': 'samples/s',
            'n_gpus': n_gpus,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'strong' if time_shard else 'weak',
            'vs_baseline': None,
            'dtype': {'f32': 'f32', 'f16x3': 'f32 via 3-term split-fp16 MFMA (fp32 accumulate, fp32 storage)',
                      'f16': 'f16 (REDUCED PRECISION extension: fp16 residual stream + fp16 MFMA, fp32 accumulate)'}[args.precision],
            'precision': args.precision,
            'parity': 'REDUCED PRECISION: max|y - y_fp64| <= 5e-3 (tests/test_gpu_f16.py); not comparable with the fp32 headline'
                      if args.precision == 'f16' else
                      'max|y - y_fp64| <= 2e-5 (measured ~3e-6 on the full model for both f32 and f16x3; tests/ -m gpu)',
            'data': 'synthetic',
            'x_realtime_22050': value / 22050.0,
            'x_realtime_16000': value / 16000.0,
            'config': {
                'workload': '%s: %d IAF flows, %d WaveNets (%d dilated layers, R=D=64, S=128, W=2), %s conditioning, '
                            '%d utterance(s) x %d samples per GPU (%.1f s @16 kHz; %d mel frames x %d mels, hop %d)'
                            % (args.case, hp.model.n_iaf, n_nets, n_layers * nets_per_flow, hp.model.cond_upsample_method,
                               utts, length, length / 16000.0, t_mel, n_mels, hop),
                'case': args.case, 'utterances_per_gpu': utts, 'samples_per_utterance': length,
                'parallelism': ('time-sharded x%d: every GPU one slice of the SAME %d x %d samples plus %d samples of recomputed look-back (exact, no data-path collective)'
                                % (n_gpus, utts, job_length, time_shard['halo'])) if time_shard else 'utterance-sharded x%d (no data-path collective)' % n_gpus,
                'noise': 'logistic, sampled on device inside the step',
                'launch': 'HIP graph replay of the forward: ONE graph launch per step (the logistic sampler is a node of the graph and draws a fresh counter range every replay; the mel resident in the graph\'s input buffer)' if graphed else 'host-enqueued launches',
            },
        }
        if dist is not None:
            result['rccl_ranks'] = world if backend == 'nccl' else 0
            result['backend'] = backend
            result['sharded_generate'] = sharded
            result['job'] = job
        if control:
            result['dryrun'] = 'control flow only: no kernels ran, value is meaningless (PWV_BENCH_DRYRUN=control)'
        mb, mf = model_algorithmic_work(hp, 2 if args.precision == 'f16' else 4)
        layer_bytes = LAYER_BYTES_PER_SAMPLE // 2 if args.precision == 'f16' else LAYER_BYTES_PER_SAMPLE
        if hp.model.cond_upsample_method == 'transposed_conv':
            # per-sample condition: every layer also reads the [rows, C] condition (fp32 tile32 / fp16 hi+lo planes: 4 B per
            # channel; fp16 mode: 2 B) -- SURVEY.md section 8d, cond_b = C * b
            layer_bytes += int(hp.model.condition_channels) * (2 if args.precision == 'f16' else 4)
        skip_all = bool(hp.model.use_skip_connection)
        layer_flop = LAYER_FLOP_PER_SAMPLE + (2 * 64 * 128 if skip_all else 0)      # (skip 64 -> 128 in EVERY layer, modules.py:243-250)
        if skip_all:
            # the skip sum [rows, 128] fp32 is read and written by every layer (modules.py:147: the sum over all layers' skip outputs)
            layer_bytes += 2 * 128 * 4
        if timing is not None and timing['kind'] == 'persist':
            # the dominant kernel is the persistent stack kernel: one launch runs `layers` residual layers of both nets of a flow
            # over all timed launches; a run that starts with the net's layer 0 reads 4 B per sample there instead of a 256 B row
            # ... and (round 5) the net's LAST layer with the head behind it rides in the same launch: it reads the residual stream once
            # (half a layer's bytes) and writes Q floats per sample; 2 x (2*64*128 + 64*128 + 128*128 + 128*Q) FLOP per sample
            q_out = 2 if bool(hp.model.get('shared_nets', False)) else 1
            tail_bytes = layer_bytes // 2 + 4 * q_out
            tail_flop = 2 * (2 * 64 * 128 + 64 * 128 + 128 * 128 + 128 * q_out)
            alg_bytes = rows * (timing['net_layers'] * layer_bytes - timing['first_net_layers'] * (layer_bytes // 2 - 4) + timing['tail_net_layers'] * tail_bytes)
            alg_flop = rows * (timing['net_layers'] * LAYER_FLOP_PER_SAMPLE + timing['tail_net_layers'] * tail_flop)
            ach_gbs = alg_bytes / (timing['total_ms'] * 1e-3) / 1e9
            ach_tf = alg_flop / (timing['total_ms'] * 1e-3) / 1e12
            arith = {'f16x3': 'split-fp16 MFMA', 'f32': 'exact fp32 MFMA', 'f16': 'fp16 rows + fp16 MFMA'}[args.precision]
            roof = {'kernel': 'stack_persist_kernel (persistent dataflow launch: %s residual layers%s x %d nets per launch, %s)'
                              % ('/'.join(str(c) for c in timing['layers_per_launch']),
                                 ' + last layer + head + IAF affine' if timing['tail_net_layers'] else '', timing['nets_per_launch'], arith),
                    'bound': 'hbm', 'achieved': ach_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': ach_gbs / PEAK_HBM_GBS,
                    'launches_timed': timing['launches'], 'launch_ms_median_by_kind': timing['launch_ms'],
                    'us_per_layer_pair': timing['total_ms'] * 1e3 / (timing['net_layers'] / timing['nets_per_launch']),
                    'alg_bytes_per_net_layer': rows * layer_bytes, 'alg_flop_per_net_layer': rows * LAYER_FLOP_PER_SAMPLE,
                    # what ONE forward launches of this kernel (tools/profile_round4_summarize.py prices the PMC / rocprof passes with it)
                    'launches_per_forward': timing['launches'] // timing['forwards'], 'net_layers_per_forward': timing['net_layers'] // timing['forwards'],
                    'first_net_layers_per_forward': timing['first_net_layers'] // timing['forwards'],
                    'tail_net_layers_per_forward': timing['tail_net_layers'] // timing['forwards'], 'alg_bytes_per_tail_net_layer': rows * tail_bytes,
                    'alg_bytes_per_forward': alg_bytes // timing['forwards'],
                    'traffic': None,
                    'note': 'achieved = algorithmic bytes of the layers a launch runs (512 B per sample, net and residual layer; 260 B for a folded '
                            'layer 0; 256 B + 4 Q for the last layer + head when they run inside the launch) / its duration, '
                            'HIP events around each persistent launch on its stream (median per kind of launch over the timed forwards)'}
            attach_profile(roof, args, rows)
            cond_flop = 2 * int(hp.model.condition_channels) * 128 if hp.model.cond_upsample_method == 'transposed_conv' else 0
            if args.precision == 'f32':
                # exact-fp32 MFMA: 80 FLOP/B >> fp32 machine balance (19.7) => the matrix pipe is the roof, the HBM view sits beside it
                result['roofline'] = dict(roof, bound='mfma', achieved=ach_tf, peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s', frac=ach_tf / PEAK_F32_MFMA_TFLOPS)
                result['roofline_hbm'] = {'bound': 'hbm', 'achieved': ach_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': ach_gbs / PEAK_HBM_GBS}
            elif args.precision == 'f16':
                result['roofline'] = roof
                tf16 = rows * timing['net_layers'] * (LAYER_FLOP_PER_SAMPLE + cond_flop) / (timing['total_ms'] * 1e-3) / 1e12
                result['roofline_mfma'] = {'bound': 'mfma', 'achieved': tf16, 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s (one fp16 product per term)',
                                           'frac': tf16 / PEAK_F16_MFMA_TFLOPS}
            else:
                result['roofline'] = roof
                # issued: three fp16 MFMAs per algorithmic one -- except in a net's layer 0 when it runs folded onto its four input
                # scalars (engine.FOLD_FIRST): ONE k-step of 16 for the filter|gate convolution instead of eight
                folded = engine.FOLD_FIRST and timing['first_net_layers'] > 0
                first_issued = 3 * 2 * (16 * 128 + 64 * 64) if folded else 3 * LAYER_FLOP_PER_SAMPLE
                issued_tf = rows * ((timing['net_layers'] - timing['first_net_layers']) * 3 * LAYER_FLOP_PER_SAMPLE
                                    + timing['first_net_layers'] * first_issued
                                    + timing['tail_net_layers'] * 3 * (tail_flop - 2 * 128 * q_out)) / (timing['total_ms'] * 1e-3) / 1e12
                roof['first_layer'] = 'folded onto its four input scalars (one MFMA k-step for filter|gate)' if folded else 'as every other layer'
                roof['limiter'] = LIMITER_K1P
                result['roofline_mfma'] = {'bound': 'mfma', 'achieved': issued_tf, 'peak': PEAK_F16_MFMA_TFLOPS,
                                           'unit': 'TFLOP/s (fp16 MFMA FLOPs issued: 3x algorithmic, a folded layer 0 less)', 'frac': issued_tf / PEAK_F16_MFMA_TFLOPS}
        elif timing is not None:
            layer_ms, nets_per_launch = timing['layer_ms'], timing['nets_per_launch']
            per_sample = 1 if hp.model.cond_upsample_method == 'transposed_conv' else 0
            # scalar / shifter chains on two streams: the measured overlap of the chains' busy intervals (sum / union) says
            # how many launches of this kernel share the chip on average
            concurrent = timing['overlap']
            flop_per_launch = rows * nets_per_launch * layer_flop
            bytes_per_launch = rows * nets_per_launch * layer_bytes
            ach_tf = concurrent * flop_per_launch / (layer_ms * 1e-3) / 1e12
            ach_gbs = concurrent * bytes_per_launch / (layer_ms * 1e-3) / 1e9
            common = {'avg_launch_ms': layer_ms, 'launches_timed': timing['launches'], 'alg_flop_per_launch': flop_per_launch,
                      'alg_bytes_per_launch': bytes_per_launch, 'traffic': None,
                      'concurrent_launches': concurrent,
                      'note': 'achieved = concurrent_launches x algorithmic work per launch / avg launch duration; concurrent_launches is '
                              'MEASURED: sum of the chains\' event intervals / length of their union (the scalar and shifter chains of a '
                              'flow run side by side on two HIP streams)' if nets_per_flow // nets_per_launch > 1
                              else 'one launch covers all nets of the flow'}
            if args.precision == 'f32':
                # exact-fp32 MFMA: 80 FLOP/B >> fp32 machine balance (19.7) => matrix-pipe bound
                result['roofline'] = dict(kernel='layer_f32_kernel (fused gated-residual layer, %d nets/launch)' % nets_per_launch,
                                          bound='mfma', achieved=ach_tf, peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s',
                                          frac=ach_tf / PEAK_F32_MFMA_TFLOPS, **common)
                result['roofline_hbm'] = {'bound': 'hbm', 'achieved': ach_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                          'frac': ach_gbs / PEAK_HBM_GBS}
            elif args.precision == 'f16':
                # fp16 rows: 160 fp16-FLOP/B < fp16 machine balance (312) => HBM bound
                result['roofline'] = dict(kernel='layer_h16_kernel<%d,0> (fused gated-residual layer, fp16 rows%s, %d nets/launch)' % (per_sample, ', per-sample condition' if per_sample else '', nets_per_launch),
                                          bound='hbm', achieved=ach_gbs, peak=PEAK_HBM_GBS, unit='GB/s',
                                          frac=ach_gbs / PEAK_HBM_GBS, **common)
                result['roofline_mfma'] = {'bound': 'mfma', 'achieved': ach_tf, 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                           'frac': ach_tf / PEAK_F16_MFMA_TFLOPS}
            else:
                # split-fp16 MFMA: 3 x 80 = 240 fp16-FLOP/B < fp16 machine balance (312) => HBM bound
                result['roofline'] = dict(kernel='layer_f16x3_kernel<%d,%d,0> (fused gated-residual layer%s%s, %d nets/launch)' % (int(skip_all), per_sample, ', per-sample condition' if per_sample else '',
                                                 ', skip sum read-modify-written in every layer (1024 B per sample on top of the 512)' if skip_all else '', nets_per_launch),
                                          bound='hbm', achieved=ach_gbs, peak=PEAK_HBM_GBS, unit='GB/s',
                                          frac=ach_gbs / PEAK_HBM_GBS, **common)
                attach_profile(result['roofline'], args, rows)
                result['roofline']['limiter'] = LIMITER_K1P
                result['roofline_mfma'] = {'bound': 'mfma', 'achieved': 3 * ach_tf, 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s (fp16 MFMA FLOPs issued = 3x algorithmic)',
                                           'frac': 3 * ach_tf / PEAK_F16_MFMA_TFLOPS}
        per_gpu = rows * args.steps / elapsed if time_shard else value / n_gpus      # (time shards: the samples a GPU actually computes, look-back included)
        result['model'] = {
            'note': 'whole-model algorithmic work per output sample (SURVEY.md section 8d "layer-streaming" model), per GPU',
            'alg_bytes_per_sample': mb, 'alg_flop_per_sample': mf,
            'hbm_GBs': mb * per_gpu / 1e9, 'hbm_frac_of_8TBs': mb * per_gpu / 1e9 / PEAK_HBM_GBS,
            'alg_TFLOPs': mf * per_gpu / 1e12,
            'vs_fp32_fma_ceiling': per_gpu / (PEAK_F32_MFMA_TFLOPS * 1e12 / mf),
        }
        if f32_exact is not None:
            e32, n32, g32 = f32_exact
            v32 = rows * n_gpus * n32 / e32
            result['f32_exact'] = {
                'note': "the same workload with precision='f32': v_mfma_f32_32x32x2_f32, the reference's own arithmetic (models.py:81-82); "
                        'matrix-pipe bound, so its roofline is the fp32 MFMA peak (157.3 TFLOP/s)',
                'value': v32, 'unit': 'samples/s', 'ms_per_step': e32 / n32 * 1e3, 'steps': n32,
                'alg_TFLOPs': mf * v32 / n_gpus / 1e12, 'mfma_frac': mf * v32 / n_gpus / 1e12 / PEAK_F32_MFMA_TFLOPS,
                'launch': 'HIP graph replay' if g32 else 'host-enqueued launches'}
        if not control and args.precision == 'f16x3':
            # how far this workload stays from the range guard of the split-fp16 arithmetic (untimed, one eager forward with the
            # run-time maxima read back): limit / observed per operand class, `range_margin` = the smallest (engine.range_report)
            try:
                zr = engine.logistic_noise_op((utts, length, 1), dev, seed=4242)
                rr = engine.range_report(lambda: model0(None, mel, is_training=False, z=zr, verify=False))
                model0.verify()
                result['range_guard'] = {'range_margin': rr['range_margin'],
                                         'classes': {k: {kk: float('%.6g' % vv) for kk, vv in c.items()} for k, c in rr['classes'].items()},
                                         'note': 'limit / observed (or bounded) maximum per fp16 operand class of this workload; below 1 the forward '
                                                 'would be rerun in exact fp32 (f32_exact is that path)'}
            except Exception as e:
                sys.stderr.write('range report failed (%s: %s)\n' % (type(e).__name__, e))
        if n_gpus == 1 and not control and not dryrun and args.precision == 'f16x3' and not args.no_f32_exact and length > 16000 and not args.length:
            # the SAME model on one SHORT utterance (16000 samples: the latency-bound end of the path, where the persistent launch takes its
            # short-input instantiation -- csrc/pwv_stack_persist.hip, DESIGN.md section 4 "Short inputs"), untimed by the contract, reported beside it
            try:
                ls = 16000
                mel_s = (torch.rand((1, 1 + ls // hop, n_mels), generator=torch.Generator().manual_seed(11)) * 2 - 1).to(dev)
                from pwv_amd.graph import GraphedVocoder
                ms_model = IAFVocoder(batch_size=1, length=ls, store=store, precision='f16x3')
                ms_model.noise_seed = 3
                ms_model(None, mel_s, is_training=False, verify=False)
                gs = GraphedVocoder(ms_model)
                gs.mel.copy_(mel_s)
                for _ in range(5):
                    gs(gs.mel)
                torch.cuda.synchronize()
                t0s = time.perf_counter()
                for _ in range(50):
                    os_ = gs(gs.mel)
                torch.cuda.synchronize()
                es = (time.perf_counter() - t0s) / 50
                ms_model.verify()
                assert torch.isfinite(os_).all()
                result['short_input'] = {'workload': 'the same model, 1 utterance x %d samples, HIP graph replay (not part of `value`)' % ls,
                                         'ms_per_step': es * 1e3, 'samples_per_s': ls / es, 'steps': 50,
                                         'hbm_frac_of_8TBs': mb * ls / es / 1e9 / PEAK_HBM_GBS}
            except Exception as e:
                sys.stderr.write('short-input leg failed (%s: %s)\n' % (type(e).__name__, e))
        if n_gpus == 1 and not args.no_cpu_baseline and not control:
            from oracle.iaf_oracle import ModelConfig          # the oracle is only ever the CPU leg, never the timed path
            result['cpu_baseline'] = cpu_baseline(args.case, ModelConfig.from_hparam(hp), args.cpu_seconds)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the LAST thing on stdout, on a line of its own whatever a library printed before it without a newline
        sys.stdout.flush()
        sys.stdout.write(('\n' if dist is not None else '') + json.dumps(result) + '\n')
        sys.stdout.flush()


if __name__ == '__main__':
    main()
