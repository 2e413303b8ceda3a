"""-m gpu: the persistent kernel for the residual layers of a stack (csrc/pwv_stack_persist.hip) against the per-layer
launches -- BIT-identical by construction (same per-unit arithmetic; only the schedule, the buffers and the synchronisation
differ), on shapes that exercise every protocol path: the production regime (40 units per workgroup: the x[t-d] look-back
of d = 512 reaches the left neighbour's top 16 units), ranges of one to four units (the look-back crosses up to 17
workgroups, waves run several layers apart: weight refills, progress words and the 3-buffer ring all matter), one net and
two, several utterances per batch (x[t-d] = 0 at utterance starts inside a range), 30-layer stacks, stacks cut into several
runs that hand the ring on, and the whole model under HIP-graph replay.  Reference loop: modules.py:138-143."""
import ctypes

import numpy as np
import pytest

from oracle import iaf_oracle as O
from tests.util import TOL_F32, set_hparams

pytestmark = pytest.mark.gpu

D10 = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]


def _nets(gpu, L, G, seed=3, cond_channels=80):
    import torch
    from pwv_amd import engine
    from pwv_amd.modules import WaveNet
    from pwv_amd.variables import VariableStore
    store = VariableStore(device=gpu, seed=seed)
    kw = dict(batch_size=1, dilations=(D10 * 3)[:L], filter_width=2, residual_channels=64, dilation_channels=64, skip_channels=128,
              quantization_channels=1, use_biases=True, condition_channels=cond_channels, use_skip_connection=False, is_training=False, store=store)
    nets = [WaveNet(name='n%d' % g, **kw) for g in range(G)]
    return store, nets


@pytest.fixture()
def persist_knobs():
    from pwv_amd import engine
    saved = (engine.PERSIST, engine.PERSIST_MIN_UNITS, engine.PERSIST_MAX_LAYERS, engine.FOLD_FIRST, engine.FUSE_TAIL)
    engine.resume_persist()
    yield engine
    engine.PERSIST, engine.PERSIST_MIN_UNITS, engine.PERSIST_MAX_LAYERS, engine.FOLD_FIRST, engine.FUSE_TAIL = saved
    engine.resume_persist()


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
@pytest.mark.parametrize('n,t,L,G,min_units,max_layers', [
    (1, 160000, 10, 2, 0, 32), (1, 16000, 10, 2, 0, 32), (1, 8000, 30, 2, 0, 32), (1, 64000, 30, 1, 0, 32), (1, 2400, 6, 2, 0, 32),
    (3, 8000, 10, 2, 0, 32), (5, 1040, 7, 1, 0, 32), (1, 65536, 4, 2, 0, 32),
    (1, 16000, 12, 2, 1, 32),       # one unit per workgroup: every look-back is a neighbour's, d = 512 reaches 16 workgroups back
    (2, 24000, 30, 2, 2, 32),       # two units per workgroup over 30 layers, two utterances
    (1, 160000, 30, 2, 0, 10),      # 28 residual layers as three runs (10 + 9 + 9) that hand the ring on
    (1, 48000, 13, 1, 0, 4),        # 11 residual layers as runs of 4 + 4 + 3: every ring rotation
    (1, 96, 5, 2, 0, 32),           # three units in all
    # round 6, the short-input instantiation (per-unit progress words, stationary units, a loader wave): 5 and 7 units per workgroup, its upper end
    # (8 units per workgroup: the general kernel again), utterance starts inside units, one net on all CUs
    (1, 20000, 10, 2, 0, 32), (1, 28000, 12, 2, 0, 32), (1, 32000, 10, 2, 0, 32), (2, 14001, 10, 2, 0, 32), (3, 10000, 15, 1, 0, 32)])
def test_persistent_stack_is_bit_identical_to_per_layer_launches(gpu, persist_knobs, n, t, L, G, min_units, max_layers, precision):
    import torch
    engine = persist_knobs
    store, nets = _nets(gpu, L, G)
    g = torch.Generator().manual_seed(n * 7 + L)
    x = torch.randn((n, t, 1), generator=g).to(gpu)
    frames = torch.rand((n, t // 80 + 1, 80), generator=g).to(gpu)
    cond = engine.RepeatedCondition(frames, 80, 40, t)
    engine.run_nets(nets, x, cond, precision=precision)  # creates the variables
    for name in list(store.vars):
        if store.vars[name].dim() == 1:
            store.vars[name].normal_(0, 0.1)
    store.version += 1
    engine.PERSIST = False
    ref = [o.clone() for o in engine.run_nets(nets, x, cond, precision=precision)]
    engine.PERSIST, engine.PERSIST_MIN_UNITS, engine.PERSIST_MAX_LAYERS = True, min_units, max_layers      # forced, whatever the size
    log = engine.EVENT_LOG = []
    try:
        for _ in range(3):
            got = engine.run_nets(nets, x, cond, precision=precision)
            torch.cuda.synchronize()
            assert engine.persist_status() == 0
            for a, b in zip(ref, got):
                assert torch.equal(a, b)
    finally:
        engine.EVENT_LOG = None
    runs = engine._persist_runs(L, 0)       # (scalar-input nets: the launch starts with the net's layer 0)
    assert [e[0] for e in log] == ['persist'] * (3 * len(runs)) and sum(e[4] for e in log) == 3 * (L - 1)      # it really was persistent
    # ... and which instantiation it was: at most 7 units of 32 rows per workgroup and layer is the short-input one (256 CUs assumed)
    from pwv_amd import _lib
    cus = _lib.lib().pwv_device_cus()
    if cus == 256 and (n, t, G, min_units) in ((1, 16000, 2, 0), (1, 20000, 2, 0), (1, 28000, 2, 0), (2, 14001, 2, 0), (3, 10000, 1, 0), (1, 16000, 2, 1)):
        assert all(e[7] == 1 for e in log)
    if cus == 256 and (n, t, G) in ((1, 160000, 2), (1, 32000, 2), (1, 65536, 2)):
        assert all(e[7] == 0 for e in log)


@pytest.mark.parametrize('n,t,L,G,Q,min_units,max_layers', [
    (1, 160000, 10, 2, 1, 0, 32), (1, 16000, 30, 2, 1, 0, 32), (3, 8000, 10, 2, 1, 0, 32), (1, 16000, 12, 2, 1, 1, 32),
    (1, 48000, 13, 2, 1, 0, 4),       # runs of 4 + 4 + 3 layers: the tail rides on the LAST run
    (1, 96, 5, 2, 1, 0, 32), (1, 40, 4, 2, 1, 0, 32),       # three units / two units in all (a ragged last unit)
    (2, 24000, 10, 1, 2, 0, 32), (1, 2400, 6, 1, 2, 2, 32)])        # one shared net with two outputs (BASELINE config 2): the affine in place
@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
def test_tail_and_affine_inside_the_launch_are_bit_identical_to_the_separate_launches(gpu, persist_knobs, n, t, L, G, Q, min_units, max_layers, precision):
    """Round 5: the net's last layer + head (modules.py:145-165) and the flow's affine (modules.py:59) run INSIDE the persistent
    launch (pwv_persist_args.tail_*: a flow is one launch instead of three).  Same operations in the same order as
    layer_f16x3_kernel<..., HEAD> and the affine kernel: the nets' outputs and the flow's output agree bit for bit with the
    separate launches (FUSE_TAIL off) and with the per-layer path -- three times in a row on one workspace (the pair counters
    and progress words are left clean), eager.  Round 6: in the exact-fp32 arithmetic too (the operations of layer_f32_kernel<8, ..., HEAD>):
    the path a range flag reruns on is one launch per flow as well."""
    import torch
    from pwv_amd.modules import WaveNet
    from pwv_amd.variables import VariableStore
    engine = persist_knobs
    store = VariableStore(device=gpu, seed=5)
    kw = dict(batch_size=n, dilations=(D10 * 3)[:L], filter_width=2, residual_channels=64, dilation_channels=64, skip_channels=128,
              quantization_channels=Q, use_biases=True, condition_channels=80, use_skip_connection=False, is_training=False, store=store)
    if Q == 2:
        kw['input_channels'] = 1          # (the shared net of BASELINE config 2: one input, scale and shift out; models.py here builds it so)
    nets = [WaveNet(name='n%d' % g, **kw) for g in range(G)]
    g = torch.Generator().manual_seed(n * 13 + L)
    x = torch.randn((n, t, 1), generator=g).to(gpu)
    cond = engine.RepeatedCondition(torch.rand((n, t // 80 + 1, 80), generator=g).to(gpu), 80, 40, t)
    engine.run_nets(nets, x, cond, precision=precision)  # creates the variables
    for name in list(store.vars):
        if store.vars[name].dim() == 1:
            store.vars[name].normal_(0, 0.1)
    store.version += 1
    engine.PERSIST = False
    ref_outs = [o.clone() for o in engine.run_nets(nets, x, cond, precision=precision)]
    ref_flow = engine.run_flow(nets, x, cond, precision=precision).clone()
    engine.PERSIST, engine.PERSIST_MIN_UNITS, engine.PERSIST_MAX_LAYERS = True, min_units, max_layers
    engine.FUSE_TAIL = False
    sep_flow = engine.run_flow(nets, x, cond, precision=precision).clone()
    torch.cuda.synchronize()
    assert engine.persist_status() == 0 and torch.equal(sep_flow, ref_flow)
    engine.FUSE_TAIL = True
    log = engine.EVENT_LOG = []
    try:
        for _ in range(3):
            outs = engine.run_nets(nets, x, cond, precision=precision)
            flow = engine.run_flow(nets, x, cond, precision=precision)
            torch.cuda.synchronize()
            assert engine.persist_status() == 0
            for a, b in zip(ref_outs, outs):
                assert torch.equal(a, b)
            assert torch.equal(flow, ref_flow)
    finally:
        engine.EVENT_LOG = None
    runs = engine._persist_runs(L, 0)
    assert len(log) == 6 * len(runs) and sum(e[6] for e in log) == 6      # every forward's last run carried the tail


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
@pytest.mark.parametrize('n,t,L,G,cond', [(1, 16000, 10, 2, 'frames'), (3, 2400, 6, 1, 'frames'), (2, 4000, 5, 2, 'none'), (1, 1234, 3, 2, 'samples')])
def test_folded_layer0_is_the_same_function_on_every_path(gpu, persist_knobs, n, t, L, G, cond, precision):
    """Scalar-input nets (split-fp16 and fp32 paths): layer 0's filter|gate convolution folded onto the four scalars it is a function
    of (pwv_pack_first_fold_f16x3 / _f32; the default) -- the persistent launch and the per-layer FIRST kernels (also with a
    per-sample condition) perform the same operations, so they agree bit for bit, and the folded result stays within the
    path's tolerance of the unfolded one (h[t] = x[t-1] w0 + x[t] w1 evaluated first, modules.py:179-180)."""
    import torch
    engine = persist_knobs
    store, nets = _nets(gpu, L, G, cond_channels=None if cond == 'none' else 80)
    g = torch.Generator().manual_seed(n * 11 + L)
    x = torch.randn((n, t, 1), generator=g).to(gpu)
    if cond == 'frames':
        c = engine.RepeatedCondition(torch.rand((n, t // 80 + 1, 80), generator=g).to(gpu), 80, 40, t)
    elif cond == 'samples':
        c = (torch.rand((n, t, 80), generator=g) * 2 - 1).to(gpu)
    else:
        c = None
    engine.run_nets(nets, x, c, precision=precision)  # creates the variables
    for name in list(store.vars):
        if store.vars[name].dim() == 1:
            store.vars[name].normal_(0, 0.1)
    store.version += 1
    engine.FOLD_FIRST = True
    engine.run_nets(nets, x, c, precision=precision)       # (the plans are made here, with the folded fragments)
    res = {}
    for fold in (False, True):
        for persist in (False, True):
            engine.FOLD_FIRST, engine.PERSIST = fold, persist
            res[fold, persist] = [o.clone() for o in engine.run_nets(nets, x, c, precision=precision)]
            torch.cuda.synchronize()
            assert engine.persist_status() == 0
    mode = 'none' if c is None else ('frames' if cond == 'frames' else 'samples')
    assert all(engine.get_plan(net, mode, engine.PRECISIONS[precision]).first_fold is not None for net in nets)
    for k in range(G):
        assert torch.equal(res[False, False][k], res[False, True][k]) and torch.equal(res[True, False][k], res[True, True][k])
        a, b = res[False, False][k], res[True, False][k]
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max()))
        assert not torch.equal(a, b)           # (it really was another evaluation order)


def test_whole_model_persistent_eager_and_graph_replay(gpu, persist_knobs):
    """The reference-default model (4 flows, 8 nets, 10/10/10/30 layers) at the headline size: per-layer launches,
    persistent launches and persistent launches replayed from a HIP graph give the same bits; and they match the fp64
    oracle on a prefix."""
    import torch
    from pwv_amd.graph import GraphedVocoder
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    engine = persist_knobs
    cfg = O.ModelConfig()
    set_hparams(cfg)
    w = O.init_weights(cfg, seed=2)
    store = VariableStore(device=gpu)
    store.load_dict(w)
    L = 160000
    mel, z = O.synthetic_inputs(1, L, cfg)
    mel_t, z_t = torch.from_numpy(mel).to(gpu), torch.from_numpy(z).to(gpu)
    model = IAFVocoder(batch_size=1, length=L, store=store)
    engine.PERSIST = False
    y0 = model(None, mel_t, is_training=False, z=z_t).clone()
    engine.PERSIST = True
    y1 = model(None, mel_t, is_training=False, z=z_t).clone()
    model.verify()
    assert torch.equal(y0, y1)
    graphed = GraphedVocoder(model)
    for _ in range(3):
        y2 = graphed(mel_t, z=z_t)
        model.verify()
        assert torch.equal(y0, y2)
    K = 4000
    want = O.iaf_vocoder_forward(w, mel[:, :K // 80 + 1], z[:, :K], cfg)
    assert np.abs(y1.cpu().numpy()[:, :K - 40] - want[:, :K - 40]).max() <= TOL_F32


def test_persistent_give_up_is_loud_and_falls_back(gpu, persist_knobs):
    """The kernel reports a give-up (unexpected placement, a poll that ran into its bound) through a sticky word in pinned
    host memory; the host then raises and SUSPENDS the persistent launches: per-layer launches for the next
    PERSIST_RETRY_AFTER forwards, then another try (a give-up is a condition of the GPU -- a co-tenant -- not of this process).
    The range word goes with it (garbage downstream may have raised it; the rerun must not meet it).  (The word is poked from the
    host here: a real give-up needs a second process on the GPU, tools/co_tenant_check.sh.)"""
    from pwv_amd._lib import PwvPersistError
    engine = persist_knobs
    engine.PERSIST = True
    assert engine.persist_status() == 0 and not engine.persist_suspended()
    engine.poke_persist_status(4)
    engine.current_words().range = 1
    with pytest.raises(PwvPersistError, match='gave up'):
        engine.raise_if_persist_failed()
    assert engine.PERSIST is True and engine.persist_suspended() and engine.persist_status() == 0 and not engine.range_flag_raised()
    assert not engine._use_persist(2, 1, 16000, [1, 2, 4, 8, 16, 32], 0)
    for _ in range(engine.PERSIST_RETRY_AFTER):
        engine.note_forward()
    assert not engine.persist_suspended() and engine._use_persist(2, 1, 16000, [1, 2, 4, 8, 16, 32], 0)


def test_auto_policy_takes_the_persistent_launch_where_the_library_supports_the_shape(gpu, persist_knobs):
    """PWV_PERSIST=auto (the default): the persistent launch wherever the library supports the shape, up to
    PERSIST_AUTO_MAX_ROWS rows per launch; shapes it refuses (stacks of fewer than four layers; a look-back over more
    neighbours than one wave instruction polls) silently take the per-layer path."""
    engine = persist_knobs
    engine.PERSIST = 'auto'
    assert engine._use_persist(2, 1, 16000, D10) and engine._use_persist(1, 3, 16000, D10 * 3) and engine._use_persist(2, 1, 160000, D10)
    assert not engine._use_persist(2, 1, 16000, D10[:3])
    saved = engine.PERSIST_AUTO_MAX_ROWS
    engine.PERSIST_AUTO_MAX_ROWS = 72000
    try:
        assert engine._use_persist(2, 1, 64000, D10) and not engine._use_persist(2, 1, 160000, D10)
    finally:
        engine.PERSIST_AUTO_MAX_ROWS = saved
    engine.PERSIST, engine.PERSIST_MIN_UNITS = True, 1
    assert engine._use_persist(2, 1, 160000, D10)
    assert not engine._use_persist(2, 1, 3200, [1, 2, 4, 4096, 8, 16])       # 129 units back at one unit per workgroup
    engine.PERSIST = False
    assert not engine._use_persist(2, 1, 16000, D10)


def test_a_shorter_callers_struct_is_not_read_past_its_end(gpu, persist_knobs):
    """pwv_persist_args grew by appended fields.  A client compiled against an earlier minor version passes a SHORTER struct
    (struct_size says how short); what lies behind it in the caller's memory -- here: a wild status pointer, tail_q = 3 with wild tail pointers,
    a wild affine_x -- must be treated as zero.  The launch then is the plain run of residual layers, bit-identical to the per-layer path."""
    import torch
    from pwv_amd import _lib
    engine = persist_knobs
    store, nets = _nets(gpu, 10, 2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 16000, 1), generator=g).to(gpu)
    cond = engine.RepeatedCondition(torch.rand((1, 201, 80), generator=g).to(gpu), 80, 40, 16000)
    engine.PERSIST = False
    ref = [o.clone() for o in engine.run_nets(nets, x, cond, precision='f16x3')]
    engine.PERSIST, engine.FUSE_TAIL = True, False      # (the engine itself does not ask for the tail: the fields are free for garbage)
    seen = []

    def truncate(pa):
        pa.struct_size = _lib.PersistArgs.status.offset      # the struct as it was before the status word, the tail and the affine were appended
        pa.status = 0xdead0000
        pa.tail_q, pa.tail_dilation = 3, 7
        for k in range(2):
            pa.tail_layer[k] = pa.tail_head[k] = pa.tail_out[k] = 0xdead0000
        pa.affine_x = pa.affine_out = 0xdead0000
        seen.append(pa.struct_size)

    engine.PERSIST_ARGS_HOOK = truncate
    try:
        got = engine.run_nets(nets, x, cond, precision='f16x3')
        torch.cuda.synchronize()
    finally:
        engine.PERSIST_ARGS_HOOK = None
    assert seen and engine.persist_status() == 0
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


def test_a_tail_that_reaches_too_far_runs_as_its_own_launch(gpu, persist_knobs):
    """the probe behind `_use_persist` asks about the launch that will be made, tail included.  A stack whose LAST dilation is
    its largest (2048 samples = 64 units, one unit per workgroup: 64 workgroups back, the poll reaches 60) keeps its persistent layers and
    runs last layer + head as a launch of their own -- it used to pass the probe and fail at the launch."""
    import torch
    from pwv_amd.modules import WaveNet
    from pwv_amd.variables import VariableStore
    engine = persist_knobs
    store = VariableStore(device=gpu, seed=4)
    kw = dict(batch_size=1, dilations=[1, 2, 4, 8, 16, 2048], filter_width=2, residual_channels=64, dilation_channels=64, skip_channels=128,
              quantization_channels=1, use_biases=True, condition_channels=80, use_skip_connection=False, is_training=False, store=store)
    nets = [WaveNet(name='n%d' % k, **kw) for k in range(2)]
    g = torch.Generator().manual_seed(6)
    x = torch.randn((1, 4000, 1), generator=g).to(gpu)      # 125 units on 128 workgroups per net: one unit each
    cond = engine.RepeatedCondition(torch.rand((1, 51, 80), generator=g).to(gpu), 80, 40, 4000)
    engine.PERSIST = False
    ref = [o.clone() for o in engine.run_nets(nets, x, cond, precision='f16x3')]
    engine.PERSIST, engine.PERSIST_MIN_UNITS = True, 1
    assert not engine._use_persist(2, 1, 4000, kw['dilations'], 0, tail_q=1) and engine._use_persist(2, 1, 4000, kw['dilations'], 0, tail_q=0)
    log = engine.EVENT_LOG = []
    try:
        got = engine.run_nets(nets, x, cond, precision='f16x3')
        torch.cuda.synchronize()
    finally:
        engine.EVENT_LOG = None
    assert engine.persist_status() == 0 and [e[0] for e in log] == ['persist'] and log[0][6] == 0      # persistent layers, no tail in the launch
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


def test_random_shapes_through_the_persistent_launch_under_load(gpu):
    """Round 6: the protocol's stress test (tools/persist_fuzz.py) -- 80 random (utterances, length, layers, nets, units per workgroup, layers per run,
    arithmetic) through the persistent launch, four times each on one workspace, against the per-layer launches bit for bit, with a second stream of
    unrelated memory traffic beside it (hand-off bugs hide on an idle chip): lengths that are no multiple of 32, utterance starts inside units, one to
    seven units per workgroup (the short-input instantiation: per-unit progress words, stationary units, the loader wave) and beyond (the general one)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'persist_fuzz.py'), '--cases', '80', '--seed', '5', '--load'],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    assert ' 0 failed' in r.stdout
