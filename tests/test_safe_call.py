"""The contract of the reference-shaped calls (engine.verified_call): a call returns a VERIFIED result by default.

CPU part: the control flow with the device taken out (stubs for the stream / the sticky words).
-m gpu part: `pred = model(None, mel, is_training=False); pred.cpu()` -- generate.py:38,68 -- with a mel far outside the
range of the split-fp16 arithmetic gives the exact-fp32 result, never inf; a persistent give-up inside the call is repaired
inside the call; verify=False hands the duty to verify()."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import iaf_oracle as O
from tests.util import TOL_F32, set_hparams, small_cfg


class _Stream:
    def __init__(self, log):
        self.log = log

    def synchronize(self):
        self.log.append('sync')


@pytest.fixture()
def stub_engine(monkeypatch):
    from pwv_amd import engine
    log = []
    state = {'persist': 0, 'range': False}
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda *a, **k: _Stream(log))
    monkeypatch.setattr(torch.cuda, 'is_current_stream_capturing', lambda: False)
    monkeypatch.setattr(engine, 'persist_status', lambda words=None: state['persist'])
    monkeypatch.setattr(engine, 'clear_persist_status', lambda words=None: state.__setitem__('persist', 0))
    monkeypatch.setattr(engine, 'range_flag_raised', lambda words=None: state['range'])
    monkeypatch.setattr(engine, 'clear_range_flag', lambda words=None: state.__setitem__('range', False))
    monkeypatch.setattr(engine, 'raise_if_persist_failed', lambda words=None: None)
    monkeypatch.setattr(engine, 'ASYNC', False)
    saved = engine.PERSIST
    engine.resume_persist()
    yield engine, log, state
    engine.PERSIST = saved
    engine.resume_persist()


def test_clean_call_synchronises_once_and_returns(stub_engine):
    engine, log, state = stub_engine
    runs = []
    out = engine.verified_call(lambda prec: (runs.append(prec), 'y')[1])
    assert out == 'y' and runs == [None] and log == ['sync']


def test_give_up_inside_the_call_is_rerun_on_per_layer_launches(stub_engine):
    engine, log, state = stub_engine
    engine.PERSIST = 'auto'
    runs = []

    def run(prec):
        runs.append((prec, engine.persist_suspended()))
        if len(runs) == 1:
            state['persist'], state['range'] = 4, True      # the give-up, and garbage downstream tripping the range guard
        return 'y%d' % len(runs)
    with pytest.warns(UserWarning, match='rerun on per-layer launches'):
        assert engine.verified_call(run) == 'y2'
    assert runs == [(None, False), (None, True)] and engine.PERSIST == 'auto' and not state['range']


def test_a_give_up_suspends_the_persistent_launches_and_they_come_back(stub_engine):
    """one give-up (a transient co-tenant) must not cost 5 % for the life of the process.  The engine stays on
    per-layer launches for PERSIST_RETRY_AFTER forwards, then tries the persistent launch again; a second give-up in a row
    doubles the pause, a forward that goes through resets it."""
    engine, log, state = stub_engine
    engine.PERSIST = 'auto'
    seen = []

    def run(prec):
        seen.append(engine.persist_suspended())
        return 'y'

    def give_up_once(prec):
        if not engine.persist_suspended():
            state['persist'] = 4
        return run(prec)
    with pytest.warns(UserWarning, match='per-layer launches'):
        engine.verified_call(give_up_once)
    k = engine.PERSIST_RETRY_AFTER
    for _ in range(k):
        engine.verified_call(run)
    # the call with the give-up: [persistent, rerun suspended]; the first k - 1 forwards after it are suspended (the rerun consumed
    # no tick, the k-th tick ends the pause at the START of the k-th forward)
    assert seen[:2] == [False, True] and seen[2:2 + k - 1] == [True] * (k - 1) and seen[2 + k - 1] is False
    assert not engine.persist_suspended() and engine._persist_backoff == k          # ... and a clean forward reset the back-off
    del seen[:]
    with pytest.warns(UserWarning):
        engine.verified_call(give_up_once)
    for _ in range(k - 1):
        engine.verified_call(run)
    with pytest.warns(UserWarning):
        engine.verified_call(give_up_once)          # gives up again on its first try after the pause
    assert engine._persist_cooldown == 2 * k        # consecutive give-ups: the pause doubles


def test_range_flag_inside_the_call_is_rerun_in_f32(stub_engine):
    engine, log, state = stub_engine
    runs = []

    def run(prec):
        runs.append(prec)
        if prec is None:
            state['range'] = True
        return prec
    with pytest.warns(UserWarning, match='rerun in exact fp32'):
        assert engine.verified_call(run) == 'f32'
    assert runs == [None, 'f32'] and log == ['sync', 'sync']


def test_a_rerun_that_still_fails_raises(stub_engine):
    from pwv_amd._lib import PwvRangeError
    engine, log, state = stub_engine

    def run(prec):
        state['range'] = True          # e.g. a NaN in the input: no arithmetic repairs that
        return prec
    with pytest.raises(PwvRangeError), pytest.warns(UserWarning):
        engine.verified_call(run)


def test_nested_async_and_opted_out_calls_only_enqueue(stub_engine):
    engine, log, state = stub_engine
    inner = []

    def outer(prec):
        inner.append(engine.verified_call(lambda p: 'inner'))      # e.g. the IAF layers inside IAFVocoder.__call__
        assert log == []                                           # ... no synchronisation per flow
        return 'outer'
    assert engine.verified_call(outer) == 'outer' and inner == ['inner'] and log == ['sync']
    del log[:]
    state['range'] = True
    assert engine.verified_call(lambda p: 'raw', verify=False) == 'raw' and log == [] and state['range']
    engine.ASYNC = True
    assert engine.verified_call(lambda p: 'raw') == 'raw' and log == []
    with pytest.warns(UserWarning):
        assert engine.verified_call(lambda p: p, verify=True) == 'f32'      # an explicit verify=True wins over PWV_ASYNC


# ---- on the device ----------------------------------------------------------------------------------------------------
def _model(gpu, cfg, length, n, precision=None, seed=6):
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    set_hparams(cfg)
    store = VariableStore(device=gpu)
    store.load_dict(O.init_weights(cfg, seed=seed))
    return IAFVocoder(batch_size=n, length=length, store=store, precision=precision), store


@pytest.mark.gpu
@pytest.mark.parametrize('method', ['repeat', 'transposed_conv'])
def test_reference_shaped_call_never_returns_inf(gpu, method):
    """generate.py:38,68 as written -- no z, no verify() -- with |mel| ~ 1e5: the split-fp16 forward trips the range guard,
    the call reruns itself in exact fp32 on the same noise and returns what a precision='f32' model returns, bit for bit."""
    from pwv_amd import engine
    from pwv_amd.models import IAFVocoder
    cfg = small_cfg(cond_upsample_method=method)
    n, length = 2, 480
    mel, _ = O.synthetic_inputs(n, length, cfg)
    mel_t = torch.from_numpy((mel * 1e5).astype(np.float32)).to(gpu)
    model, store = _model(gpu, cfg, length, n)
    model.noise_seed = 77
    with pytest.warns(UserWarning, match='rerun in exact fp32'):      # (the repair is not silent)
        pred = model(None, mel_t, is_training=False)
    got = pred.cpu()
    assert bool(torch.isfinite(got).all())
    assert not engine.range_flag_raised() and engine.persist_status() == 0      # nothing left behind for the next caller
    z = engine.logistic_noise_op((n, length, 1), gpu, seed=77, offset=0)         # the noise that call drew
    m32 = IAFVocoder(batch_size=n, length=length, store=store, precision='f32')
    want = m32(None, mel_t, is_training=False, z=z)
    assert torch.equal(got, want.cpu())
    with np.errstate(all='ignore'):
        ref = O.iaf_vocoder_forward(O.init_weights(cfg, seed=6), (mel * 1e5).astype(np.float32), z.cpu().numpy(), cfg)
    assert np.abs(got.numpy() - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max())      # (fp32 itself, on saturated gates)
    # in range the same model object answers in its own arithmetic again
    mel1 = torch.from_numpy(mel).to(gpu)
    y = model(None, mel1, is_training=False, z=z).cpu().numpy()
    assert np.abs(y - O.iaf_vocoder_forward(O.init_weights(cfg, seed=6), mel, z.cpu().numpy(), cfg)).max() <= TOL_F32


@pytest.mark.gpu
def test_opting_out_hands_the_duty_to_verify(gpu):
    from pwv_amd._lib import PwvRangeError
    cfg = small_cfg()
    mel, z = O.synthetic_inputs(1, 480, cfg)
    model, _ = _model(gpu, cfg, 480, 1)
    bad = torch.from_numpy((mel * 1e5).astype(np.float32)).to(gpu)
    model(None, bad, is_training=False, z=torch.from_numpy(z).to(gpu), verify=False)      # returns at once, unverified
    with pytest.raises(PwvRangeError):
        model.verify()
    model.verify()                                                                         # reported once, then clean


@pytest.mark.gpu
def test_give_up_inside_a_call_is_repaired_inside_the_call(gpu, monkeypatch):
    """A persistent launch that gives up during THIS call (the word is poked from the host while the call is enqueuing: a real
    give-up needs a second process on the GPU, tools/co_tenant_check.sh) -> the call reruns on per-layer launches and hands
    back exactly what the per-layer path computes."""
    from pwv_amd import engine
    cfg = O.ModelConfig(dilations=[[1, 2, 4, 8, 16, 32], [1, 2, 4, 8, 16, 32]], n_iaf=2)
    n, length = 1, 16000
    mel, z = O.synthetic_inputs(n, length, cfg)
    mel_t, z_t = torch.from_numpy(mel).to(gpu), torch.from_numpy(z).to(gpu)
    model, _ = _model(gpu, cfg, length, n, seed=2)
    saved = engine.PERSIST
    try:
        engine.PERSIST = False
        want = model(None, mel_t, is_training=False, z=z_t).clone()
        engine.PERSIST = True
        real, poked = engine._run_stack_persist, []

        def spy(*a, **k):
            r = real(*a, **k)
            if not poked:
                poked.append(1)
                engine.poke_persist_status(4)
            return r
        monkeypatch.setattr(engine, '_run_stack_persist', spy)
        with pytest.warns(UserWarning, match='per-layer launches'):
            got = model(None, mel_t, is_training=False, z=z_t)
        assert poked and engine.persist_suspended() and engine.PERSIST is True and engine.persist_status() == 0
        assert torch.equal(got, want)
    finally:
        engine.PERSIST = saved
        engine.resume_persist()


@pytest.mark.gpu
def test_wavenet_and_iaf_layer_called_directly_are_safe_too(gpu):
    """modules.WaveNet / LinearIAFLayer used on their own (generate.py:11-13 imports them): a flow input beyond the range of
    the split-fp16 arithmetic comes back as the exact-fp32 result, not inf."""
    from pwv_amd.modules import LinearIAFLayer, WaveNet
    from pwv_amd.variables import VariableStore, variable_scope
    store = VariableStore(device=gpu)
    kw = dict(batch_size=1, dilations=[1, 2, 4, 8], filter_width=2, residual_channels=64, dilation_channels=64, skip_channels=128,
              quantization_channels=1, use_biases=True, condition_channels=None, use_skip_connection=False, store=store)
    with variable_scope('t'):
        sc, sh = WaveNet(name='scalar', **kw), WaveNet(name='shifter', **kw)
        sc32, sh32 = WaveNet(name='scalar', precision='f32', **kw), WaveNet(name='shifter', precision='f32', **kw)
    x = (torch.randn((1, 640, 1), generator=torch.Generator().manual_seed(3)) * 3e5).to(gpu)
    with pytest.warns(UserWarning, match='exact fp32'):
        y = sc(x)
    assert bool(torch.isfinite(y).all()) and torch.equal(y, sc32(x))
    with pytest.warns(UserWarning, match='exact fp32'):
        out = LinearIAFLayer(1, sc, sh)(x)
    assert bool(torch.isfinite(out).all()) and torch.equal(out, LinearIAFLayer(1, sc32, sh32)(x))


@pytest.mark.gpu
def test_a_parity_test_cannot_pass_on_the_calls_own_repair(gpu):
    """tests/test_gpu_fullsize.py::_model / test_gpu_golden.py::_full_model call ``model(...)`` in its default,
    self-repairing form; an f16x3 forward that tripped the range guard would be rerun in exact fp32 and PASS an f16x3 parity
    check with nothing but a UserWarning.  tests/conftest.py::_repairs_are_failures turns every ``pwv:`` warning inside a
    ``-m gpu`` test into an exception -- this test is such a full-size-style check with the guard forced, and it fails (the
    ``pytest.raises`` below finds no exception and the comparison would succeed on the fp32 bits) if that fixture is removed."""
    from pwv_amd import engine
    cfg = small_cfg()
    n, length = 1, 960
    mel, z = O.synthetic_inputs(n, length, cfg)
    model, _ = _model(gpu, cfg, length, n, precision='f16x3')
    bad = torch.from_numpy((mel * 1e5).astype(np.float32)).to(gpu)
    with pytest.raises(UserWarning, match='pwv: an input or activation left the exponent range'):
        model(None, bad, is_training=False, z=torch.from_numpy(z).to(gpu))
    engine.clear_range_flag()              # (the exception left the call before its own clean-up of the rerun)
    engine.clear_persist_status()
    torch.cuda.synchronize()
    # and the same model, in range, answers in its own arithmetic without a warning
    y = model(None, torch.from_numpy(mel).to(gpu), is_training=False, z=torch.from_numpy(z).to(gpu)).cpu().numpy()
    assert np.abs(y - O.iaf_vocoder_forward(O.init_weights(cfg, seed=6), mel, z, cfg)).max() <= TOL_F32


@pytest.mark.gpu
@pytest.mark.allow_pwv_repair
def test_two_threads_on_two_streams_do_not_see_each_others_flags(gpu):
    """the sticky words are per THREAD (engine.current_words), so a thread whose forwards trip the
    range guard (and are repaired in exact fp32, with a warning) cannot make another thread's in-range split-fp16 forward rerun --
    or, worse, consume that thread's flag.  Thread B's results are the split-fp16 bits of a single-threaded run, every time."""
    import threading
    import warnings
    from pwv_amd import engine
    from pwv_amd.models import IAFVocoder
    cfg = small_cfg()
    n, length = 1, 960
    mel, z = O.synthetic_inputs(n, length, cfg)
    mel_t, z_t = torch.from_numpy(mel).to(gpu), torch.from_numpy(z).to(gpu)
    bad_t = torch.from_numpy((mel * 1e5).astype(np.float32)).to(gpu)
    model_a, store_a = _model(gpu, cfg, length, n)
    model_b, _ = _model(gpu, cfg, length, n)
    want_b = model_b(None, mel_t, is_training=False, z=z_t).clone()                      # split-fp16, single-threaded
    m32 = IAFVocoder(batch_size=n, length=length, store=store_a, precision='f32')
    want_a = m32(None, bad_t, is_training=False, z=z_t).clone()                          # what A's repaired calls must return
    with pytest.warns(UserWarning, match='exact fp32'):
        assert torch.equal(model_a(None, bad_t, is_training=False, z=z_t), want_a)       # (single-threaded: it does)
    main_words = engine.current_words().addr
    res = {'a': [], 'b': [], 'words': [], 'err': []}

    def worker(name, model, inp, rounds):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=gpu)):
                res['words'].append(engine.current_words().addr)
                for _ in range(rounds):
                    res[name].append(model(None, inp, is_training=False, z=z_t).clone())
                torch.cuda.current_stream().synchronize()
        except Exception as e:          # noqa: BLE001
            res['err'].append((name, e))

    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        ta = threading.Thread(target=worker, args=('a', model_a, bad_t, 6))
        tb = threading.Thread(target=worker, args=('b', model_b, mel_t, 24))
        ta.start(); tb.start(); ta.join(); tb.join()
    assert not res['err'], res['err']
    assert len(set(res['words'] + [main_words])) == 3                                    # three threads, three pairs of words
    diag = [(float((y - want_a).abs().max()), bool(torch.equal(y, res['a'][0]))) for y in res['a']]
    assert len(res['a']) == 6 and all(torch.equal(y, want_a) for y in res['a']), diag    # A: repaired in exact fp32, every time
    assert len(res['b']) == 24 and all(torch.equal(y, want_b) for y in res['b'])         # B: its own arithmetic, never rerun
    msgs = [str(w.message) for w in caught if str(w.message).startswith('pwv:')]
    assert len(msgs) == 6 and all('exact fp32' in m for m in msgs)                       # exactly A's six repairs, nothing for B
    assert not engine.range_flag_raised() and engine.persist_status() == 0               # (this thread's words: untouched)


@pytest.mark.gpu
def test_status_words_of_finished_threads_are_pooled(gpu):
    """every thread gets its own pinned pair of sticky words; a serving process that spawns short-lived workers must not
    leave one pinned allocation behind per worker.  A pair returns to the pool when its thread (and every graph that captured it) is gone,
    and comes out of it cleared."""
    import gc
    import threading
    from pwv_amd import engine
    addrs = []

    def worker():
        w = engine.current_words(gpu)
        w.range = 1                      # left raised: the next owner must not see it
        addrs.append(w.addr)

    for _ in range(4):
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        gc.collect()
    assert len(set(addrs)) < len(addrs), addrs          # later threads were handed pairs of earlier ones
    fresh = []
    t = threading.Thread(target=lambda: fresh.append((engine.current_words(gpu).addr, engine.current_words(gpu).range, engine.current_words(gpu).persist)))
    t.start()
    t.join()
    assert fresh[0][0] in addrs and fresh[0][1:] == (0, 0)


@pytest.mark.gpu
def test_default_model_stays_a_decade_inside_the_range_guard(gpu):
    """the margin between what the split-fp16 range guard allows and what the default model shows on the inputs every
    parity test and bench.py use (glorot weights, N(0, 0.1) biases, logistic noise, mel in [-1, 1]) -- limit / observed per operand class
    (engine.range_report) -- is at least 10x, i.e. the fast arithmetic is not one unlucky sample away from the exact-fp32 rerun.
    (tools/precision_report.py --range prints the table, also for a trained-like weight set: DESIGN.md section 2.)"""
    import torch
    from oracle import iaf_oracle as O
    from pwv_amd import engine
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    from tests.util import set_hparams
    cfg = O.ModelConfig()
    set_hparams(cfg)
    store = VariableStore(device=gpu)
    store.load_dict(O.init_weights(cfg, seed=2))
    n, length = 1, 4000
    mel_np, z_np = O.synthetic_inputs(n, length, cfg)
    mel, z = torch.from_numpy(mel_np).to(gpu), torch.from_numpy(z_np).to(gpu)
    model = IAFVocoder(batch_size=n, length=length, store=store, precision='f16x3')
    rep = engine.range_report(lambda: model(None, mel, is_training=False, z=z, verify=False))
    model.verify()
    assert set(rep['classes']) >= {'weights', 'residual', 'head_operand', 'flow_input', 'mel'}, rep
    assert rep['range_margin'] >= 10.0, rep
    assert rep['range_margin'] == min(c['margin'] for c in rep['classes'].values())
    # the report is diagnostics: with the log off nothing is read back
    assert engine.RANGE_LOG is None
