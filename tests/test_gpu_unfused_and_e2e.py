"""-m gpu: the un-fused GPU path (architectures outside the fused kernels' shape, normalisers: SURVEY.md 8 f-4)
against the oracle, and generate() end to end (hparams case -> checkpoint by TF name -> forward -> files)."""
import os

import numpy as np
import pytest

from oracle import iaf_oracle as O
from tests.util import TOL_F32, run_vocoder_hip, set_hparams, small_cfg

pytestmark = pytest.mark.gpu


def test_unfused_other_channel_counts(gpu):
    """R=32, D=48, S=64, W=3, C=40: composed from pwv_causal_conv_f32 + device elementwise ops."""
    import torch
    from pwv_amd.modules import WaveNet
    from pwv_amd.variables import VariableStore, variable_scope
    cfg = O.ModelConfig(dilations=[[1, 2, 5]], n_iaf=1, filter_width=3, residual_channels=32, dilation_channels=48,
                        skip_channels=64, condition_channels=40, use_skip_connection=True)
    w = O.init_weights(cfg, seed=8)
    rng = np.random.RandomState(0)
    x = rng.randn(2, 100, 1).astype(np.float32)
    cond = rng.randn(2, 100, 40).astype(np.float32)
    want = O.wavenet_forward(w, 'iaf_vocoder/iaf0/scalar', x, cond, dilations=[1, 2, 5], use_biases=True,
                             use_skip_connection=True)
    store = VariableStore(device=gpu)
    store.load_dict(w)
    with variable_scope('iaf_vocoder'), variable_scope('iaf0'):
        net = WaveNet(2, [1, 2, 5], 3, 32, 48, 64, quantization_channels=1, use_biases=True, condition_channels=40,
                      use_skip_connection=True, name='scalar', store=store)
    t = lambda a: torch.from_numpy(a).to(gpu)
    assert not net.fused_supported(t(cond))
    got = net(t(x), t(cond)).cpu().numpy()
    assert got.shape == want.shape and np.abs(got - want).max() <= TOL_F32


@pytest.mark.parametrize('method', ['in', 'bn'])
def test_normaliser_variants(gpu, method):
    """normalize_wavenet / normalize / normalize_cond = 'in' | 'bn' (modules.py:263-284): identity-initialised
    gamma/beta/moving stats, so the oracle needs no extra weights."""
    cfg = small_cfg(normalize_wavenet=method, normalize=method, normalize_cond=method)
    w = O.init_weights(cfg, seed=2)
    mel, z = O.synthetic_inputs(2, 320, cfg)
    want = O.iaf_vocoder_forward(w, mel, z, cfg)
    got = run_vocoder_hip(cfg, w, mel, z, gpu)
    assert np.isfinite(got).all() and np.abs(got - want).max() <= 5e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('method,skip', [('bn', False), ('bn', True), ('in', False)])
def test_normalisers_with_trained_statistics(gpu, method, skip):
    """SURVEY 8 f-4 with NON-trivial normaliser parameters (gamma / beta / moving statistics as a checkpoint would hold
    them).  'bn' at inference is folded into the packed weights (modules.WaveNet.folded_variables: every batch norm sits
    next to a convolution; the one on the residual output scales the identity branch and is carried as a diagonal
    re-scaling of the stream), so the net stays on the FUSED kernels; 'in' runs the un-fused path on HIP ops only
    (pwv_instance_norm_f32, pwv_gate_f32, ...).  Both against the fp64 oracle (modules.py:263-284)."""
    import torch
    from pwv_amd import engine
    from pwv_amd.models import IAFVocoder
    from pwv_amd.modules import WaveNet
    from pwv_amd.variables import VariableStore
    cfg = small_cfg(normalize_wavenet=method, normalize=method, normalize_cond=method, use_skip_connection=skip)
    set_hparams(cfg)
    store = VariableStore(device=gpu)
    store.load_dict(O.init_weights(cfg, seed=2))
    mel, z = O.synthetic_inputs(2, 320, cfg)
    mel_t, z_t = torch.from_numpy(mel).to(gpu), torch.from_numpy(z).to(gpu)
    model = IAFVocoder(batch_size=2, length=320, store=store)
    model(None, mel_t, is_training=False, z=z_t)            # creates the normaliser variables (identity-initialised)
    g = torch.Generator().manual_seed(11)
    n_norm = 0
    for name, v in store.vars.items():
        leaf = name.rsplit('/', 1)[1]
        if leaf in ('gamma', 'moving_variance'):
            v.copy_((torch.rand(v.shape, generator=g) + 0.5).to(gpu))
            n_norm += 1
        elif leaf in ('beta', 'moving_mean') and ('normalize' in name or 'batch_normalization' in name):
            v.copy_((torch.randn(v.shape, generator=g) * 0.2).to(gpu))
    assert n_norm >= 20
    store.version += 1
    w = store.numpy()
    want = O.iaf_vocoder_forward(w, mel, z, cfg)
    # the oracle really used the randomised parameters (guards against a naming mismatch silently meaning "identity")
    assert np.abs(want - O.iaf_vocoder_forward(O.init_weights(cfg, seed=2), mel, z, cfg)).max() > 1e-3
    log = []
    orig = engine._run_nets
    engine._run_nets = lambda *a, **k: (log.append(1), orig(*a, **k))[1]
    try:
        got = model(None, mel_t, is_training=False, z=z_t)
        model.verify()
    finally:
        engine._run_nets = orig
    assert (len(log) == cfg.n_iaf) == (method == 'bn')       # 'bn': one fused run_flow / run_nets per flow; 'in': the un-fused path
    err = np.abs(got.cpu().numpy() - want).max()
    assert err <= 5e-5 * max(1.0, np.abs(want).max()), err
    if method == 'bn':
        for prec in ('f32',):
            m32 = IAFVocoder(batch_size=2, length=320, store=store, precision=prec)
            e32 = np.abs(m32(None, mel_t, is_training=False, z=z_t).cpu().numpy() - want).max()
            assert e32 <= 5e-5 * max(1.0, np.abs(want).max()), e32


def test_instance_norm_op_matches_fp64_and_is_repeatable(gpu):
    """pwv_instance_norm_f32 alone: moments over time per (utterance, channel), eps 1e-8 (modules.py:274-284), odd sizes,
    a long time axis (many partial-sum chunks), large mean / small variance (fp64 accumulation), bitwise repeatable."""
    import torch
    from pwv_amd import engine
    rng = np.random.RandomState(0)
    for n, t, c in ((1, 7, 3), (3, 320, 64), (2, 5000, 80), (1, 300000, 64), (2, 1000, 130)):
        x = (rng.randn(n, t, c) * rng.uniform(0.01, 3.0, size=(1, 1, c)) + rng.uniform(-50, 50, size=(1, 1, c))).astype(np.float32)
        gamma, beta = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.randn(c).astype(np.float32)
        x64 = x.astype(np.float64)
        mean, var = x64.mean(axis=1, keepdims=True), x64.var(axis=1, keepdims=True)
        want = gamma * (x64 - mean) / np.sqrt(var + 1e-8) + beta
        xt = torch.from_numpy(x).to(gpu)
        y1 = engine.instance_norm_op(xt, torch.from_numpy(gamma).to(gpu), torch.from_numpy(beta).to(gpu))
        y2 = engine.instance_norm_op(xt, torch.from_numpy(gamma).to(gpu), torch.from_numpy(beta).to(gpu))
        assert torch.equal(y1, y2)
        # fp32 input quantisation of a value near 50 limits the result to ~1e-6 * 50 / sigma
        tol = 2e-5 * max(1.0, np.abs(want).max())
        assert np.abs(y1.cpu().numpy() - want).max() <= tol, (n, t, c)


def test_generate_end_to_end_with_tf_checkpoint(gpu, tmp_path, monkeypatch):
    """generate('bench/c1') restores a TensorFlow-format checkpoint by variable name (EMA shadows win), runs
    the HIP forward and writes wav / npy files; the result equals the oracle run on the EMA weights."""
    import torch
    from pwv_amd import tf_checkpoint as T
    from pwv_amd.generate import generate
    from pwv_amd.hparam import hparam as hp
    hp.set_hparam_yaml('bench/c1')
    cfg = O.ModelConfig.from_hparam(hp)
    ema = O.init_weights(cfg, seed=21)
    raw = O.init_weights(cfg, seed=22)
    ck = dict(raw)
    ck.update({k + '/ExponentialMovingAverage': v for k, v in ema.items()})
    logdir = tmp_path / 'logdir'
    logdir.mkdir()
    T.write_tf_checkpoint(str(logdir / 'model-100'), ck)
    (logdir / 'checkpoint').write_text('model_checkpoint_path: "model-100"\n')
    monkeypatch.setenv('PWV_LOGDIR', str(logdir))
    monkeypatch.setattr('pwv_amd.engine.logistic_noise_op', lambda shape, device, seed, offset=0, out=None: torch.zeros(shape, device=device))
    pred = generate('bench/c1')
    assert pred.shape == (1, 16000, 1)
    mel = (torch.rand((1, 201, 80), generator=torch.Generator().manual_seed(0)) * 2 - 1).numpy()
    want = O.iaf_vocoder_forward(ema, mel, np.zeros((1, 16000, 1), np.float32), cfg)
    assert np.abs(pred - want).max() <= TOL_F32
    assert os.path.exists(logdir / 'pred_0.wav') and os.path.exists(logdir / 'pred_wav.npy')
    # a checkpoint that lacks a model variable must fail like tf.train.Saver.restore (generate.py:59-63), not
    # silently run on random weights
    short = {k: v for k, v in ck.items() if 'layer3/dense' not in k}
    T.write_tf_checkpoint(str(logdir / 'model-200'), short)
    (logdir / 'checkpoint').write_text('model_checkpoint_path: "model-200"\n')
    with pytest.raises(KeyError, match='layer3/dense'):
        generate('bench/c1')
    # ... and with train.use_ema (default.yaml:47) a variable whose SHADOW is missing fails too: the reference's Saver is
    # keyed by the shadow names (generate.py:59-63), it never falls back to the raw variable
    no_shadow = {k: v for k, v in ck.items() if k != 'iaf_vocoder/iaf0/shifter/dilated_stack/layer2/gate/ExponentialMovingAverage'}
    T.write_tf_checkpoint(str(logdir / 'model-300'), no_shadow)
    (logdir / 'checkpoint').write_text('model_checkpoint_path: "model-300"\n')
    with pytest.raises(KeyError, match='no ExponentialMovingAverage shadow.*layer2/gate'):
        generate('bench/c1')


def test_sampled_noise_differs_between_calls_and_models(gpu):
    """models.py:32-33 draws fresh Logistic noise per sess.run: consecutive forwards of one model, an eager and a
    graph-replayed forward, and two model objects must not repeat the same noise."""
    import torch
    from pwv_amd.graph import GraphedVocoder
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    cfg = small_cfg()
    set_hparams(cfg)
    store = VariableStore(device=gpu)
    store.load_dict(O.init_weights(cfg, seed=2))
    mel = torch.from_numpy(O.synthetic_inputs(1, 800, cfg)[0]).to(gpu)
    m = IAFVocoder(batch_size=1, length=800, store=store)
    a, b = m(None, mel), m(None, mel)
    assert not torch.equal(a, b)
    g = GraphedVocoder(m)
    c = g(mel).clone()
    d = g(mel).clone()
    assert not torch.equal(c, d) and not torch.equal(c, a) and not torch.equal(c, b)
    m2 = IAFVocoder(batch_size=1, length=800, store=store)
    assert not torch.equal(m2(None, mel), a)
    # a pinned seed reproduces the stream
    m3, m4 = IAFVocoder(1, 800, store=store), IAFVocoder(1, 800, store=store)
    m3.noise_seed = m4.noise_seed = 77
    assert torch.equal(m3(None, mel), m4(None, mel))


def test_plan_cache_distinguishes_architectures_in_one_scope(gpu):
    """Two WaveNets built in the SAME store + scope with different structure (skip accumulation toggled, fewer
    layers): no variable is created by the second one, so the packed-plan cache must key on the architecture."""
    import torch
    from pwv_amd.modules import WaveNet
    from pwv_amd.variables import VariableStore
    store = VariableStore(device=gpu, seed=3)
    kw = dict(batch_size=1, filter_width=2, residual_channels=64, dilation_channels=64, skip_channels=128,
              quantization_channels=1, use_biases=True, condition_channels=None, is_training=False, name='net', store=store)
    x = torch.randn(1, 300, 1, device=gpu)
    full = WaveNet(dilations=[1, 2, 4, 8], use_skip_connection=True, **kw)
    y_skip = full(x)
    w = {k: v.cpu().numpy() for k, v in store.vars.items()}
    plain = WaveNet(dilations=[1, 2, 4, 8], use_skip_connection=False, **kw)
    short = WaveNet(dilations=[1, 2], use_skip_connection=False, **kw)
    y_plain, y_short = plain(x), short(x)
    xn = x.cpu().numpy().astype(np.float64)
    for y, dil, skip in ((y_skip, [1, 2, 4, 8], True), (y_plain, [1, 2, 4, 8], False), (y_short, [1, 2], False)):
        want = O.wavenet_forward(w, 'net', xn, None, dil, True, skip)
        assert np.abs(y.cpu().numpy() - want).max() <= TOL_F32


def test_generate_sharded_over_rccl_world1(gpu):
    """The utterance scatter / gather collectives on DEVICE tensors over backend 'nccl' (= RCCL), world size 1
    (all this box has), with the real HIP forward: result equals the direct batched forward bit for bit."""
    import socket
    import torch
    import torch.distributed as dist
    from pwv_amd.distributed import generate_sharded
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    cfg = small_cfg()
    set_hparams(cfg)
    store = VariableStore(device=gpu)
    store.load_dict(O.init_weights(cfg, seed=2))
    mel, z = O.synthetic_inputs(3, 240, cfg)
    mel_d, z_d = torch.from_numpy(mel).to(gpu), torch.from_numpy(z).to(gpu)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=gpu)
    try:
        def forward(mel_local, z_local):
            m = IAFVocoder(batch_size=mel_local.shape[0], length=240, store=store)
            return m(None, mel_local, is_training=False, z=z_local)
        got = generate_sharded(forward, mel_d, (mel.shape[1], mel.shape[2]), 240, gpu, z=z_d)
        want = forward(mel_d, z_d)
        assert got.is_cuda and torch.equal(got, want)
    finally:
        dist.destroy_process_group()


def test_generate_from_wav_files(gpu, tmp_path, monkeypatch):
    """generate() on a directory of .wav files: data_load.py-style split (last 10 % of the files), wav -> mel
    front-end, HIP forward, one waveform per file written next to the checkpoint directory."""
    from scipy.io import wavfile
    from pwv_amd.generate import generate
    from pwv_amd.hparam import hparam as hp
    sr = 16000
    for i in range(10):
        t = np.arange(sr + 4000) / sr
        wav = 0.3 * np.sin(2 * np.pi * (200 + 40 * i) * t) * (t > 0.1)
        wavfile.write(str(tmp_path / ('a%02d.wav' % i)), sr, (wav * 32767).astype(np.int16))
    logdir = tmp_path / 'out'
    monkeypatch.setenv('PWV_LOGDIR', str(logdir))
    orig = type(hp).set_hparam_yaml

    def patched(self, case, *a, **k):          # what a user's hparams.yaml case would override
        r = orig(self, case, *a, **k)
        self.data_path = str(tmp_path / '*.wav')
        self.generate.length, self.generate.batch_size = 8000, 2
        self.model.n_iaf, self.model.dilations = 1, [[1, 2, 4, 8]]
        return r

    monkeypatch.setattr(type(hp), 'set_hparam_yaml', patched)
    pred = generate('default')
    assert pred.shape == (2, 8000, 1) and np.isfinite(pred).all()
    assert (logdir / 'pred_0.wav').exists() and (logdir / 'pred_1.wav').exists()
    rate, data = wavfile.read(str(logdir / 'pred_0.wav'))
    assert rate == sr and data.shape == (8000,)


def test_graphed_vocoder_matches_eager_bitwise(gpu):
    """pwv_amd/graph.py: the captured forward is the same launches -> same bits; new inputs go through the static
    buffers; changing a weight re-captures (the graph would otherwise replay stale packed weights)."""
    import torch
    from oracle import iaf_oracle as O
    from pwv_amd.graph import GraphedVocoder
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    from tests.util import set_hparams, small_cfg
    cfg = small_cfg()
    set_hparams(cfg)
    store = VariableStore(device=gpu)
    store.load_dict(O.init_weights(cfg, seed=4))
    n, length = 2, 480
    model = IAFVocoder(batch_size=n, length=length, store=store)
    mel_np, z_np = O.synthetic_inputs(n, length, cfg)
    mel, z = torch.from_numpy(mel_np).to(gpu), torch.from_numpy(z_np).to(gpu)
    want = model(None, mel, is_training=False, z=z).clone()
    graphed = GraphedVocoder(model)
    assert torch.equal(graphed(mel, z=z), want)
    mel2, z2 = mel * 0.5, z.flip(1).contiguous()
    assert torch.equal(graphed(mel2, z=z2).clone(), model(None, mel2, is_training=False, z=z2))
    # sampled noise: the sampler is a node of the graph (pwv_logistic_noise_stream_f32: its counter range lives in device memory and
    # the captured kernel moves it on), so a forward is ONE graph launch -- and every replay draws the model's next counter range,
    # exactly what IAFVocoder.sample_noise draws eagerly
    from pwv_amd import engine
    model.noise_seed, off, numel = 1234, model.noise_offset, n * length
    a = graphed(mel).clone()
    b = graphed(mel).clone()
    assert torch.isfinite(a).all() and not torch.equal(a, b) and model.noise_offset == off + 2 * numel
    za = engine.logistic_noise_op((n, length, 1), gpu, seed=1234, offset=off)
    zb = engine.logistic_noise_op((n, length, 1), gpu, seed=1234, offset=off + numel)
    assert torch.equal(a, model(None, mel, is_training=False, z=za)) and torch.equal(b, model(None, mel, is_training=False, z=zb))
    eager = model(None, mel, is_training=False).clone()                      # an eager draw in between takes the next range ...
    zc = engine.logistic_noise_op((n, length, 1), gpu, seed=1234, offset=off + 2 * numel)
    assert torch.equal(eager, model(None, mel, is_training=False, z=zc))
    c = graphed(mel, seed=77).clone()                                        # ... and the graph goes on behind it, here with another seed
    zd = engine.logistic_noise_op((n, length, 1), gpu, seed=77, offset=off + 3 * numel)
    assert torch.equal(c, model(None, mel, is_training=False, z=zd))
    assert torch.equal(graphed(mel, z=z), want)                              # explicit z again: the sampler node leaves it alone
    name = 'iaf_vocoder/iaf0/scalar/postprocessing/postprocess2_bias'
    store.assign(name, (store.vars[name] + 0.25).cpu().numpy())
    got = graphed(mel, z=z).clone()
    assert torch.equal(got, model(None, mel, is_training=False, z=z)) and not torch.equal(got, want)
    with pytest.raises(ValueError):
        graphed(mel[:, :-1])


def test_graphed_vocoder_recaptures_when_the_engine_switches_launch_paths(gpu):
    """A captured graph holds the launches of the path the engine was on.  After a persistent give-up the engine is on per-layer
    launches: GraphedVocoder.verify() raises like IAFVocoder.verify() AND re-captures, so the caller's rerun replays launches that
    can complete; a switch made behind its back (engine.PERSIST set by another model's verified call) is noticed at the next
    replay.  (The give-up word is poked from the host: a real one needs a second process on the GPU.)"""
    import ctypes
    import torch
    from pwv_amd import engine
    from pwv_amd._lib import PwvPersistError
    from pwv_amd.graph import GraphedVocoder
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    cfg = O.ModelConfig(dilations=[[1, 2, 4, 8, 16, 32], [1, 2, 4, 8, 16, 32]], n_iaf=2)
    set_hparams(cfg)
    store = VariableStore(device=gpu)
    store.load_dict(O.init_weights(cfg, seed=2))
    n, length = 1, 16000
    mel_np, z_np = O.synthetic_inputs(n, length, cfg)
    mel, z = torch.from_numpy(mel_np).to(gpu), torch.from_numpy(z_np).to(gpu)
    model = IAFVocoder(batch_size=n, length=length, store=store)
    saved = engine.PERSIST
    try:
        engine.PERSIST = False
        want = model(None, mel, is_training=False, z=z).clone()
        engine.PERSIST = True
        graphed = GraphedVocoder(model)
        captured = graphed.graph
        assert torch.equal(graphed(mel, z=z), want)
        graphed.verify()
        # BOTH words raised, as after a real give-up whose garbage tripped the range guard downstream: verify() raises the
        # persist error, clears the range word with it, re-captures -- and the prescribed rerun goes through
        engine.poke_persist_status(4)
        engine.current_words().range = 1
        with pytest.raises(PwvPersistError):
            graphed.verify()
        assert engine.persist_suspended() and graphed.graph is not captured      # re-captured on the per-layer path
        assert not engine.range_flag_raised()
        assert torch.equal(graphed(mel, z=z), want)
        graphed.verify()
        # ... the suspension counts down with the replays; when it is over the next call re-captures on the persistent path
        recaptured = graphed.graph
        for _ in range(engine.PERSIST_RETRY_AFTER):
            assert torch.equal(graphed(mel, z=z), want)
        graphed.verify()
        assert not engine.persist_suspended() and graphed.graph is not recaptured
        # ... and a switch it was not told about: noticed at the next replay
        recaptured = graphed.graph
        engine.PERSIST = False
        assert torch.equal(graphed(mel, z=z), want) and graphed.graph is not recaptured
        graphed.verify()
    finally:
        engine.PERSIST = saved
        engine.resume_persist()


def test_graphed_rerun_after_a_failed_verify_draws_the_same_noise_and_large_seeds_work(gpu):
    """the sampler is a node of the graph and the model's noise offset moves on when a replay is ENQUEUED; a verify() that
    fails must hand the range back, so that the prescribed rerun (z = None again) is a rerun on the same noise, as on the eager path.  And a
    seed >= 2**63 -- accepted by the eager sampler (uint64) -- must reach the device state with the same bits."""
    import torch
    from pwv_amd import engine
    from pwv_amd._lib import PwvPersistError, PwvRangeError
    from pwv_amd.graph import GraphedVocoder
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    from tests.util import small_cfg
    cfg = small_cfg()
    set_hparams(cfg)
    store = VariableStore(device=gpu)
    store.load_dict(O.init_weights(cfg, seed=4))
    n, length = 2, 480
    model = IAFVocoder(batch_size=n, length=length, store=store)
    mel = torch.from_numpy(O.synthetic_inputs(n, length, cfg)[0]).to(gpu)
    graphed = GraphedVocoder(model)
    try:
        big = (1 << 63) + 12345
        model.noise_seed, off = big, model.noise_offset
        first = graphed(mel).clone()
        graphed.verify()
        z0 = engine.logistic_noise_op((n, length, 1), gpu, seed=big, offset=off)
        assert torch.equal(first, model(None, mel, is_training=False, z=z0))
        # a replay whose verification fails (the range word poked from the host; a real one needs an out-of-range mel) ...
        off = model.noise_offset
        failed = graphed(mel).clone()
        engine.current_words().range = 1
        with pytest.raises(PwvRangeError):
            graphed.verify()
        assert model.noise_offset == off                                      # ... has handed its noise range back
        again = graphed(mel).clone()
        graphed.verify()
        assert torch.equal(again, failed) and model.noise_offset == off + n * length
        # ... the same after a give-up (the graph is re-captured on the per-layer path: same noise, same bits by the cross-path identity)
        off = model.noise_offset
        failed = graphed(mel).clone()
        engine.poke_persist_status(4)
        with pytest.raises(PwvPersistError):
            graphed.verify()
        assert model.noise_offset == off
        again = graphed(mel).clone()
        graphed.verify()
        assert torch.equal(again, failed)
    finally:
        engine.resume_persist()


def test_bench_multi_rank_control_flow_on_one_gpu():
    """bench.py for N > 1, both ways it can be started -- under torch.distributed.run (one process per rank) and plainly
    (`python bench.py --gpus 2`: it then spawns its ranks itself) -- with the test hook PWV_BENCH_DRYRUN_ONE_GPU=1: both
    ranks share GPU 0 and rendezvous over gloo.  Checks what does not need N GPUs: every
    rank builds / waits, the timed region is bracketed by barriers, rank 0 alone prints ONE JSON line with whole-job
    throughput for n_gpus = 2 and weak scaling."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PWV_BENCH_DRYRUN_ONE_GPU='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    tail = [os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--case', 'bench/c1', '--no-cpu-baseline']
    launcher = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                '--master-port', str(port)]
    for cmd in (launcher + tail, [sys.executable] + tail):
        res = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1, res.stdout[-2000:]
        j = json.loads(lines[0])
        assert j['n_gpus'] == 2 and j['scaling'] == 'weak' and j['steps'] == 3 and j['value'] > 0
        assert 'utterance-sharded x2' in j['config']['parallelism'] and 'cpu_baseline' not in j
        assert 'scattered over 2 ranks' in j['sharded_generate'] and j['f32_exact']['value'] > 0


@pytest.mark.parametrize('extra,want', [(['--case', 'bench/c1'], 'scattered over 1 ranks'),
                                        (['--case', 'bench/c1', '--length', '32000', '--shard', 'time'], 'cut into 1 time shards')])
def test_bench_takes_every_rccl_branch_at_world_size_one(extra, want):
    """PWV_BENCH_FORCE_DIST=1: `python bench.py --gpus 1` initialises the `nccl` backend (RCCL) at world size 1 and takes every
    distributed branch the driver's 8-GPU run takes -- init_process_group('nccl', device_id=...), barriers, the MAX all-reduce
    of the elapsed time and the give-up vote on DEVICE tensors, generate_sharded / generate_time_sharded_ranks over nccl
    broadcast / scatter / gather with device chunks.  What is left for N > 1 is the fabric, not the code path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PWV_BENCH_FORCE_DIST='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'PWV_BENCH_DRYRUN', 'PWV_BENCH_DRYRUN_ONE_GPU'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-f32-exact'] + extra
    res = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 1 and j['rccl_ranks'] == 1 and j['backend'] == 'nccl' and j['value'] > 0
    assert want in j['sharded_generate'] and j['sharded_generate'].endswith('ok')
    # ... and the job-level number next to the forward-only `value`: rank 0's inputs -> RCCL scatter -> verified forward -> gather
    job = j['job']
    assert job['calls'] == 5 and job['job_samples_per_s'] > 0 and job['job_samples_per_s'] < 1.05 * j['value']


def test_device_mel_frontend_matches_numpy_restatement(gpu):
    """pwv_wav_to_mel_db_f32 (STFT -> Slaney mel -> dB -> [-1, 1]) against the numpy restatement of
    data_load.py:51-54 / audio.py:102-141,232-243,254-286,341-356 on speech-like and edge-case signals; <= 1e-5."""
    import torch
    from pwv_amd import audio_frontend as A
    from pwv_amd.hparam import hparam as hp
    hp.set_hparam_yaml('default')
    s = hp.signal
    rng = np.random.RandomState(5)
    L = 16000
    t = np.arange(L) / s.sr
    wavs = np.stack([
        (0.3 * np.sin(2 * np.pi * 220 * t) * np.exp(-3 * t) + 0.05 * rng.randn(L)),         # decaying tone + noise
        0.8 * rng.randn(L) * (t > 0.3),                                                    # silence then loud noise (top_db clip)
        np.concatenate([np.zeros(L // 2), 1e-4 * rng.randn(L // 2)]),                       # near the amin floor
        np.sign(np.sin(2 * np.pi * 50 * t)) * 0.5,                                         # square wave: many harmonics
    ]).astype(np.float32)
    want = np.stack([A.wav2melspec_db(w, s.sr, s.n_fft, s.win_length, s.hop_length, s.n_mels, max_db=s.max_db, min_db=s.min_db) for w in wavs])
    got = A.wav_to_mel_device(torch.from_numpy(wavs).to(gpu)).cpu().numpy()
    assert got.shape == want.shape == (4, 201, 80)
    assert np.abs(got - want).max() <= 1e-5
    raw_want = np.stack([A.wav2melspec_db(w, s.sr, s.n_fft, s.win_length, s.hop_length, s.n_mels) for w in wavs[:1]])
    raw_got = A.wav_to_mel_device(torch.from_numpy(wavs[:1]).to(gpu), normalise=False).cpu().numpy()
    assert np.abs(raw_got - raw_want).max() <= 5e-4                                       # dB units, before normalisation
