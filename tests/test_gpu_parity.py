"""-m gpu: parity of the HIP path (through the C ABI, via the reference-shaped host API) against
the CPU oracle on the same seeded inputs.  fp32 tolerance: max|y - y_fp64| <= 2e-5 (util.TOL_F32)."""
import numpy as np
import pytest

from oracle import iaf_oracle as O
from pwv_amd._lib import PwvRangeError
from tests.util import TOL_F32, run_vocoder_hip, set_hparams, small_cfg

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


@pytest.mark.parametrize('N,T,Cin,Cout,W,d', [
    (1, 10, 1, 64, 2, 1), (3, 33, 5, 7, 2, 3), (3, 37, 5, 7, 3, 4), (1, 64, 64, 64, 2, 8),
    (2, 10, 4, 4, 2, 512), (3, 37, 8, 12, 3, 16), (2, 100, 80, 64, 1, 1), (1, 1, 3, 3, 2, 1),
])
def test_causal_conv(gpu, N, T, Cin, Cout, W, d):
    from pwv_amd.modules import causal_conv
    rng = np.random.RandomState(0)
    x = rng.randn(N, T, Cin).astype(np.float32)
    f = rng.randn(W, Cin, Cout).astype(np.float32)
    want = O.causal_conv_literal(x.astype(np.float64), f.astype(np.float64), d)
    got = causal_conv(_t(x, gpu), _t(f, gpu), d).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_causal_conv_empty(gpu):
    from pwv_amd.modules import causal_conv
    import torch
    y = causal_conv(torch.zeros((2, 0, 4), device=gpu), torch.zeros((2, 4, 6), device=gpu), 2)
    assert tuple(y.shape) == (2, 0, 6)


@pytest.mark.parametrize('M,K,Nout,relu,bias', [(201, 80, 80, True, False), (50, 80, 400, True, False),
                                                (333, 80, 1280, False, True), (7, 64, 36, False, True),
                                                (129, 128, 132, True, True), (65, 24, 260, False, True),
                                                (2001, 80, 3840, False, True)])
@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_linear(gpu, M, K, Nout, relu, bias, precision):
    from pwv_amd import engine
    rng = np.random.RandomState(1)
    x = rng.randn(M, K).astype(np.float32)
    w = (rng.randn(K, Nout) / np.sqrt(K)).astype(np.float32)
    b = rng.randn(Nout).astype(np.float32) if bias else None
    want = x.astype(np.float64) @ w.astype(np.float64) + (b.astype(np.float64) if bias else 0)
    if relu:
        want = np.maximum(want, 0)
    got = engine.linear_op(_t(x, gpu), _t(w, gpu), _t(b, gpu) if bias else None, relu, precision=precision).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-5


PRECS = ['f32', 'f16x3']


def _wavenet_case(gpu, cond_mode, use_skip, use_biases, q_out, T=300, N=2, dil=(1, 2, 4, 8, 16), precision='f32'):
    import torch
    from pwv_amd.engine import RepeatedCondition
    from pwv_amd.modules import WaveNet
    from pwv_amd.variables import VariableStore, variable_scope
    rng = np.random.RandomState(3)
    cfg = O.ModelConfig(dilations=[list(dil)], n_iaf=1, use_skip_connection=use_skip, use_biases=use_biases,
                        cond_upsample_method='repeat' if cond_mode != 'none' else 'none',
                        shared_nets=(q_out == 2))
    weights = O.init_weights(cfg, seed=5)
    net_name = 'shared' if q_out == 2 else 'scalar'
    scope = 'iaf_vocoder/iaf0/' + net_name
    x = rng.randn(N, T, 1).astype(np.float32)
    hop = 80
    cond_np = None
    cond_dev = None
    if cond_mode != 'none':
        assert T % hop == 0
        frames = np.maximum(rng.randn(N, T // hop + 1, 80), 0).astype(np.float32)
        cond_np = np.repeat(frames, hop, axis=1)[:, hop // 2: -(hop // 2), :]
        if cond_mode == 'frames':
            cond_dev = RepeatedCondition(_t(frames, gpu), hop, hop // 2, T)
        else:
            cond_dev = _t(cond_np, gpu)
    want = O.wavenet_forward(weights, scope, x, cond_np, dilations=list(dil), use_biases=use_biases,
                             use_skip_connection=use_skip)
    store = VariableStore(device=gpu)
    store.load_dict(weights)
    with variable_scope('iaf_vocoder'), variable_scope('iaf0'):
        net = WaveNet(batch_size=N, dilations=list(dil), filter_width=2, residual_channels=64, dilation_channels=64,
                      skip_channels=128, quantization_channels=q_out, input_channels=1, use_biases=use_biases,
                      condition_channels=80 if cond_mode != 'none' else None, use_skip_connection=use_skip,
                      name=net_name, store=store, precision=precision)
    assert net.fused_supported(cond_dev)
    got = net(_t(x, gpu), cond_dev)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert got.shape == want.shape
    err = np.abs(got - want).max()
    assert err <= TOL_F32, err


@pytest.mark.parametrize('precision', PRECS)
@pytest.mark.parametrize('cond_mode', ['frames', 'none', 'samples'])
@pytest.mark.parametrize('use_skip', [False, True])
def test_wavenet_fused(gpu, cond_mode, use_skip, precision):
    _wavenet_case(gpu, cond_mode, use_skip, use_biases=True, q_out=1, T=320, precision=precision)


@pytest.mark.parametrize('precision', PRECS)
def test_wavenet_no_biases_q2(gpu, precision):
    _wavenet_case(gpu, 'frames', False, use_biases=False, q_out=2, T=240, precision=precision)


@pytest.mark.parametrize('precision', PRECS)
def test_wavenet_ragged_tail_and_big_dilation(gpu, precision):
    # T not a multiple of the 32-row unit, dilation larger than a tile and larger than T/2
    _wavenet_case(gpu, 'none', False, True, 1, T=333, N=3, dil=(1, 512, 2, 256, 128), precision=precision)


@pytest.mark.parametrize('precision', PRECS)
@pytest.mark.parametrize('method', ['repeat', 'transposed_conv'])
def test_vocoder_small(gpu, method, precision):
    cfg = small_cfg(cond_upsample_method=method)
    weights = O.init_weights(cfg, seed=2)
    mel, z = O.synthetic_inputs(2, 480, cfg)
    want = O.iaf_vocoder_forward(weights, mel, z, cfg)
    got = run_vocoder_hip(cfg, weights, mel, z, gpu, precision=precision)
    err = np.abs(got - want).max()
    assert got.shape == want.shape and err <= TOL_F32, err


@pytest.mark.parametrize('precision', PRECS)
def test_vocoder_default_model_1s(gpu, precision):
    """The reference-default model (4 flows, 8 nets, 120 layers) on 0.5 s of synthetic mel."""
    cfg = O.ModelConfig()
    weights = O.init_weights(cfg, seed=2)
    mel, z = O.synthetic_inputs(1, 8000, cfg)
    want = O.iaf_vocoder_forward(weights, mel, z, cfg)
    got = run_vocoder_hip(cfg, weights, mel, z, gpu, precision=precision)
    err = np.abs(got - want).max()
    assert err <= TOL_F32, err


@pytest.mark.parametrize('precision', PRECS)
def test_vocoder_shared_nets(gpu, precision):
    cfg = small_cfg(shared_nets=True)
    weights = O.init_weights(cfg, seed=2)
    mel, z = O.synthetic_inputs(2, 400, cfg)
    want = O.iaf_vocoder_forward(weights, mel, z, cfg)
    got = run_vocoder_hip(cfg, weights, mel, z, gpu, precision=precision)
    assert np.abs(got - want).max() <= TOL_F32


@pytest.mark.parametrize('zscale,wscale', [(100, 1), (20, 2), (100, 4)])
def test_large_magnitude_activations_f16x3(gpu, zscale, wscale):
    """The split keeps ~22 mantissa bits at any scale inside fp16's range: with inputs 20-100x larger and
    weights up to 4x larger (outputs up to ~4e5, heavy cancellation) the split-fp16 path must stay as close
    to the fp64 oracle as the exact-fp32 path does (both lose the same digits to conditioning)."""
    cfg = small_cfg()
    weights = {k: (v * wscale if v.ndim > 1 else v) for k, v in O.init_weights(cfg, seed=6).items()}
    mel, z = O.synthetic_inputs(1, 480, cfg)
    z = z * zscale
    want = O.iaf_vocoder_forward(weights, mel, z, cfg)
    e32 = np.abs(run_vocoder_hip(cfg, weights, mel, z, gpu, precision='f32') - want).max()
    scale = max(1.0, np.abs(want).max())
    try:
        e16 = np.abs(run_vocoder_hip(cfg, weights, mel, z, gpu, precision='f16x3') - want).max()
    except PwvRangeError:
        return      # the (conservative) range guard refused the input: loud, and the f32 path above is the answer
    assert np.isfinite(e16) and e16 <= max(3 * e32, 2e-5 * scale) and e16 <= 2e-4 * scale, (e16, e32, scale)


@pytest.mark.parametrize('zscale,wscale,melscale', [(1e4, 1, 1), (1, 64, 1), (1e4, 64, 1), (1, 1, 1e5), (float('nan'), 1, 1)])
def test_f16x3_range_guard_never_returns_garbage(gpu, zscale, wscale, melscale):
    """fp16 has 5 exponent bits: a GEMM operand beyond 65504 would turn into inf in the split-fp16 kernels where the
    reference's fp32 (models.py:81-82) is fine.  Outside the range the path must EITHER still match the fp64 oracle as
    well as the exact-fp32 path does (weights alone out of range: the plan falls back to 'f32') OR raise PwvRangeError
    -- never hand back inf / garbage silently.  The f32 rerun is then the answer."""
    import warnings
    cfg = small_cfg()
    weights = {k: (v * wscale if v.ndim > 1 else v) for k, v in O.init_weights(cfg, seed=6).items()}
    mel, z = O.synthetic_inputs(1, 480, cfg)
    z = (z * zscale).astype(np.float32)
    mel = (mel * melscale).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if not np.isfinite(zscale):
            with pytest.raises(PwvRangeError):
                run_vocoder_hip(cfg, weights, mel, z, gpu, precision='f16x3')
            return
        with np.errstate(all='ignore'):
            want = O.iaf_vocoder_forward(weights, mel, z, cfg)
        scale = max(1.0, np.abs(want).max())
        y32 = run_vocoder_hip(cfg, weights, mel, z, gpu, precision='f32')
        e32 = np.abs(y32 - want).max()
        assert np.isfinite(y32).all()       # (with weights x 64 the model is ill-conditioned: fp32 itself drifts from fp64)
        try:
            y16 = run_vocoder_hip(cfg, weights, mel, z, gpu, precision='f16x3')
        except PwvRangeError:
            raised = True
        else:
            raised = False
            e16 = np.abs(y16 - want).max()
            assert np.isfinite(y16).all() and e16 <= max(3 * e32, 2e-5 * scale), (e16, e32, scale)
        if melscale > 1:
            assert raised            # |mel| * ||dense||_1 is beyond fp16 for sure
        # the flag is sticky only until it has been reported: the next in-range forward is clean
        mel1, z1 = O.synthetic_inputs(1, 480, cfg)
        w1 = O.init_weights(cfg, seed=6)
        got = run_vocoder_hip(cfg, w1, mel1, z1, gpu, precision='f16x3')
        assert np.abs(got - O.iaf_vocoder_forward(w1, mel1, z1, cfg)).max() <= TOL_F32


def test_plan_cache_is_per_store(gpu):
    """Packed-weight plans are keyed by a never-reused store uid: a new store with different weights under
    the same scope names must not see the previous store's packed weights."""
    cfg = small_cfg()
    mel, z = O.synthetic_inputs(1, 240, cfg)
    for seed in (11, 12, 13):
        weights = O.init_weights(cfg, seed=seed)
        want = O.iaf_vocoder_forward(weights, mel, z, cfg)
        got = run_vocoder_hip(cfg, weights, mel, z, gpu)
        assert np.abs(got - want).max() <= TOL_F32


def test_upsample_cond_api(gpu):
    """IAFVocoder._upsample_cond returns the materialised [N, T, C] condition (models.py:105-136)."""
    import torch
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    for method in ('repeat', 'transposed_conv'):
        cfg = small_cfg(cond_upsample_method=method)
        weights = O.init_weights(cfg, seed=4)
        mel, _ = O.synthetic_inputs(2, 320, cfg)
        set_hparams(cfg)
        store = VariableStore(device=gpu)
        store.load_dict(weights)
        model = IAFVocoder(batch_size=2, length=320, store=store)
        got = model._upsample_cond(torch.from_numpy(mel).to(gpu), is_training=False, strides=[4, 4, 5]).cpu().numpy()
        if method == 'repeat':
            want = O.upsample_cond_repeat(weights, mel, 80)
        else:
            want = O.upsample_cond_transposed(weights, mel, 80, (4, 4, 5))
        assert got.shape == want.shape == (2, 320, 80)
        assert np.abs(got - want).max() <= 1e-5


def test_logistic_noise(gpu):
    from pwv_amd import engine
    z = engine.logistic_noise_op((4, 50000, 1), gpu, seed=7).cpu().numpy().ravel()
    z2 = engine.logistic_noise_op((4, 50000, 1), gpu, seed=7).cpu().numpy().ravel()
    assert np.array_equal(z, z2)                       # counter-based: reproducible
    assert np.isfinite(z).all()
    assert abs(z.mean()) < 0.02 and abs(z.var() - np.pi ** 2 / 3) < 0.1   # Logistic(0,1)
    # never +-inf: 2^25 consecutive counters hit every 23-bit pattern of u several times, including the extremes
    # u = 2^-24 and u = 1 - 2^-24 (a 24-bit u rounded to exactly 1.0 once in 2^24 samples: z = +inf, round 2)
    big = engine.logistic_noise_op((1, 1 << 25, 1), gpu, seed=3)
    import torch
    assert bool(torch.isfinite(big).all()) and float(big.abs().max()) <= 16.7


@pytest.mark.parametrize('precision', PRECS)
@pytest.mark.parametrize('N,T,dil', [(1, 1, (1, 2)), (1, 7, (1, 2, 4)), (5, 33, (1, 16, 64)), (7, 10, (1, 2, 512)),
                                     (2, 129, (128, 1))])
def test_wavenet_tiny_and_ragged(gpu, N, T, dil, precision):
    """Edge cases: a single sample, T shorter than one 32-row unit, several utterances inside one unit (x[t-d]
    must not leak across utterance boundaries), dilation > T, T one past a 128-row boundary."""
    _wavenet_case(gpu, 'none', False, True, 1, T=T, N=N, dil=dil, precision=precision)


def test_vocoder_one_frame(gpu):
    """Shortest legal utterance: one hop (t_mel = 2)."""
    cfg = small_cfg()
    weights = O.init_weights(cfg, seed=2)
    mel, z = O.synthetic_inputs(3, 80, cfg)
    want = O.iaf_vocoder_forward(weights, mel, z, cfg)
    got = run_vocoder_hip(cfg, weights, mel, z, gpu)
    assert got.shape == (3, 80, 1) and np.abs(got - want).max() <= TOL_F32


def test_misuse_raises(gpu):
    """Shape errors surface as exceptions (the reference asserts / TF shape errors), never as silent garbage."""
    import torch
    from pwv_amd.models import IAFVocoder
    from pwv_amd.modules import causal_conv
    from pwv_amd.variables import VariableStore
    cfg = small_cfg()
    set_hparams(cfg)
    store = VariableStore(device=gpu)
    with pytest.raises(ValueError):                       # length must be a multiple of hop (crop at models.py:133)
        IAFVocoder(2, 100, store=store)(None, torch.zeros(2, 2, 80, device=gpu), False)
    with pytest.raises(ValueError):                       # wrong t_mel
        IAFVocoder(2, 160, store=store)(None, torch.zeros(2, 5, 80, device=gpu), False)
    with pytest.raises(ValueError):                       # z shape
        IAFVocoder(2, 160, store=store)(None, torch.zeros(2, 3, 80, device=gpu), False, z=torch.zeros(2, 80, 1, device=gpu))
    with pytest.raises(ValueError):                       # filter Cin mismatch
        causal_conv(torch.zeros(1, 8, 4, device=gpu), torch.zeros(2, 5, 4, device=gpu), 1)
    hp = set_hparams(cfg)
    hp.signal.hop_length = 40                             # prod(strides) != hop  (assert at models.py:106)
    with pytest.raises(AssertionError):
        IAFVocoder(1, 80, store=store)(None, torch.zeros(1, 3, 80, device=gpu), False)
    set_hparams(cfg)


@pytest.mark.parametrize('rows,C', [(1, 4), (31, 64), (32, 64), (33, 80), (1000, 128)])
def test_tile32_round_trip(gpu, rows, C):
    """include/pwv_hip.h: float index of (row, c) = (row/32)*32*C + (c/4)*128 + (row%32)*4 + c%4."""
    import ctypes
    import torch
    from pwv_amd import _lib
    lib = _lib.lib()
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.randn(rows, C, device=gpu)
    nf = lib.pwv_tile32_floats(rows, C)
    blocks = (rows + 31) // 32
    assert nf == blocks * 32 * C
    tiled = torch.full((nf,), -7.0, device=gpu)
    _lib.check(lib.pwv_rows_to_tile32_f32(x.data_ptr(), tiled.data_ptr(), rows, C, s))
    pad = torch.full((blocks * 32, C), -7.0, device=gpu)
    pad[:rows] = x
    want = pad.reshape(blocks, 32, C // 4, 4).permute(0, 2, 1, 3).reshape(-1)
    assert torch.equal(tiled, want)
    back = torch.empty_like(x)
    _lib.check(lib.pwv_tile32_to_rows_f32(tiled.data_ptr(), back.data_ptr(), rows, C, s))
    assert torch.equal(back, x)


def test_sample_condition_cache_sees_in_place_updates(gpu):
    """The tile32 copy of a per-sample condition is reused across the flows of a forward pass; writing to the
    tensor in place (or passing a different one) must not be served from that copy."""
    import torch
    from pwv_amd.modules import WaveNet
    from pwv_amd.variables import VariableStore
    store = VariableStore(device=gpu)
    net = WaveNet(batch_size=2, dilations=[1, 2, 4], filter_width=2, residual_channels=64, dilation_channels=64,
                  skip_channels=128, quantization_channels=1, input_channels=1, use_biases=True, condition_channels=80,
                  use_skip_connection=False, name='scalar', store=store)
    x = torch.randn(2, 100, 1, device=gpu)
    cond = torch.randn(2, 100, 80, device=gpu)
    y0 = net(x, cond)
    assert torch.equal(net(x, cond), y0)
    cond.mul_(0.5)
    y1 = net(x, cond)
    assert not torch.equal(y1, y0)
    assert torch.equal(net(x, cond.clone()), y1)


@pytest.mark.parametrize('seed', range(16))
def test_vocoder_random_configurations(gpu, seed):
    """Seeded sweep over the structural switches of the path (flows, dilation lists incl. d > T, biases, skip
    accumulation, shared nets, repeat / transposed-conv / no conditioning, batch, length, both arithmetics):
    every draw must meet the fp32 bar against the fp64 oracle."""
    rng = np.random.RandomState(1000 + seed)
    n_iaf = int(rng.randint(1, 3))
    dil = [[int(2 ** rng.randint(0, 10)) for _ in range(rng.randint(1, 6))] for _ in range(n_iaf)]
    method = ['repeat', 'transposed_conv', 'none'][int(rng.randint(0, 3))]
    shared = bool(rng.randint(0, 2))
    cfg = O.ModelConfig(dilations=dil, n_iaf=n_iaf, use_biases=bool(rng.randint(0, 2)),
                        use_skip_connection=bool(rng.randint(0, 2)), cond_upsample_method=method, shared_nets=shared)
    weights = O.init_weights(cfg, seed=int(rng.randint(0, 1 << 20)))
    n, length = int(rng.randint(1, 4)), 80 * int(rng.randint(1, 7))
    mel, z = O.synthetic_inputs(n, length, cfg, mel_seed=seed, z_seed=seed + 100)
    want = O.iaf_vocoder_forward(weights, mel, z, cfg)
    precision = ['f32', 'f16x3'][seed % 2]
    got = run_vocoder_hip(cfg, weights, mel, z, gpu, precision=precision)
    scale = max(1.0, float(np.abs(want).max()))
    err = np.abs(got - want).max()
    assert got.shape == want.shape and err <= TOL_F32 * scale, (err, scale, dil, method, shared, precision)


@pytest.mark.parametrize('precision', ['f16x3', 'f32', 'f16'])
@pytest.mark.parametrize('method', ['repeat', 'transposed_conv'])
def test_fused_first_layer_and_fused_head_are_bit_identical(gpu, method, precision, monkeypatch):
    """All three arithmetics (the fp16 storage mode fuses the head behind a per-sample-condition layer too): layer 0 rebuilds h[t] = x[t-1] w0 + x[t] w1 from the scalar input (pwv_layer_args.x_first)
    with the front kernel's own two fp32 operations per channel -- switching the front kernel back on must not change
    a single bit (ragged length, several utterances, dilation of layer 0 > 1 in the second flow, a one-layer net)."""
    from pwv_amd import engine
    cfg = O.ModelConfig(dilations=[[1, 2, 4], [4, 1, 8, 2], [2]], n_iaf=3, cond_upsample_method=method)
    weights = O.init_weights(cfg, seed=21)
    mel, z = O.synthetic_inputs(3, 80 * 7, cfg)
    monkeypatch.setattr(engine, 'FUSE_FIRST', True)
    # (the default of all three arithmetics goes one step further and folds layer 0's convolution onto the scalars: another
    # evaluation order, checked against this one below and in tests/test_gpu_persist.py)
    monkeypatch.setattr(engine, 'FOLD_FIRST', False)
    a = run_vocoder_hip(cfg, weights, mel, z, gpu, precision=precision)
    monkeypatch.setattr(engine, 'FUSE_FIRST', False)
    b = run_vocoder_hip(cfg, weights, mel, z, gpu, precision=precision)
    assert np.array_equal(a, b)
    monkeypatch.setattr(engine, 'FUSE_FIRST', True)
    monkeypatch.setattr(engine, 'FOLD_FIRST', True)
    f = run_vocoder_hip(cfg, weights, mel, z, gpu, precision=precision)
    assert not np.array_equal(a, f) and np.abs(a - f).max() <= (5e-3 if precision == 'f16' else 1e-5)
    assert np.abs(f - O.iaf_vocoder_forward(weights, mel, z, cfg)).max() <= (5e-3 if precision == 'f16' else TOL_F32)
    monkeypatch.setattr(engine, 'FOLD_FIRST', False)
    # ... and the same for the head fused behind the last layer (the gated output stays in registers)
    monkeypatch.setattr(engine, 'FUSE_HEAD', False)
    c = run_vocoder_hip(cfg, weights, mel, z, gpu, precision=precision)
    assert np.array_equal(a, c)
    # (the reduced-precision fp16 storage mode has its own stated tolerance, tests/test_gpu_f16.py)
    assert np.abs(a - O.iaf_vocoder_forward(weights, mel, z, cfg)).max() <= (5e-3 if precision == 'f16' else TOL_F32)


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
def test_one_projection_gemm_for_all_flows_is_bit_identical(gpu, precision, monkeypatch):
    """engine.project_all: the frame-rate projections of every net as one GEMM ahead of the first flow -- each column is the
    same dot product in the same order as in the per-flow GEMMs, so nothing may change (3 flows of different depth, ragged
    batch); and the nets really read the shared result (row stride = all columns)."""
    from pwv_amd import engine
    cfg = O.ModelConfig(dilations=[[1, 2, 4], [4, 1, 8, 2], [2]], n_iaf=3, cond_upsample_method='repeat')
    weights = O.init_weights(cfg, seed=33)
    mel, z = O.synthetic_inputs(3, 80 * 5, cfg)
    seen = []
    real = engine.linear_op

    def spy(x2d, w, b, relu, precision=None):
        seen.append(tuple(w.shape))
        return real(x2d, w, b, relu, precision=precision)
    monkeypatch.setattr(engine, 'linear_op', spy)
    monkeypatch.setattr(engine, 'HOIST_P', True)
    monkeypatch.setattr(engine, 'FUSE_PROLOGUE', False)
    a = run_vocoder_hip(cfg, weights, mel, z, gpu, precision=precision)
    hoisted = [s for s in seen if s[1] == 128 * 2 * (3 + 4 + 1)]
    assert len(hoisted) == 1 and not [s for s in seen if s[1] in (128 * 3, 128 * 4, 128)], seen
    monkeypatch.setattr(engine, 'HOIST_P', False)
    b = run_vocoder_hip(cfg, weights, mel, z, gpu, precision=precision)
    assert np.array_equal(a, b)
    assert np.abs(a - O.iaf_vocoder_forward(weights, mel, z, cfg)).max() <= TOL_F32
    # round 5: range check + dense / relu + that GEMM as ONE launch (pwv_cond_project_f32; split-fp16 arithmetic only) -- the same bits
    # again, and none of the three launches it replaces is enqueued
    monkeypatch.setattr(engine, 'HOIST_P', True)
    monkeypatch.setattr(engine, 'FUSE_PROLOGUE', True)
    del seen[:]
    c = run_vocoder_hip(cfg, weights, mel, z, gpu, precision=precision)
    assert np.array_equal(a, c)
    if precision == 'f16x3':
        assert not seen, seen


def test_one_launch_prologue_is_bit_identical_and_guards_the_range(gpu):
    """pwv_cond_project_f32 against the launches it replaces, through the C ABI: frames = relu(mel @ dense) (models.py:128-130) and
    P = frames @ bank + bias, ragged row counts (row tiles that end inside a tile, fewer rows than one tile), column counts off the
    128-column blocks; and the range guard: one value beyond the limit / one NaN anywhere in the mel raises the flag."""
    import ctypes
    import torch
    from pwv_amd import _lib, engine
    lib = _lib.lib()
    g = torch.Generator().manual_seed(9)
    for m, nout in ((201, 15360), (33, 128 * 9), (7, 260), (2001, 128 * 18)):
        mel = (torch.rand((m, 80), generator=g) * 2 - 1).to(gpu)
        dense = (torch.randn((80, 80), generator=g) * 0.2).to(gpu)
        bank_w = (torch.randn((80, nout), generator=g) * 0.3).to(gpu)
        bank_b = torch.randn((nout,), generator=g).to(gpu)
        want_f = engine.linear_op(mel, dense, None, relu=True, precision='f16x3')
        want_p = engine.linear_op(want_f, bank_w, bank_b, relu=False, precision='f16x3')
        frames = torch.full((m, 80), float('nan'), device=gpu)
        P = torch.full((m, nout), float('nan'), device=gpu)
        w = engine.current_words()
        for poison, raised in ((None, False), ((m // 2, 17, 3.0e4), True), ((m - 1, 79, float('nan')), True)):
            x = mel.clone()
            if poison:
                x[poison[0], poison[1]] = poison[2]
            w.range = 0
            engine.check(lib.pwv_cond_project_f32(x.data_ptr(), dense.data_ptr(), 80, bank_w.data_ptr(), bank_b.data_ptr(), frames.data_ptr(),
                                                  P.data_ptr(), m, 80, nout, 1.0e4, engine.range_flag_ptr(), engine._stream()), 'pwv_cond_project_f32')
            torch.cuda.synchronize()
            assert bool(w.range) == raised, (m, nout, poison)
            w.range = 0
            if not poison:
                assert torch.equal(frames, want_f) and torch.equal(P, want_p), (m, nout)


def test_capturable_noise_sampler_draws_the_same_stream(gpu):
    """pwv_logistic_noise_stream_f32 (the sampler as a node of a HIP graph: {seed, offset, ticket, skip} in device memory, the last block
    of a launch moves the offset on) against pwv_logistic_noise_f32 with the offsets passed by value: launch k draws the range
    [offset + k n, offset + (k + 1) n) of the same stream, for sizes that are and are not multiples of the block; skip leaves z and the
    state alone; the ticket is zero between launches."""
    import torch
    from pwv_amd import engine
    for n_el, seed, off in ((16000, 5, 0), (1000 * 3 + 7, (1 << 55) + 12345, 999983), (1, 9, 3)):
        state = torch.tensor([seed, off, 0, 0], dtype=torch.int64, device=gpu)
        z = torch.empty((1, n_el, 1), dtype=torch.float32, device=gpu)
        for k in range(3):
            engine.logistic_noise_stream_op(z, state)
            want = engine.logistic_noise_op((1, n_el, 1), gpu, seed=seed, offset=off + k * n_el)
            assert torch.equal(z, want)
            assert state.tolist() == [seed, off + (k + 1) * n_el, 0, 0]
        state[3] = 1
        keep = z.clone()
        engine.logistic_noise_stream_op(z, state)
        assert torch.equal(z, keep) and state.tolist() == [seed, off + 3 * n_el, 0, 1]


def test_plain_c_client_runs(gpu, tmp_path):
    """The C99 client (examples/c_abi_smoke.c) drives pwv_causal_conv_f32 and the tile32 converters with raw
    hipMalloc'd pointers and checks them against loops written in C."""
    import subprocess
    from tests.util import build_c_abi_smoke
    exe = str(tmp_path / 'c_abi_smoke')
    res = build_c_abi_smoke(exe)
    assert res.returncode == 0, res.stdout
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert run.returncode == 0, run.stdout
