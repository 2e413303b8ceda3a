"""CPU tests that pin the ORACLE itself (the reference ships no tests / golden vectors, SURVEY.md
section 8c): independent formulations must agree, analytic known answers must hold, and the
committed golden fixtures must reproduce."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import iaf_oracle as O
from oracle.make_golden import VOCODER_CASES, weights_digest
from oracle.torch_cpu import causal_conv_torch, iaf_vocoder_forward_torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('N', [1, 3])
@pytest.mark.parametrize('T', [10, 33, 37, 64])
@pytest.mark.parametrize('d', [1, 3, 4, 8, 16, 512])
@pytest.mark.parametrize('W', [2, 3])
def test_literal_vs_direct_vs_torch(N, T, d, W):
    """KAT 1: the op-by-op time_to_batch restatement (modules.py:11-43), the closed form and
    torch's conv1d agree in float64 to 1e-12."""
    rng = np.random.RandomState(N * 1000 + T * 10 + d + W)
    x = rng.randn(N, T, 5)
    f = rng.randn(W, 5, 7)
    a = O.causal_conv_literal(x, f, d)
    b = O.causal_conv_direct(x, f, d)
    c = causal_conv_torch(torch.from_numpy(x), torch.from_numpy(f), d).numpy()
    assert a.shape == (N, T, 7)
    assert np.abs(a - b).max() <= 1e-12
    assert np.abs(a - c).max() <= 1e-12


def test_golden_causal_conv():
    z = np.load(os.path.join(GOLD, 'causal_conv.npz'))
    for i in range(6):
        y = O.causal_conv_direct(z['x%d' % i].astype(np.float64), z['f%d' % i].astype(np.float64), int(z['d%d' % i]))
        assert np.abs(y - z['y%d' % i]).max() <= 1e-12


@pytest.mark.parametrize('name', sorted(VOCODER_CASES))
def test_golden_vocoder(name):
    z = np.load(os.path.join(GOLD, name + '.npz'))
    cfg = O.ModelConfig(**json.loads(str(z['cfg'])))
    w = O.init_weights(cfg, seed=int(z['weight_seed']))
    assert weights_digest(w) == str(z['weights_sha256'])          # generator drift guard
    if name == 'vocoder_default_0p25s':                            # full model: check fp64 via the literal conv on 1 flow only
        y, flows = O.iaf_vocoder_forward(w, z['mel'], z['z'], cfg, return_flows=True)
    else:
        y, flows = O.iaf_vocoder_forward(w, z['mel'], z['z'], cfg, conv=O.causal_conv_literal, return_flows=True)
    assert np.abs(y - z['y']).max() <= 1e-11
    assert np.abs(flows[0] - z['flow0']).max() <= 1e-11


def test_fp32_roundoff_budget():
    """The whole 4-flow stack in float32 stays ~1e-6 from float64: the 2e-5 parity bar leaves >10x
    headroom for a different summation order and fast exp/rcp."""
    z = np.load(os.path.join(GOLD, 'vocoder_default_0p25s.npz'))
    cfg = O.ModelConfig()
    w = O.init_weights(cfg, seed=2)
    y32 = O.iaf_vocoder_forward(w, z['mel'], z['z'], cfg, dtype=np.float32)
    yt = iaf_vocoder_forward_torch(w, z['mel'], z['z'], cfg)
    assert np.abs(y32 - z['y']).max() < 5e-6
    assert np.abs(yt - z['y']).max() < 5e-6      # the timed cpu_baseline port computes the same function


@pytest.mark.parametrize('method', ['repeat', 'transposed_conv'])
def test_chunked_cpu_port_is_the_same_function(method):
    """bench.py's cpu_baseline evaluates the port flow by flow in time chunks with each flow's look-back recomputed, one chunk
    per worker (oracle/torch_cpu.py ChunkedForward): the same function as the unchunked port, chunk boundaries off the hop grid
    and worker pools (threads; fork()ed processes) included."""
    from oracle.torch_cpu import iaf_vocoder_forward_torch_chunked, flow_halo
    cfg = O.ModelConfig(dilations=[[1, 2, 4, 8, 16], [1, 2, 4, 8, 16, 32, 64]], n_iaf=2, cond_upsample_method=method)
    assert flow_halo(O.ModelConfig(), 0) == 1024 and flow_halo(O.ModelConfig(), 3) == 3070      # SURVEY 8c KAT 5: RF - 1
    w = O.init_weights(cfg, seed=4)
    mel, z = O.synthetic_inputs(2, 1600, cfg)
    want = iaf_vocoder_forward_torch(w, mel, z, cfg)
    for chunk, workers, mode in ((400, 1, 'thread'), (250, 3, 'thread'), (560, 2, 'process')):
        got = iaf_vocoder_forward_torch_chunked(w, mel, z, cfg, chunk=chunk, workers=workers, mode=mode)
        assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6, (chunk, workers, mode)


def _one_net_cfg(**kw):
    base = dict(dilations=[[1, 2, 4, 8]], n_iaf=1)
    base.update(kw)
    return O.ModelConfig(**base)


def test_causality():
    """KAT 2 (README.md:17, triangular Jacobian): perturbing z[t0] / mel frame f0 leaves earlier outputs
    bit-identical."""
    cfg = O.ModelConfig(dilations=[[1, 2, 4], [1, 2, 4]], n_iaf=2)
    w = O.init_weights(cfg)
    mel, z = O.synthetic_inputs(1, 400, cfg)
    y0 = O.iaf_vocoder_forward(w, mel, z, cfg)
    z2 = z.copy()
    z2[0, 250, 0] += 1.0
    y1 = O.iaf_vocoder_forward(w, mel, z2, cfg)
    assert np.array_equal(y0[0, :250], y1[0, :250]) and not np.array_equal(y0[0, 250:], y1[0, 250:])
    mel2 = mel.copy()
    mel2[0, 3, :] += 0.5          # frame 3 conditions samples t with (t+40)//80 == 3, i.e. t in [200, 280)
    y2 = O.iaf_vocoder_forward(w, mel2, z, cfg)
    assert np.array_equal(y0[0, :200], y2[0, :200]) and not np.array_equal(y0[0, 200:], y2[0, 200:])


def test_zero_weights_kat():
    """KAT 3: all matrices zero, biases b: every net outputs postprocess2_bias; IAF -> z*bs + bb."""
    cfg = _one_net_cfg()
    w = {k: np.zeros_like(v) for k, v in O.init_weights(cfg).items()}
    w['iaf_vocoder/iaf0/scalar/postprocessing/postprocess2_bias'][:] = 0.75
    w['iaf_vocoder/iaf0/shifter/postprocessing/postprocess2_bias'][:] = -0.25
    mel, z = O.synthetic_inputs(2, 160, cfg)
    y = O.iaf_vocoder_forward(w, mel, z, cfg)
    assert np.allclose(y, z.astype(np.float64) * 0.75 - 0.25, atol=1e-15)


def test_impulse_delay_kat():
    """KAT 4: a one-hot filter tap k delays by exactly (W-1-k)*d samples; the left edge is zeros."""
    for d in (1, 3, 8):
        x = np.random.RandomState(d).randn(1, 40, 2)
        f = np.zeros((2, 2, 2))
        f[0] = np.eye(2)          # tap 0 multiplies x[t-d]
        y = O.causal_conv(x, f, d)
        assert np.array_equal(y[0, d:], x[0, :-d]) and np.all(y[0, :d] == 0)
        f = np.zeros((2, 2, 2))
        f[1] = np.eye(2)          # tap 1 multiplies x[t]
        assert np.array_equal(O.causal_conv(x, f, d), x)


def test_receptive_field_kat():
    """KAT 5: RF-1 = 1024 (10-layer towers) / 3070 (30-layer tower); whole chain halo 6142."""
    d10 = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
    assert O.receptive_field(2, d10) - 1 == 1024
    assert O.receptive_field(2, d10 * 3) - 1 == 3070
    # empirical: an impulse at t0 reaches exactly RF-1 samples ahead through a net with positive weights
    dil = [1, 2, 4]
    rf = O.receptive_field(2, dil)
    cfg = O.ModelConfig(dilations=[dil], n_iaf=1, cond_upsample_method='none')
    w = {k: np.abs(v) * 0.05 + 1e-3 for k, v in O.init_weights(cfg).items()}   # positive, far from saturation
    x = np.zeros((1, 40, 1))
    x2 = x.copy()
    x2[0, 5, 0] = 1.0
    kw = dict(dilations=dil, use_biases=True, use_skip_connection=False)
    a = O.wavenet_forward(w, 'iaf_vocoder/iaf0/scalar', x, None, **kw)
    b = O.wavenet_forward(w, 'iaf_vocoder/iaf0/scalar', x2, None, **kw)
    changed = np.nonzero(np.abs(a - b)[0, :, 0] > 0)[0]
    assert changed.min() == 5 and changed.max() == 5 + rf - 1


def test_cond_alignment_kat():
    """KAT 6: repeat upsampling: sample t uses frame (t+40)//80; the first / last 40 samples use frames
    0 / t_mel-1 (models.py:131-133)."""
    cfg = O.ModelConfig()
    w = {'iaf_vocoder/cond/dense': np.eye(80, dtype=np.float32)[None]}
    t_mel, hop = 5, 80
    mel = np.arange(1, t_mel + 1, dtype=np.float32)[None, :, None] * np.ones((1, 1, 80), np.float32)
    c = O.upsample_cond_repeat(w, mel, hop)
    assert c.shape == (1, (t_mel - 1) * hop, 80)
    t = np.arange(c.shape[1])
    assert np.array_equal(c[0, :, 0], ((t + 40) // 80 + 1).astype(np.float64))
    assert np.all(c[0, :40, 0] == 1) and np.all(c[0, -40:, 0] == t_mel)


def test_transposed_conv_kat():
    """KAT 7: conv2d_transpose with kernel width == stride (models.py:110-120) equals
    torch conv_transpose1d with weight[ci, co, j] = w_tf[0, j, co, ci]."""
    cfg = O.ModelConfig(cond_upsample_method='transposed_conv')
    w = O.init_weights(cfg, seed=9)
    mel, _ = O.synthetic_inputs(2, 240, cfg)
    want = O.upsample_cond_transposed(w, mel, 80, (4, 4, 5))
    c = torch.from_numpy(mel.astype(np.float64)).transpose(1, 2)            # [N, C, T]
    for i, s in enumerate((4, 4, 5)):
        wt = torch.from_numpy(w['iaf_vocoder/cond/transposed_conv_%d_weights' % i].astype(np.float64))[0]   # [s, Cout, Cin]
        c = torch.relu(F.conv_transpose1d(c, wt.permute(2, 1, 0).contiguous(), stride=s))
    got = c.transpose(1, 2).numpy()[:, 40:-40, :]
    assert got.shape == want.shape == (2, 240, 80)
    assert np.abs(got - want).max() <= 1e-12


@pytest.mark.parametrize('use_skip', [False, True])
@pytest.mark.parametrize('use_biases', [False, True])
def test_structure_variants_literal_conv(use_skip, use_biases):
    """KAT 8: skip-connection / biases on and off: direct and literal conv give the same net output."""
    cfg = _one_net_cfg(use_skip_connection=use_skip, use_biases=use_biases)
    w = O.init_weights(cfg, seed=4)
    mel, z = O.synthetic_inputs(2, 160, cfg)
    a = O.iaf_vocoder_forward(w, mel, z, cfg)
    b = O.iaf_vocoder_forward(w, mel, z, cfg, conv=O.causal_conv_literal)
    assert np.abs(a - b).max() <= 1e-12 and np.isfinite(a).all()


def test_variable_inventory_matches_survey():
    """4,848,392 parameters in repeat mode and the TF names of section 8 f-1."""
    cfg = O.ModelConfig()
    shapes = O.variable_shapes(cfg)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 4848392
    assert shapes['iaf_vocoder/cond/dense'] == (1, 80, 80)
    assert shapes['iaf_vocoder/iaf3/shifter/dilated_stack/layer29/gc_gate'] == (1, 80, 64)
    assert shapes['iaf_vocoder/iaf0/scalar/causal_layer/filter'] == (2, 1, 64)
    assert shapes['iaf_vocoder/iaf0/scalar/postprocessing/postprocess2'] == (1, 128, 1)
    t = O.variable_shapes(O.ModelConfig(cond_upsample_method='transposed_conv'))
    assert t['iaf_vocoder/cond/transposed_conv_2_weights'] == (1, 5, 80, 80)
    assert sum(int(np.prod(s)) for s in t.values()) == 4848392 + 83200 - 6400


def test_instance_norm_and_bn_identity_defaults():
    x = np.random.RandomState(0).randn(2, 50, 4)
    assert O.normalize(x, '', {}, 's') is x and O.normalize(x, None, {}, 's') is x
    y = O.normalize(x, 'in', {}, 's')
    assert np.allclose(y.mean(axis=1), 0, atol=1e-12) and np.allclose(y.var(axis=1), 1, atol=1e-6)
    assert np.allclose(O.normalize(x, 'bn', {}, 's'), x / np.sqrt(1 + 1e-3))


@pytest.mark.parametrize('method', ['repeat', 'transposed_conv'])
def test_window_restatement_used_by_the_full_size_gpu_tests(method):
    """tests/test_gpu_fullsize.py checks the END of a full-size run against the oracle run on a late window
    [s, L): pin that window logic (mel frame slicing, halo discard) on the oracle itself."""
    from pwv_amd.timeshard import chain_halo
    cfg = O.ModelConfig(dilations=[[1, 2, 4, 8], [1, 2, 4, 8, 16]], n_iaf=2, cond_upsample_method=method)
    hop = cfg.hop_length
    halo = chain_halo(cfg.dilations, cfg.filter_width, cfg.n_iaf, hop)
    assert halo == 80
    w = O.init_weights(cfg, seed=5)
    L, keep = 1600, 400
    mel, z = O.synthetic_inputs(2, L, cfg)
    full = O.iaf_vocoder_forward(w, mel, z, cfg)
    s = L - keep - halo
    win = O.iaf_vocoder_forward(w, mel[1:2, s // hop:], z[1:2, s:], cfg)
    assert np.abs(full[1:2, L - keep:] - win[:, halo:]).max() <= 1e-12
    K = 800
    pre = O.iaf_vocoder_forward(w, mel[:, :K // hop + 1], z[:, :K], cfg)
    assert np.abs(full[:, :K - hop // 2] - pre[:, :K - hop // 2]).max() <= 1e-12
    # one hop less of halo than the receptive field needs is NOT enough (the check has teeth)
    win2 = O.iaf_vocoder_forward(w, mel[1:2, (s + hop) // hop:], z[1:2, s + hop:], cfg)
    assert np.abs(full[1:2, s + hop + 16:s + hop + 30] - win2[:, 16:30]).max() > 1e-9
