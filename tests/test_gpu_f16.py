"""-m gpu: the fp16 mode (precision='f16', PWV_PREC_F16) against the fp64 oracle.

BUILD EXTENSION, not reference parity: the reference is fp32 only (models.py:81-82); BASELINE.json config 5
names an fp16 variant with the tolerance stated against the fp64 restatement (~2e-3).
The residual stream is rounded to fp16 (2^-11 relative) after every layer, so the error grows with depth:
the bars below are for O(1) outputs of the golden configurations and are ~100x looser than TOL_F32."""
import json
import os

import numpy as np
import pytest

from oracle import iaf_oracle as O
from oracle.make_golden import VOCODER_CASES
from tests.util import run_vocoder_hip, small_cfg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

TOL_F16 = 5e-3          # max |y - y_fp64|, outputs O(1)
TOL_F16_RMS = 1e-3


def _fixture(name):
    z = np.load(os.path.join(GOLD, name + '.npz'))
    cfg = O.ModelConfig(**json.loads(str(z['cfg'])))
    return z, cfg, O.init_weights(cfg, seed=int(z['weight_seed']))


@pytest.mark.parametrize('name', sorted(n for n in VOCODER_CASES if 'skipconn' not in n))
def test_golden_vocoder_f16(gpu, name):
    z, cfg, w = _fixture(name)
    got = run_vocoder_hip(cfg, w, z['mel'], z['z'], gpu, precision='f16')
    d = got - z['y']
    assert got.shape == z['y'].shape
    assert np.abs(d).max() <= TOL_F16, np.abs(d).max()
    assert np.sqrt((d ** 2).mean()) <= TOL_F16_RMS


def test_f16_rejects_skip_accumulation(gpu):
    from pwv_amd._lib import PwvError
    z, cfg, w = _fixture('vocoder_skipconn')
    with pytest.raises(PwvError):
        run_vocoder_hip(cfg, w, z['mel'], z['z'], gpu, precision='f16')


@pytest.mark.parametrize('method', ['repeat', 'transposed_conv'])
def test_f16_tracks_f32_path_ragged(gpu, method):
    """Length 720 (22.5 of the 32-sample units; utterance boundaries fall inside units), batch 3, both conditioning modes:
    the fp16 mode stays within its bar of the exact path and is bitwise repeatable."""
    cfg = small_cfg(cond_upsample_method=method)
    w = O.init_weights(cfg, seed=11)
    length = 80 * 9
    mel, zz = O.synthetic_inputs(3, length, cfg)
    ref = run_vocoder_hip(cfg, w, mel, zz, gpu, precision='f32')
    a = run_vocoder_hip(cfg, w, mel, zz, gpu, precision='f16')
    b = run_vocoder_hip(cfg, w, mel, zz, gpu, precision='f16')
    assert np.array_equal(a, b)
    assert np.abs(a - ref).max() <= TOL_F16


def test_f16_layout_round_trip(gpu):
    """pwv_iaf_front_f16 with a one-tap filter exposes the fp16 tile32 order documented in include/pwv_hip.h:
    block u = rows 32u..32u+31 as [8 chunks][32 rows][8 halfs], chunk s*2+h, half q = channel 16s + 8(q>>2) + 4h + (q&3)."""
    import ctypes
    import torch
    from pwv_amd import _lib
    lib = _lib.lib()
    n, t = 2, 45
    x = torch.randn(n, t, device=gpu)
    filt = torch.arange(64, dtype=torch.float32, device=gpu).reshape(1, 1, 64) / 64 + 0.5
    rows = n * t
    blocks = (rows + 31) // 32
    assert lib.pwv_tile32_floats(rows, 64) == blocks * 2048
    out = torch.zeros(blocks * 2048, dtype=torch.float16, device=gpu)
    fp = (ctypes.c_void_p * 1)(filt.data_ptr())
    op = (ctypes.c_void_p * 1)(out.data_ptr())
    _lib.check(lib.pwv_iaf_front_f16(x.data_ptr(), None, None, 1, None, 1, fp, op, n, t, 1, 64,
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    perm = [16 * s + 8 * (q >> 2) + 4 * h + (q & 3) for s in range(4) for h in range(2) for q in range(8)]
    want = torch.zeros(blocks * 32, 64, dtype=torch.float16, device=gpu)
    want[:rows] = (x.reshape(rows, 1) * filt.reshape(64)[perm]).half()
    want = want.reshape(blocks, 32, 8, 8).permute(0, 2, 1, 3).reshape(-1)      # [block][chunk][row][8]
    assert torch.equal(out, want)


def _untile_f16(buf, rows):
    """fp16 tile32 (64 channels) -> [rows, 64] float64 in channel order (include/pwv_hip.h)."""
    blocks = buf.numel() // 2048
    perm = np.array([16 * s + 8 * (q >> 2) + 4 * h + (q & 3) for s in range(4) for h in range(2) for q in range(8)])
    v = buf.cpu().numpy().astype(np.float64).reshape(blocks, 8, 32, 8).transpose(0, 2, 1, 3).reshape(blocks * 32, 64)[:rows]
    out = np.zeros_like(v)
    out[:, perm] = v
    return out


@pytest.mark.parametrize('dense_kind', ['identity', 'random'])
def test_f16_residual_layer_matches_its_own_gated_output(gpu, dense_kind):
    """One residual layer through the C ABI: out = fp16(x + fp16(o) @ fp16(dense) + bias) must follow from the
    kernel's OWN gated output o (same launch arguments, out_mode GATED) -- every operand of that expression is an
    fp16 value, so only a rare rounding flip (fp32 vs fp64 summation order) may differ.  Guards the MFMA-result ->
    VALU read path: a gating instruction issued before the last MFMA had written its accumulator showed up here
    as a ~1 % error in one register of o, in the residual variant only."""
    import ctypes
    import torch
    from pwv_amd import _lib, engine
    from pwv_amd._lib import LayerArgs, check
    from pwv_amd.modules import WaveNet
    from pwv_amd.variables import VariableStore, variable_scope
    lib = _lib.lib()
    n, t = 2, 173
    cfg = O.ModelConfig(dilations=[[1, 2]], n_iaf=1, use_skip_connection=False, use_biases=True, cond_upsample_method='none')
    w = O.init_weights(cfg, seed=5)
    key = 'iaf_vocoder/iaf0/scalar/dilated_stack/layer0/'
    if dense_kind == 'identity':
        w[key + 'dense'] = np.eye(64, dtype=np.float32)[None]
    dense = w[key + 'dense'].astype(np.float16).astype(np.float64)[0]
    bias = w[key + 'dense_bias'].astype(np.float64)
    store = VariableStore(device=gpu)
    store.load_dict(w)
    with variable_scope('iaf_vocoder'), variable_scope('iaf0'):
        net = WaveNet(batch_size=n, dilations=[1, 2], filter_width=2, residual_channels=64, dilation_channels=64,
                      skip_channels=128, quantization_channels=1, input_channels=1, use_biases=True,
                      condition_channels=None, use_skip_connection=False, name='scalar', store=store, precision='f16')
    plan = engine.get_plan(net, 'none', _lib.PREC_F16)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = n * t
    nb = lib.pwv_tile32_floats(rows, 64)
    x = torch.randn(n, t, 1, device=gpu)
    b0, b1, b2 = (torch.zeros(nb, dtype=torch.float16, device=gpu) for _ in range(3))
    filt = (ctypes.c_void_p * 1)(plan.causal_filter.data_ptr())
    hout = (ctypes.c_void_p * 1)(b0.data_ptr())
    check(lib.pwv_iaf_front_f16(x.data_ptr(), None, None, 1, None, 1, filt, hout, n, t, 2, 64, s))
    a = LayerArgs()
    a.G, a.N, a.T, a.dilation, a.skip_init = 1, n, t, 1, 1
    a.proj_row_stride = 128 * 2
    a.precision = _lib.PREC_F16
    a.x_in[0], a.packed[0], a.proj[0] = b0.data_ptr(), plan.packed_layers[0].data_ptr(), plan.proj_b.data_ptr()
    for out, mode in ((b1, _lib.OUT_RESIDUAL), (b2, _lib.OUT_GATED)):
        a.x_out[0], a.out_mode = out.data_ptr(), mode
        check(lib.pwv_wavenet_layer_f32(ctypes.byref(a), s))
    torch.cuda.synchronize()
    h0, h1, og = (_untile_f16(b, rows) for b in (b0, b1, b2))
    want = (h0 + og @ dense + bias).astype(np.float16).astype(np.float64)
    diff = np.abs(h1 - want)
    ulp = np.spacing(np.maximum(np.abs(want), 2.0 ** -14).astype(np.float16)).astype(np.float64)
    assert (diff > 0).mean() < 0.01 and (diff <= ulp).all(), ((diff > 0).mean(), diff.max())


def test_f16_shallow_net_matches_fp16_storage_model(gpu):
    """Three layers: the HIP fp16 mode against the fp64 oracle run on the mode's own storage model (fp16 weights as
    packed, every stored activation rounded to fp16: tests/util.f16_storage_model).  ~10x tighter than TOL_F16 --
    what is left are rounding flips."""
    import torch
    from pwv_amd.modules import WaveNet
    from pwv_amd.variables import VariableStore, variable_scope
    from tests.util import f16_storage_model
    dil = [1, 2, 4]
    cfg = O.ModelConfig(dilations=[dil], n_iaf=1, use_skip_connection=False, use_biases=True, cond_upsample_method='none')
    w = O.init_weights(cfg, seed=5)
    xn = np.random.RandomState(3).randn(2, 400, 1).astype(np.float32)
    store = VariableStore(device=gpu)
    store.load_dict(w)
    with variable_scope('iaf_vocoder'), variable_scope('iaf0'):
        net = WaveNet(batch_size=2, dilations=dil, filter_width=2, residual_channels=64, dilation_channels=64,
                      skip_channels=128, quantization_channels=1, input_channels=1, use_biases=True,
                      condition_channels=None, use_skip_connection=False, name='scalar', store=store, precision='f16')
    got = net(torch.from_numpy(xn).to(gpu), None).cpu().numpy()
    w16, r16 = f16_storage_model(w, cfg)
    model = O.wavenet_forward(w16, 'iaf_vocoder/iaf0/scalar', xn, None, dilations=dil, use_biases=True,
                              use_skip_connection=False, act_round=r16)
    exact = O.wavenet_forward(w, 'iaf_vocoder/iaf0/scalar', xn, None, dilations=dil, use_biases=True,
                              use_skip_connection=False)
    assert np.abs(got - model).max() <= 3e-4 and np.abs(got - exact).max() <= TOL_F16


@pytest.mark.parametrize('seed', range(6))
def test_f16_random_configurations(gpu, seed):
    """The structural sweep of test_gpu_parity.py::test_vocoder_random_configurations in the fp16 mode (no skip
    accumulation there): within the stated fp16 bar of the fp64 oracle, relative to the output scale."""
    rng = np.random.RandomState(2000 + seed)
    n_iaf = int(rng.randint(1, 3))
    dil = [[int(2 ** rng.randint(0, 10)) for _ in range(rng.randint(1, 6))] for _ in range(n_iaf)]
    method = ['repeat', 'transposed_conv', 'none'][int(rng.randint(0, 3))]
    cfg = O.ModelConfig(dilations=dil, n_iaf=n_iaf, use_biases=bool(rng.randint(0, 2)), use_skip_connection=False,
                        cond_upsample_method=method, shared_nets=bool(rng.randint(0, 2)))
    w = O.init_weights(cfg, seed=int(rng.randint(0, 1 << 20)))
    n, length = int(rng.randint(1, 4)), 80 * int(rng.randint(1, 7))
    mel, z = O.synthetic_inputs(n, length, cfg, mel_seed=seed, z_seed=seed + 100)
    want = O.iaf_vocoder_forward(w, mel, z, cfg)
    got = run_vocoder_hip(cfg, w, mel, z, gpu, precision='f16')
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(got - want).max() <= TOL_F16 * scale, (np.abs(got - want).max(), scale, dil, method)
