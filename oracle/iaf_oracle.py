"""CPU oracle for the IAF-WaveNet student generation path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``parallel-wavenet-vocoder_amd``) never imports anything under
``oracle/`` and fails loudly when the HIP library is missing.

PARITY UNPINNED: the reference (andabi/parallel-wavenet-vocoder) ships no tests,
fixtures or golden vectors, and its arithmetic lives in TensorFlow 1.x
(``requirements.txt:1`` ``tensorflow >= 1.4``, un-vendored, not installed here), so
the reference itself cannot be run in this container.  The pins are therefore:
  * two independent restatements of the causal convolution that must agree in
    float64 (``causal_conv_literal`` follows the reference's pad / reshape /
    transpose / VALID-conv / inverse sequence op by op; ``causal_conv_direct`` is the
    closed form),
  * a third independent implementation (``torch.nn.functional.conv1d``) in the tests,
  * the analytic known-answer tests in ``tests/test_oracle.py``.

Everything is channels-last ``[N, T, C]`` and all weights are in TensorFlow layout
``[width, Cin, Cout]`` exactly as the reference creates them.  All functions take a
``dtype`` (float64 for the parity reference, float32 to measure fp32 round-off).

file:line citations are into /root/reference/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np

Weights = Dict[str, np.ndarray]


# --------------------------------------------------------------------------------------
# causal_conv  (modules.py:11-43)
# --------------------------------------------------------------------------------------
def _conv1d_valid(x: np.ndarray, f: np.ndarray) -> np.ndarray:
    """tf.nn.conv1d(value, filters, stride=1, padding='VALID'): NWC cross-correlation.

    x [B, T, Cin], f [W, Cin, Cout] -> [B, T-W+1, Cout];  out[b,t] = sum_k x[b,t+k] @ f[k].
    """
    w = f.shape[0]
    t_out = x.shape[1] - w + 1
    out = np.zeros((x.shape[0], max(t_out, 0), f.shape[2]), dtype=x.dtype)
    for k in range(w):
        out += x[:, k:k + t_out, :] @ f[k]
    return out


def causal_conv_literal(value: np.ndarray, filter_: np.ndarray, dilation: int) -> np.ndarray:
    """Op-by-op restatement of modules.py:11-43 (time_to_batch / conv1d VALID / batch_to_time)."""
    n, t, c = value.shape
    width = filter_.shape[0]
    if dilation > 1:
        # time_to_batch, modules.py:12-18
        pad_elements = dilation - 1 - (t + dilation - 1) % dilation          # :14
        padded = np.pad(value, [(0, 0), (0, pad_elements), (0, 0)])            # :15
        reshaped = padded.reshape(-1, dilation, c)                             # :16
        transposed = reshaped.transpose(1, 0, 2)                               # :17
        transformed = transposed.reshape(n * dilation, -1, c)                  # :18
        # left pad so the VALID conv is causal, modules.py:32-33
        padded2 = np.pad(transformed, [(0, 0), (width - 1, 0), (0, 0)])
        conv = _conv1d_valid(padded2, filter_)
        # batch_to_time, modules.py:20-25
        cout = conv.shape[2]
        prepared = conv.reshape(dilation, -1, cout)                            # :22
        transposed2 = prepared.transpose(1, 0, 2)                              # :23
        restored = transposed2.reshape(conv.shape[0] // dilation, -1, cout)    # :24-25
        return restored[:, :t, :]                                              # :37-39
    padded = np.pad(value, [(0, 0), (width - 1, 0), (0, 0)])                   # :41
    return _conv1d_valid(padded, filter_)                                      # :42


def causal_conv_direct(value: np.ndarray, filter_: np.ndarray, dilation: int) -> np.ndarray:
    """Closed form of modules.py:11-43:
    y[n,t,:] = sum_k x[n, t-(W-1-k)*d, :] @ f[k], with x[t<0] = 0; len(y) == len(x)."""
    n, t, _ = value.shape
    width = filter_.shape[0]
    out = np.zeros((n, t, filter_.shape[2]), dtype=value.dtype)
    for k in range(width):
        shift = (width - 1 - k) * dilation
        if shift >= t:
            continue
        out[:, shift:, :] += value[:, :t - shift, :] @ filter_[k]
    return out


causal_conv = causal_conv_direct


# --------------------------------------------------------------------------------------
# normalisers  (modules.py:263-284) -- identity at default hparams (default.yaml:30-32)
# --------------------------------------------------------------------------------------
def normalize(x: np.ndarray, method: Optional[str], weights: Weights, scope: str,
              create_missing: bool = True) -> np.ndarray:
    """modules.py:263-270.  'bn' = tf.layers.batch_normalization in inference mode
    (moving_mean / moving_variance, epsilon 1e-3); 'in' = instance_normalization
    (modules.py:274-284: moments over the time axis, epsilon 1e-8); else identity."""
    if method == 'bn':
        c = x.shape[-1]
        g = weights.get(scope + '/batch_normalization/gamma', np.ones(c))
        b = weights.get(scope + '/batch_normalization/beta', np.zeros(c))
        mu = weights.get(scope + '/batch_normalization/moving_mean', np.zeros(c))
        var = weights.get(scope + '/batch_normalization/moving_variance', np.ones(c))
        return ((x - mu.astype(x.dtype)) / np.sqrt(var.astype(x.dtype) + x.dtype.type(1e-3))
                * g.astype(x.dtype) + b.astype(x.dtype))
    if method == 'in':
        c = x.shape[-1]
        beta = weights.get(scope + '/beta', np.zeros(c)).astype(x.dtype)
        gamma = weights.get(scope + '/gamma', np.ones(c)).astype(x.dtype)
        mean = x.mean(axis=1, keepdims=True)                       # modules.py:279
        var = ((x - mean) ** 2).mean(axis=1, keepdims=True)
        normalized = (x - mean) / ((var + x.dtype.type(1e-8)) ** x.dtype.type(.5))  # :282
        return gamma * normalized + beta                           # :283
    return x


# --------------------------------------------------------------------------------------
# WaveNet  (modules.py:64-259)
# --------------------------------------------------------------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def wavenet_forward(weights: Weights, scope: str, input_batch: np.ndarray,
                    condition_batch: Optional[np.ndarray], dilations: Sequence[int],
                    use_biases: bool, use_skip_connection: bool,
                    normalize_method: Optional[str] = None,
                    dtype=np.float64, conv=causal_conv_direct, act_round=None) -> np.ndarray:
    """WaveNet.__call__ (modules.py:129-166) with _create_causal_layer (:174-183) and
    _create_dilation_layer (:185-259).  ``scope`` is the TF variable-scope prefix
    (e.g. 'iaf_vocoder/iaf0/scalar').

    ``act_round`` (default None = the reference's arithmetic) models the reduced-precision STORAGE of the
    fp16 build extension (PWV_PREC_F16, no reference counterpart): it is applied wherever that mode keeps an
    activation as fp16 -- the residual stream after the causal layer and after every layer, the gated output
    feeding dense / skip, relu(total) feeding postprocess1, and a per-sample condition."""
    rnd = act_round if act_round is not None else (lambda v: v)
    W = {k: v.astype(dtype) for k, v in weights.items() if k.startswith(scope + '/')}
    g = lambda name: W[scope + '/' + name]
    x = input_batch.astype(dtype)
    cond = None if condition_batch is None else rnd(condition_batch.astype(dtype))

    # causal layer: no bias even with use_biases (modules.py:179-183)
    cur = rnd(conv(x, g('causal_layer/filter'), 1))
    if normalize_method:
        cur = normalize(cur, normalize_method, W, scope + '/causal_layer/normalize')

    outputs = []
    for j, d in enumerate(dilations):
        p = 'dilated_stack/layer%d/' % j
        conv_filter = conv(cur, g(p + 'filter'), d)                       # :213
        conv_gate = conv(cur, g(p + 'gate'), d)                           # :214
        if cond is not None:                                              # :216-222
            conv_filter = conv_filter + cond @ g(p + 'gc_filter')[0]
            conv_gate = conv_gate + cond @ g(p + 'gc_gate')[0]
        if use_biases:                                                    # :224-228
            conv_filter = conv_filter + g(p + 'filter_bias')
            conv_gate = conv_gate + g(p + 'gate_bias')
        if normalize_method:                                              # :230-234
            conv_filter = normalize(conv_filter, normalize_method, W, scope + '/' + p + 'normalize_filter')
            conv_gate = normalize(conv_gate, normalize_method, W, scope + '/' + p + 'normalize_gate')
        out = rnd(np.tanh(conv_filter) * _sigmoid(conv_gate))             # :236
        transformed = out @ g(p + 'dense')[0]                             # :239-240
        skip_output = out @ g(p + 'skip')[0]                              # :243-244
        if use_biases:                                                    # :246-250
            transformed = transformed + g(p + 'dense_bias')
            skip_output = skip_output + g(p + 'skip_bias')
        dense_output = rnd(cur + transformed)                             # :251
        if normalize_method:                                              # :253-257
            skip_output = normalize(skip_output, normalize_method, W, scope + '/' + p + 'normalize_skip_output')
            dense_output = normalize(dense_output, normalize_method, W, scope + '/' + p + 'normalize_dense_output')
        outputs.append(skip_output)
        cur = dense_output

    pp = 'postprocessing/'
    total = sum(outputs) if use_skip_connection else outputs[-1]          # :147
    t1 = rnd(np.maximum(total, 0))                                        # :148
    if normalize_method:
        t1 = normalize(t1, normalize_method, W, scope + '/' + pp + 'normalize_postprocess1')
    c1 = t1 @ g(pp + 'postprocess1')[0]                                   # :152-153
    if use_biases:
        c1 = c1 + g(pp + 'postprocess1_bias')                             # :154-156
    t2 = np.maximum(c1, 0)                                                # :157
    if normalize_method:
        t2 = normalize(t2, normalize_method, W, scope + '/' + pp + 'normalize_postprocess2')
    c2 = t2 @ g(pp + 'postprocess2')[0]                                   # :161-162
    if use_biases:
        c2 = c2 + g(pp + 'postprocess2_bias')                             # :163-165
    return c2


def linear_iaf(weights: Weights, scope: str, x: np.ndarray, cond: Optional[np.ndarray],
               dilations, use_biases, use_skip_connection, normalize_method=None,
               dtype=np.float64, conv=causal_conv_direct, act_round=None) -> np.ndarray:
    """LinearIAFLayer.__call__ (modules.py:53-60): out = input*scaler(...) + shifter(...)."""
    kw = dict(dilations=dilations, use_biases=use_biases, use_skip_connection=use_skip_connection,
              normalize_method=normalize_method, dtype=dtype, conv=conv, act_round=act_round)
    scale = wavenet_forward(weights, scope + '/scalar', x, cond, **kw)    # :57
    shift = wavenet_forward(weights, scope + '/shifter', x, cond, **kw)   # :58
    return x.astype(dtype) * scale + shift                                # :59


def shared_iaf(weights: Weights, scope: str, x, cond, dilations, use_biases, use_skip_connection,
               dtype=np.float64, conv=causal_conv_direct, act_round=None) -> np.ndarray:
    """BUILD EXTENSION (BASELINE.json configs[1], "shared mean/var"; no reference code):
    one WaveNet per flow (scope '<iaf>/shared') with 1 input channel and 2 output
    channels; channel 0 = scale, channel 1 = shift; out = x*scale + shift."""
    y = wavenet_forward(weights, scope + '/shared', x, cond, dilations=dilations,
                        use_biases=use_biases, use_skip_connection=use_skip_connection,
                        dtype=dtype, conv=conv, act_round=act_round)
    return x.astype(dtype) * y[..., 0:1] + y[..., 1:2]


# --------------------------------------------------------------------------------------
# condition upsampling  (models.py:105-136)
# --------------------------------------------------------------------------------------
def upsample_cond_repeat(weights: Weights, mel: np.ndarray, hop: int, dtype=np.float64,
                         scope: str = 'iaf_vocoder/cond') -> np.ndarray:
    """models.py:127-133.  1x1 conv (no bias) + relu, each frame repeated hop times
    (tile on the channel axis + reshape == repeat along time), crop [hop//2 : -hop//2]."""
    w = weights[scope + '/dense'].astype(dtype)[0]
    c = np.maximum(mel.astype(dtype) @ w, 0)                               # :129-130
    n, t_mel, ch = c.shape
    tiled = np.tile(c, (1, 1, hop)).reshape(n, t_mel * hop, ch)            # :131-132
    return tiled[:, hop // 2: -(hop // 2), :]                              # :133


def frame_cond_repeat(weights: Weights, mel: np.ndarray, dtype=np.float64,
                      scope: str = 'iaf_vocoder/cond') -> np.ndarray:
    """Frame-rate part of models.py:128-130 only (before the repeat)."""
    w = weights[scope + '/dense'].astype(dtype)[0]
    return np.maximum(mel.astype(dtype) @ w, 0)


def upsample_cond_transposed(weights: Weights, mel: np.ndarray, hop: int, strides: Sequence[int],
                             normalize_cond: Optional[str] = None, dtype=np.float64,
                             scope: str = 'iaf_vocoder/cond') -> np.ndarray:
    """models.py:109-124.  conv2d_transpose with kernel width == stride, SAME padding
    => non-overlapping: out[n, t*s + j, co] = sum_ci in[n, t, ci] * w[0, j, co, ci]; relu
    after every stage; crop [hop//2 : -hop//2] at the end."""
    assert int(np.prod(np.array(strides))) == hop                          # :106
    cond = mel.astype(dtype)
    for i, s in enumerate(strides):
        w = weights[scope + '/transposed_conv_%d_weights' % i].astype(dtype)   # [1, s, Cout, Cin]
        n, t, _ = cond.shape
        # out[n, t, j, co] = sum_ci cond[n,t,ci] * w[0,j,co,ci]
        o = np.einsum('ntc,joc->ntjo', cond, w[0])
        cond = np.maximum(o.reshape(n, t * s, w.shape[2]), 0)              # :118-120
        cond = normalize(cond, normalize_cond, weights, scope + '/normalize_transposed_conv_%d' % i)
    return cond[:, hop // 2: -(hop // 2), :]                               # :124


# --------------------------------------------------------------------------------------
# IAFVocoder.__call__  (models.py:23-78)
# --------------------------------------------------------------------------------------
class ModelConfig:
    """The subset of hparams the generation path reads (default.yaml:1-33)."""

    def __init__(self, dilations=None, filter_width=2, residual_channels=64, dilation_channels=64,
                 skip_channels=128, condition_channels=80, use_biases=True,
                 use_skip_connection=False, n_iaf=4, normalize='', normalize_cond='',
                 normalize_wavenet='', cond_upsample_method='repeat', n_mels=80, hop_length=80,
                 strides=(4, 4, 5), shared_nets=False):
        d10 = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
        self.dilations = dilations if dilations is not None else [d10, d10, d10, d10 * 3]
        self.filter_width = filter_width
        self.residual_channels = residual_channels
        self.dilation_channels = dilation_channels
        self.skip_channels = skip_channels
        self.condition_channels = condition_channels
        self.use_biases = use_biases
        self.use_skip_connection = use_skip_connection
        self.n_iaf = n_iaf
        self.normalize = normalize
        self.normalize_cond = normalize_cond
        self.normalize_wavenet = normalize_wavenet
        self.cond_upsample_method = cond_upsample_method
        self.n_mels = n_mels
        self.hop_length = hop_length
        self.strides = tuple(strides)
        self.shared_nets = shared_nets   # build extension (BASELINE config 2)

    @staticmethod
    def from_hparam(hp) -> 'ModelConfig':
        m = hp['model']
        return ModelConfig(dilations=[list(d) for d in m['dilations']], filter_width=m['filter_width'],
                           residual_channels=m['residual_channels'], dilation_channels=m['dilation_channels'],
                           skip_channels=m['skip_channels'], condition_channels=m['condition_channels'],
                           use_biases=m['use_biases'], use_skip_connection=m['use_skip_connection'],
                           n_iaf=m['n_iaf'], normalize=m.get('normalize', ''),
                           normalize_cond=m.get('normalize_cond', ''),
                           normalize_wavenet=m.get('normalize_wavenet', ''),
                           cond_upsample_method=m['cond_upsample_method'],
                           n_mels=hp['signal']['n_mels'], hop_length=hp['signal']['hop_length'],
                           shared_nets=bool(m.get('shared_nets', False)))


def net_names(cfg: ModelConfig) -> List[str]:
    return ['shared'] if cfg.shared_nets else ['scalar', 'shifter']


def variable_shapes(cfg: ModelConfig) -> Dict[str, tuple]:
    """TF variable names and shapes the generation graph creates (SURVEY.md section 8 f-1;
    models.py:114-115,128; modules.py:152-164,179,210-248)."""
    s: Dict[str, tuple] = {}
    C, M = cfg.condition_channels, cfg.n_mels
    if cfg.cond_upsample_method == 'transposed_conv':
        cin = M
        for i, st in enumerate(cfg.strides):
            s['iaf_vocoder/cond/transposed_conv_%d_weights' % i] = (1, st, C, cin)
            cin = C
    elif cfg.cond_upsample_method == 'repeat':
        s['iaf_vocoder/cond/dense'] = (1, M, C)
    has_cond = cfg.cond_upsample_method in ('repeat', 'transposed_conv')
    Wd, R, D, S = cfg.filter_width, cfg.residual_channels, cfg.dilation_channels, cfg.skip_channels
    for i in range(cfg.n_iaf):
        for net in net_names(cfg):
            p = 'iaf_vocoder/iaf%d/%s/' % (i, net)
            q_in, q_out = (1, 2) if cfg.shared_nets else (1, 1)
            s[p + 'causal_layer/filter'] = (Wd, q_in, R)
            for j, _ in enumerate(cfg.dilations[i]):
                l = p + 'dilated_stack/layer%d/' % j
                s[l + 'filter'] = (Wd, R, D)
                s[l + 'gate'] = (Wd, R, D)
                if has_cond:
                    s[l + 'gc_filter'] = (1, C, D)
                    s[l + 'gc_gate'] = (1, C, D)
                if cfg.use_biases:
                    s[l + 'filter_bias'] = (D,)
                    s[l + 'gate_bias'] = (D,)
                s[l + 'dense'] = (1, D, R)
                s[l + 'skip'] = (1, D, S)
                if cfg.use_biases:
                    s[l + 'dense_bias'] = (R,)
                    s[l + 'skip_bias'] = (S,)
            s[p + 'postprocessing/postprocess1'] = (1, S, S)
            s[p + 'postprocessing/postprocess2'] = (1, S, q_out)
            if cfg.use_biases:
                s[p + 'postprocessing/postprocess1_bias'] = (S,)
                s[p + 'postprocessing/postprocess2_bias'] = (q_out,)
    return s


def init_weights(cfg: ModelConfig, seed: int = 2, bias_std: float = 0.1) -> Weights:
    """Deterministic synthetic weights: glorot-uniform per tensor in TF
    layout (tf.get_variable's default initializer; fan_in = prod(shape[:-2])*shape[-2],
    fan_out = prod(shape[:-2])*shape[-1]), biases N(0, bias_std^2) so the bias paths are
    exercised (TF default is zeros)."""
    rng = np.random.RandomState(seed)
    out: Weights = {}
    for name, shape in variable_shapes(cfg).items():
        if len(shape) == 1:
            out[name] = (rng.randn(*shape) * bias_std).astype(np.float32)
        else:
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan_in, fan_out = rf * shape[-2], rf * shape[-1]
            limit = math.sqrt(6.0 / (fan_in + fan_out))
            out[name] = rng.uniform(-limit, limit, size=shape).astype(np.float32)
    return out


def synthetic_inputs(n: int, length: int, cfg: ModelConfig, mel_seed: int = 0, z_seed: int = 1):
    """mel ~ U(-1,1) [N, 1+L/hop, n_mels]; z = log u - log1p(-u),
    u ~ U(1e-7, 1-1e-7) (the Logistic(0,1) sampler of models.py:32-33), [N, L, 1]."""
    assert length % cfg.hop_length == 0
    t_mel = 1 + length // cfg.hop_length                                    # models.py:20
    mel = np.random.RandomState(mel_seed).uniform(-1, 1, size=(n, t_mel, cfg.n_mels)).astype(np.float32)
    u = np.random.RandomState(z_seed).uniform(1e-7, 1 - 1e-7, size=(n, length, 1))
    z = (np.log(u) - np.log1p(-u)).astype(np.float32)
    return mel, z


def iaf_vocoder_forward(weights: Weights, mel: np.ndarray, z: np.ndarray, cfg: ModelConfig,
                        dtype=np.float64, conv=causal_conv_direct, return_flows: bool = False, act_round=None):
    """IAFVocoder.__call__ (models.py:23-78) with the logistic noise ``z`` given explicitly
    (models.py:32-33 samples it; TF's RNG stream is not reproducible).  ``act_round``: see wavenet_forward."""
    hop = cfg.hop_length
    if cfg.cond_upsample_method == 'transposed_conv':
        cond = upsample_cond_transposed(weights, mel, hop, cfg.strides, cfg.normalize_cond or None, dtype)
    elif cfg.cond_upsample_method == 'repeat':
        cond = upsample_cond_repeat(weights, mel, hop, dtype)
    else:
        cond = None                                                         # models.py:134-135
    if cond is not None and cfg.normalize_cond:                             # models.py:27-29
        # models.py:27-29: tf.variable_scope('normalize') around normalize(), whose own default scope is 'normalize' again
        cond = normalize(cond, cfg.normalize_cond, weights, 'iaf_vocoder/cond/normalize/normalize')
    x = z.astype(dtype)
    flows = []
    for i in range(cfg.n_iaf):                                              # models.py:34-70
        kw = dict(dilations=cfg.dilations[i], use_biases=cfg.use_biases,
                  use_skip_connection=cfg.use_skip_connection, dtype=dtype, conv=conv, act_round=act_round)
        if cfg.shared_nets:
            x = shared_iaf(weights, 'iaf_vocoder/iaf%d' % i, x, cond, **kw)
        else:
            x = linear_iaf(weights, 'iaf_vocoder/iaf%d' % i, x, cond,
                           normalize_method=cfg.normalize_wavenet or None, **kw)
        x = normalize(x, cfg.normalize or None, weights, 'iaf_vocoder/normalize%d' % i)   # :70
        flows.append(x)
    return (x, flows) if return_flows else x


def receptive_field(filter_width: int, dilations: Sequence[int]) -> int:
    """modules.py:168-172 (commented-out helper): (W-1)*sum(d) + 1 + (W-1)."""
    return (filter_width - 1) * sum(dilations) + 1 + (filter_width - 1)
