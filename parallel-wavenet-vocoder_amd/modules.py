"""The reference's op surface (modules.py) on MI355X: causal_conv, LinearIAFLayer, WaveNet,
normalize -- same names, constructor / call signatures, NTC tensors and TF weight layouts,
so the reference's generate.py / models.py read the same.  All arithmetic runs in
libpwv_hip.so (HIP, gfx950); there is no CPU path and no torch-eager fallback for the
default architecture.

  causal_conv        /root/reference/modules.py:11-43
  LinearIAFLayer     /root/reference/modules.py:46-60   (alias IAFLayer, BASELINE.json's name)
  WaveNet            /root/reference/modules.py:64-259
  normalize          /root/reference/modules.py:263-284
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import engine
from .engine import RepeatedCondition
from .variables import VariableStore, current_scope, get_default_store, variable_scope


def causal_conv(value, filter_, dilation, name='causal_conv'):
    """Left-zero-padded dilated 1-D cross-correlation, output length == input length
    (modules.py:11-43).  value [N,T,Cin], filter_ [W,Cin,Cout] -> [N,T,Cout]."""
    return engine.causal_conv_op(value, filter_, int(dilation))


BN_EPS = 1e-3        # tf.layers.batch_normalization default epsilon (modules.py:266)
IN_EPS = 1e-8        # instance_normalization(epsilon=1e-8), modules.py:274


def bn_inference_affine(store: VariableStore, prefix: str, c: int):
    """tf.layers.batch_normalization(training=False) as y = x * a + b with a = gamma / sqrt(moving_variance + eps),
    b = beta - moving_mean * a (variables under `<prefix>/batch_normalization/`, TF's names)."""
    p = prefix + '/batch_normalization/'
    gamma = store.get_variable(p + 'gamma', [c], 'ones')
    beta = store.get_variable(p + 'beta', [c], 'zeros')
    mean = store.get_variable(p + 'moving_mean', [c], 'zeros')
    var = store.get_variable(p + 'moving_variance', [c], 'ones')
    a = gamma / torch.sqrt(var + BN_EPS)
    return a, beta - mean * a


def normalize(input, is_training, method='bn', name='normalize', store: Optional[VariableStore] = None):
    """modules.py:263-270.  Identity unless method is 'bn' / 'in' (the default hparams use '').
    'bn' at inference is a per-channel affine of the moving statistics (pwv_channel_affine_f32; inside a fused WaveNet it
    is folded into the packed weights instead, WaveNet.folded_variables); 'in' normalises over the time axis
    (modules.py:274-284, pwv_instance_norm_f32)."""
    if method not in ('bn', 'in'):
        return input
    store = store or get_default_store()
    from .variables import scoped
    prefix = scoped(name)
    c = input.shape[-1]
    if method == 'bn':
        if is_training:
            raise NotImplementedError('batch-norm training statistics are out of scope (generation path only)')
        a, b = bn_inference_affine(store, prefix, c)
        return engine.channel_affine_op(engine._require_cuda_f32(input, 'input'), a, b)
    beta = store.get_variable(prefix + '/beta', [c], 'zeros')
    gamma = store.get_variable(prefix + '/gamma', [c], 'ones')
    x = engine._require_cuda_f32(input, 'input')
    squeeze = x.dim() == 2
    y = engine.instance_norm_op(x.unsqueeze(0) if squeeze else x, gamma, beta, IN_EPS)
    return y[0] if squeeze else y


class WaveNet(object):
    """ibab-style WaveNet tower with local conditioning (modules.py:64-259).

    Constructor arguments are the reference's (modules.py:78-91).  Extensions (keyword-only):
    ``store`` (variable store; default: the global one), ``input_channels`` (defaults to
    ``quantization_channels`` as in the reference, where the same number is the causal layer's
    input width and the head's output width)."""

    def __init__(self,
                 batch_size,
                 dilations,
                 filter_width,
                 residual_channels,
                 dilation_channels,
                 skip_channels,
                 quantization_channels=2 ** 8,
                 use_biases=False,
                 condition_channels=None,
                 use_skip_connection=True,
                 normalize=None,
                 is_training=True,
                 name='wavenet',
                 *, store: Optional[VariableStore] = None, input_channels: Optional[int] = None,
                 precision: Optional[str] = None):
        self.batch_size = batch_size
        self.dilations = list(dilations)
        self.filter_width = filter_width
        self.residual_channels = residual_channels
        self.dilation_channels = dilation_channels
        self.quantization_channels = quantization_channels
        self.use_biases = use_biases
        self.skip_channels = skip_channels
        self.condition_channels = condition_channels
        self.use_skip_connection = use_skip_connection
        self.normalize = normalize
        self.is_training = is_training
        self.name = name
        self.store = store or get_default_store()
        self.in_channels = input_channels if input_channels is not None else quantization_channels
        self.out_channels = quantization_channels
        self.precision = precision
        # like tf.variable_scope at construction+call time: remember the enclosing scope
        outer = current_scope()
        self.full_scope = (outer + '/' + name) if outer else name

    # -- variables (names as in modules.py:152-164,179,210-248) ------------------------------------
    def device(self):
        return self.store.device

    def _var(self, rel, shape, init=None):
        return self.store.get_variable(self.full_scope + '/' + rel, shape, init)

    def causal_filter(self):
        return self._var('causal_layer/filter', [self.filter_width, self.in_channels, self.residual_channels])

    def layer_variables(self, j: int, with_cond: bool) -> Dict[str, torch.Tensor]:
        p = 'dilated_stack/layer%d/' % j
        W, R, D, S, C = (self.filter_width, self.residual_channels, self.dilation_channels, self.skip_channels,
                         self.condition_channels)
        v = {'filter': self._var(p + 'filter', [W, R, D]), 'gate': self._var(p + 'gate', [W, R, D])}
        if with_cond:
            v['gc_filter'] = self._var(p + 'gc_filter', [1, C, D])
            v['gc_gate'] = self._var(p + 'gc_gate', [1, C, D])
        if self.use_biases:
            v['filter_bias'] = self._var(p + 'filter_bias', [D], 'zeros')
            v['gate_bias'] = self._var(p + 'gate_bias', [D], 'zeros')
        v['dense'] = self._var(p + 'dense', [1, D, R])
        v['skip'] = self._var(p + 'skip', [1, D, S])
        if self.use_biases:
            v['dense_bias'] = self._var(p + 'dense_bias', [R], 'zeros')
            v['skip_bias'] = self._var(p + 'skip_bias', [S], 'zeros')
        return v

    def head_variables(self) -> Dict[str, torch.Tensor]:
        p = 'postprocessing/'
        S, Q = self.skip_channels, self.out_channels
        v = {'postprocess1': self._var(p + 'postprocess1', [1, S, S]),
             'postprocess2': self._var(p + 'postprocess2', [1, S, Q])}
        if self.use_biases:
            v['postprocess1_bias'] = self._var(p + 'postprocess1_bias', [S], 'zeros')
            v['postprocess2_bias'] = self._var(p + 'postprocess2_bias', [Q], 'zeros')
        return v

    # -- batch norm at inference folded into the weights -------------------------------------
    def _bn(self, rel, c):
        return bn_inference_affine(self.store, self.full_scope + '/' + rel, c)

    def _residual_scales(self):
        """S_j: the per-channel factor between the residual stream x_j the reference computes and the stream y_j the fused
        kernels carry (x_j = S_j * y_j): the batch norm of a layer's dense output scales the IDENTITY branch too
        (x_{j+1} = a (x_j + o W + b) + c, modules.py:251-257), which the fused layer cannot express -- but a diagonal scale
        of the stream can be moved into the next layer's filter rows and out of this layer's dense columns."""
        R = self.residual_channels
        S = [torch.ones(R, dtype=torch.float32, device=self.device())]
        for j in range(len(self.dilations) - 1):
            a, _ = self._bn('dilated_stack/layer%d/normalize_dense_output' % j, R)
            S.append(S[-1] * a)
        return S

    def _bn_foldable(self) -> bool:
        key = (self.store.uid, self.store.version)
        if getattr(self, '_bn_ok_key', None) != key:
            S = torch.stack(self._residual_scales()).abs()
            self._bn_ok_key, self._bn_ok = key, bool(((S > 1e-4) & (S < 1e4)).all().item())
        return self._bn_ok

    def folded_variables(self, with_cond: bool):
        """The weights a batch-norm-free net of the same architecture needs to compute what this net with
        normalize='bn' (inference) computes: every normaliser of modules.py:150-160,182,230-234,253-257 is a per-channel
        affine next to a convolution.  Returns {'causal_filter', 'causal_bias', 'layers': [...], 'head': {...}} in TF layouts."""
        R, D, S_, Q = self.residual_channels, self.dilation_channels, self.skip_channels, self.out_channels
        L = len(self.dilations)
        a0, c0 = self._bn('causal_layer/normalize', R)
        out = {'causal_filter': self.causal_filter() * a0, 'causal_bias': c0.contiguous(), 'layers': []}
        S = self._residual_scales()
        zeros = lambda n: torch.zeros(n, dtype=torch.float32, device=self.device())
        for j in range(L):
            v = self.layer_variables(j, with_cond)
            p = 'dilated_stack/layer%d/' % j
            aF, cF = self._bn(p + 'normalize_filter', D)
            aG, cG = self._bn(p + 'normalize_gate', D)
            aS, cS = self._bn(p + 'normalize_skip_output', S_)
            w = {'filter': (v['filter'] * S[j][None, :, None] * aF).contiguous(), 'gate': (v['gate'] * S[j][None, :, None] * aG).contiguous(),
                 'filter_bias': (v.get('filter_bias', zeros(D)) * aF + cF).contiguous(),
                 'gate_bias': (v.get('gate_bias', zeros(D)) * aG + cG).contiguous(),
                 'skip': (v['skip'] * aS).contiguous(), 'skip_bias': (v.get('skip_bias', zeros(S_)) * aS + cS).contiguous()}
            if with_cond:
                w['gc_filter'] = (v['gc_filter'] * aF).contiguous()
                w['gc_gate'] = (v['gc_gate'] * aG).contiguous()
            if j < L - 1:
                _, cD = self._bn(p + 'normalize_dense_output', R)
                w['dense'] = (v['dense'] / S[j]).contiguous()
                w['dense_bias'] = (v.get('dense_bias', zeros(R)) / S[j] + cD / S[j + 1]).contiguous()
            else:       # the last layer's dense output is not used (modules.py:147: only its skip output is)
                w['dense'] = v['dense']
                w['dense_bias'] = v.get('dense_bias', zeros(R))
            out['layers'].append(w)
        hv = self.head_variables()
        a1, c1 = self._bn('postprocessing/normalize_postprocess1', S_)
        a2, c2 = self._bn('postprocessing/normalize_postprocess2', S_)
        out['head'] = {'postprocess1': (hv['postprocess1'] * a1[None, :, None]).contiguous(),
                       'postprocess1_bias': (hv.get('postprocess1_bias', zeros(S_)) + c1 @ hv['postprocess1'][0]).contiguous(),
                       'postprocess2': (hv['postprocess2'] * a2[None, :, None]).contiguous(),
                       'postprocess2_bias': (hv.get('postprocess2_bias', zeros(Q)) + c2 @ hv['postprocess2'][0]).contiguous()}
        return out

    def fused_supported(self, condition) -> bool:
        """The fused HIP layer/head kernels cover the default architecture
        (hparams/default.yaml:22-26): W=2, R=D=64, S=128, no normalisers."""
        ok = (self.filter_width == 2 and self.residual_channels == 64 and self.dilation_channels == 64
              and self.skip_channels == 128 and 1 <= self.out_channels <= 4 and len(self.dilations) >= 1
              and (not self.normalize or (self.normalize == 'bn' and not self.is_training and self.in_channels == 1
                                          and self._bn_foldable())))
        if condition is None:
            return ok
        if isinstance(condition, RepeatedCondition):
            c = self.condition_channels
            return ok and c is not None and c % 8 == 0 and c <= 128
        return ok and self.condition_channels == 80

    # -- network ---------------------------------------------------------------------------------------
    def __call__(self, input_batch, condition_batch=None, verify=None):
        """modules.py:129-166.  Called on its own it returns a verified result like IAFVocoder.__call__ (engine.verified_call:
        rerun on per-layer launches / in exact fp32 if a sticky word is raised); nested in an IAF layer / the vocoder it only
        enqueues and the outermost call verifies."""
        if self.fused_supported(condition_batch):
            return engine.verified_call(lambda prec: engine.run_nets([self], input_batch, condition_batch, precision=prec or self.precision)[0], verify)
        _note_unfused(self)
        return self._call_unfused(input_batch, condition_batch)

    def _call_unfused(self, input_batch, condition_batch):
        """Architectures outside the fused kernels' shape (other widths / channel counts /
        normalisers): composed on the GPU from pwv_causal_conv_f32 (every convolution, 1x1 included) and the
        library's elementwise kernels (pwv_gate_f32, pwv_add_f32, pwv_channel_affine_f32, pwv_instance_norm_f32).
        Exact fp32, no range guard needed.  Slow path, same math (modules.py:129-259)."""
        if isinstance(condition_batch, RepeatedCondition):
            condition_batch = condition_batch.materialize()
        x = engine._require_cuda_f32(input_batch, 'input_batch')
        cond = None if condition_batch is None else engine._require_cuda_f32(condition_batch, 'condition_batch')
        cc = engine.causal_conv_op
        add, bias_add = engine.add_op, lambda t, b: engine.channel_affine_op(t, None, b)
        scope = self.full_scope
        cur = cc(x, self.causal_filter(), 1)
        if self.normalize:
            with variable_scope(scope + '/causal_layer', absolute=True):
                cur = normalize(cur, self.is_training, self.normalize, store=self.store)
        outputs = []
        for j, d in enumerate(self.dilations):
            v = self.layer_variables(j, with_cond=cond is not None)
            f = cc(cur, v['filter'], d)
            g = cc(cur, v['gate'], d)
            if cond is not None:
                f = add(f, cc(cond, v['gc_filter'], 1))
                g = add(g, cc(cond, v['gc_gate'], 1))
            if self.use_biases:
                f = bias_add(f, v['filter_bias'])
                g = bias_add(g, v['gate_bias'])
            lscope = scope + '/dilated_stack/layer%d' % j
            if self.normalize:
                with variable_scope(lscope, absolute=True):
                    f = normalize(f, self.is_training, self.normalize, 'normalize_filter', self.store)
                    g = normalize(g, self.is_training, self.normalize, 'normalize_gate', self.store)
            out = engine.gate_op(f, g)                                   # tanh(f) * sigmoid(g), modules.py:236
            transformed = cc(out, v['dense'], 1)
            skip = cc(out, v['skip'], 1)
            if self.use_biases:
                transformed = bias_add(transformed, v['dense_bias'])
                skip = bias_add(skip, v['skip_bias'])
            dense_out = add(cur, transformed)
            if self.normalize:
                with variable_scope(lscope, absolute=True):
                    skip = normalize(skip, self.is_training, self.normalize, 'normalize_skip_output', self.store)
                    dense_out = normalize(dense_out, self.is_training, self.normalize, 'normalize_dense_output', self.store)
            outputs.append(skip)
            cur = dense_out
        hv = self.head_variables()
        if self.use_skip_connection:
            total = outputs[0]
            for o in outputs[1:]:
                total = add(total, o)
        else:
            total = outputs[-1]
        t1 = engine.channel_affine_op(total, None, None, relu=True)
        pscope = scope + '/postprocessing'
        if self.normalize:
            with variable_scope(pscope, absolute=True):
                t1 = normalize(t1, self.is_training, self.normalize, 'normalize_postprocess1', self.store)
        c1 = cc(t1, hv['postprocess1'], 1)
        t2 = engine.channel_affine_op(c1, None, hv.get('postprocess1_bias'), relu=True)
        if self.normalize:
            with variable_scope(pscope, absolute=True):
                t2 = normalize(t2, self.is_training, self.normalize, 'normalize_postprocess2', self.store)
        c2 = cc(t2, hv['postprocess2'], 1)
        if self.use_biases:
            c2 = bias_add(c2, hv['postprocess2_bias'])
        return c2


_unfused_noted = set()


def _note_unfused(net) -> None:
    """Said once per architecture: this net is outside the fused kernels' shape and runs on the composed path."""
    key = (net.filter_width, net.residual_channels, net.dilation_channels, net.skip_channels, net.out_channels, net.condition_channels, net.normalize or '')
    if key not in _unfused_noted:
        _unfused_noted.add(key)
        import warnings
        warnings.warn('pwv-info: WaveNet %s (W=%d R=%d D=%d S=%d Q=%d C=%s normalize=%r) is outside the shape of the fused kernels (W=2, R=D=64, S=128, '
                      'Q<=4, C=80 per sample or C%%8==0 at frame rate; hparams/default.yaml:22-26): it runs on the composed path -- every op a HIP '
                      'kernel, same math (modules.py:129-259), about 12 launches per layer instead of one launch per flow'
                      % (net.full_scope, net.filter_width, net.residual_channels, net.dilation_channels, net.skip_channels, net.out_channels,
                         net.condition_channels, net.normalize))


class LinearIAFLayer(object):
    """out = input * scaler(input, cond) + shifter(input, cond)   (modules.py:46-60; the scale is
    linear, there is no exp).  When both nets fit the fused kernels they are evaluated side by
    side in the same launches (they share input and condition)."""

    def __init__(self, batch_size, scaler, shifter):
        self.batch_size = batch_size
        self.scaler = scaler
        self.shifter = shifter

    def nets(self):
        return [self.scaler, self.shifter]

    def __call__(self, input, condition=None, verify=None):
        '''
        input = (n, t, h), condition = (n, t, h)
        '''
        return engine.verified_call(lambda prec: self._enqueue(input, condition, prec), verify)

    def _enqueue(self, input, condition, prec=None):
        sc, sh = self.scaler, self.shifter
        both_fused = (isinstance(sc, WaveNet) and isinstance(sh, WaveNet) and sc.fused_supported(condition)
                      and sh.fused_supported(condition) and engine._same_structure(sc, sh)
                      and sc.precision == sh.precision)
        if both_fused and input.shape[-1] == 1 and sc.out_channels == 1:
            return engine.run_flow([sc, sh], input, condition, precision=prec or sc.precision)      # (one launch per flow on the default path)
        if both_fused:
            scale, shift = engine.run_nets([sc, sh], input, condition, precision=prec or sc.precision)
        elif prec is not None and isinstance(sc, WaveNet) and isinstance(sh, WaveNet):
            scale = engine.run_nets([sc], input, condition, precision=prec)[0] if sc.fused_supported(condition) else sc(input, condition)
            shift = engine.run_nets([sh], input, condition, precision=prec)[0] if sh.fused_supported(condition) else sh(input, condition)
        else:
            scale = sc(input, condition)
            shift = sh(input, condition)
        if scale.shape[-1] == 1 and input.shape[-1] == 1 and input.is_cuda:
            x = engine._require_cuda_f32(input, 'input')
            return engine.iaf_affine_op(x, scale, shift, 1)
        return input * scale + shift


IAFLayer = LinearIAFLayer   # the name BASELINE.json uses


class SharedIAFLayer(object):
    """BUILD EXTENSION (BASELINE.json configs[1], "shared mean/var"; the reference has no such
    code): one WaveNet with 1 input and 2 output channels; out = x*y[...,0] + y[...,1]."""

    def __init__(self, batch_size, net):
        self.batch_size = batch_size
        self.net = net

    def nets(self):
        return [self.net]

    def __call__(self, input, condition=None, verify=None):
        return engine.verified_call(lambda prec: self._enqueue(input, condition, prec), verify)

    def _enqueue(self, input, condition, prec=None):
        net = self.net
        if net.fused_supported(condition) and input.shape[-1] == 1 and net.out_channels == 2:
            return engine.run_flow([net], input, condition, precision=prec or net.precision)
        if prec is not None and net.fused_supported(condition):
            y = engine.run_nets([net], input, condition, precision=prec)[0]
        else:
            y = net(input, condition)                       # [N, T, 2]
        x = engine._require_cuda_f32(input, 'input')
        flat = y.reshape(-1)
        return engine.iaf_affine_op(x, flat, flat[1:], 2)
