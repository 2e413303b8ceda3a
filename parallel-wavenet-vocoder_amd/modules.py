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


def normalize(input, is_training, method='bn', name='normalize', store: Optional[VariableStore] = None):
    """modules.py:263-270.  Identity unless method is 'bn' / 'in' (the default hparams use '').
    'bn' at inference is a per-channel affine of the moving statistics; 'in' normalises over
    the time axis (modules.py:274-284).  These two are "next" rows: they run
    as torch device ops, not in the fused kernels."""
    if method not in ('bn', 'in'):
        return input
    store = store or get_default_store()
    with variable_scope(name):
        c = input.shape[-1]
        if method == 'bn':
            if is_training:
                raise NotImplementedError('batch-norm training statistics are out of scope (generation path only)')
            with variable_scope('batch_normalization'):
                from .variables import get_variable
                gamma = get_variable('gamma', [c], 'ones', store)
                beta = get_variable('beta', [c], 'zeros', store)
                mean = get_variable('moving_mean', [c], 'zeros', store)
                var = get_variable('moving_variance', [c], 'ones', store)
            return (input - mean) / torch.sqrt(var + 1e-3) * gamma + beta
        from .variables import get_variable
        beta = get_variable('beta', [c], 'zeros', store)
        gamma = get_variable('gamma', [c], 'ones', store)
        mean = input.mean(dim=1, keepdim=True)
        var = ((input - mean) ** 2).mean(dim=1, keepdim=True)
        return gamma * (input - mean) / torch.sqrt(var + 1e-8) + beta


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

    def fused_supported(self, condition) -> bool:
        """The fused HIP layer/head kernels cover the default architecture
        (hparams/default.yaml:22-26): W=2, R=D=64, S=128, no normalisers."""
        ok = (self.filter_width == 2 and self.residual_channels == 64 and self.dilation_channels == 64
              and self.skip_channels == 128 and not self.normalize and 1 <= self.out_channels <= 4
              and len(self.dilations) >= 1)
        if condition is None:
            return ok
        if isinstance(condition, RepeatedCondition):
            c = self.condition_channels
            return ok and c is not None and c % 8 == 0 and c <= 128
        return ok and self.condition_channels == 80

    # -- network ---------------------------------------------------------------------------------------
    def __call__(self, input_batch, condition_batch=None):
        if self.fused_supported(condition_batch):
            return engine.run_nets([self], input_batch, condition_batch, precision=self.precision)[0]
        return self._call_unfused(input_batch, condition_batch)

    def _call_unfused(self, input_batch, condition_batch):
        """Architectures outside the fused kernels' shape (other widths / channel counts /
        normalisers): composed on the GPU from pwv_causal_conv_f32 (every convolution, 1x1
        included) and torch device elementwise ops.  Slow path, same math (modules.py:129-259)."""
        if isinstance(condition_batch, RepeatedCondition):
            condition_batch = condition_batch.materialize()
        x = engine._require_cuda_f32(input_batch, 'input_batch')
        cond = None if condition_batch is None else engine._require_cuda_f32(condition_batch, 'condition_batch')
        cc = engine.causal_conv_op
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
                f = f + cc(cond, v['gc_filter'], 1)
                g = g + cc(cond, v['gc_gate'], 1)
            if self.use_biases:
                f = f + v['filter_bias']
                g = g + v['gate_bias']
            lscope = scope + '/dilated_stack/layer%d' % j
            if self.normalize:
                with variable_scope(lscope, absolute=True):
                    f = normalize(f, self.is_training, self.normalize, 'normalize_filter', self.store)
                    g = normalize(g, self.is_training, self.normalize, 'normalize_gate', self.store)
            out = torch.tanh(f) * torch.sigmoid(g)
            transformed = cc(out, v['dense'], 1)
            skip = cc(out, v['skip'], 1)
            if self.use_biases:
                transformed = transformed + v['dense_bias']
                skip = skip + v['skip_bias']
            dense_out = cur + transformed
            if self.normalize:
                with variable_scope(lscope, absolute=True):
                    skip = normalize(skip, self.is_training, self.normalize, 'normalize_skip_output', self.store)
                    dense_out = normalize(dense_out, self.is_training, self.normalize, 'normalize_dense_output', self.store)
            outputs.append(skip)
            cur = dense_out
        hv = self.head_variables()
        total = sum(outputs) if self.use_skip_connection else outputs[-1]
        t1 = torch.relu(total)
        pscope = scope + '/postprocessing'
        if self.normalize:
            with variable_scope(pscope, absolute=True):
                t1 = normalize(t1, self.is_training, self.normalize, 'normalize_postprocess1', self.store)
        c1 = cc(t1, hv['postprocess1'], 1)
        if self.use_biases:
            c1 = c1 + hv['postprocess1_bias']
        t2 = torch.relu(c1)
        if self.normalize:
            with variable_scope(pscope, absolute=True):
                t2 = normalize(t2, self.is_training, self.normalize, 'normalize_postprocess2', self.store)
        c2 = cc(t2, hv['postprocess2'], 1)
        if self.use_biases:
            c2 = c2 + hv['postprocess2_bias']
        return c2


class LinearIAFLayer(object):
    """out = input * scaler(input, cond) + shifter(input, cond)   (modules.py:46-60; the scale is
    linear, there is no exp).  When both nets fit the fused kernels they are evaluated side by
    side in the same launches (they share input and condition)."""

    def __init__(self, batch_size, scaler, shifter):
        self.batch_size = batch_size
        self.scaler = scaler
        self.shifter = shifter

    def __call__(self, input, condition=None):
        '''
        input = (n, t, h), condition = (n, t, h)
        '''
        sc, sh = self.scaler, self.shifter
        both_fused = (isinstance(sc, WaveNet) and isinstance(sh, WaveNet) and sc.fused_supported(condition)
                      and sh.fused_supported(condition) and engine._same_structure(sc, sh)
                      and sc.precision == sh.precision)
        if both_fused:
            scale, shift = engine.run_nets([sc, sh], input, condition, precision=sc.precision)
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

    def __call__(self, input, condition=None):
        y = self.net(input, condition)                      # [N, T, 2]
        x = engine._require_cuda_f32(input, 'input')
        flat = y.reshape(-1)
        return engine.iaf_affine_op(x, flat, flat[1:], 2)
