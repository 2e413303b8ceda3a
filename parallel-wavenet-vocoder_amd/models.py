"""IAFVocoder: the 4-flow IAF-WaveNet student, generation (forward) path only.

Counterpart of /root/reference/models.py:16-141 with the same constructor / call signature
(`IAFVocoder(batch_size, length)`, `model(wav, melspec, is_training, name='iaf_vocoder')`,
`_upsample_cond(melspec, is_training, strides)`), reading the same global `hparam`.
Training hooks (tensorpack ModelDesc, losses, optimizer: models.py:80-103) are out of scope.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import engine
from .engine import RepeatedCondition
from .hparam import hparam as hp
from .modules import LinearIAFLayer, SharedIAFLayer, WaveNet, normalize
from .variables import VariableStore, get_default_store, get_variable, variable_scope


class IAFVocoder(object):

    def __init__(self, batch_size, length, store: Optional[VariableStore] = None, precision: Optional[str] = None):
        self.batch_size = batch_size
        self.t_mel = 1 + length // hp.signal.hop_length          # models.py:20
        self.length = length
        self.store = store
        self.precision = precision
        self.ema = None
        # Logistic noise (models.py:32-33 draws a fresh sample per sess.run): counter-based sampler, one seed per model
        # object and a running offset so that every call -- eager or graph replay -- continues the same stream.  The
        # default seed is drawn from the OS (PWV_NOISE_SEED pins it); ranks of a sharded job get different seeds that way.
        self.noise_seed = None
        self.noise_offset = 0            # counters drawn so far (a running total: calls may differ in batch size)

    def _seed(self, seed=None):
        import os
        if seed is not None:
            return int(seed)
        if self.noise_seed is None:
            env = os.environ.get('PWV_NOISE_SEED')
            self.noise_seed = int(env) if env else int.from_bytes(os.urandom(7), 'little')
        return self.noise_seed

    def sample_noise(self, n, device, out=None, seed=None):
        """[n, length, 1] Logistic(0,1) noise (models.py:32-33); consecutive calls draw consecutive counter ranges."""
        numel = n * self.length
        z = engine.logistic_noise_op((n, self.length, 1), device, seed=self._seed(seed), offset=self.noise_offset, out=out)
        self.noise_offset += numel
        return z

    # -- network (models.py:23-78) -------------------------------------------------------------------
    def __call__(self, wav, melspec, is_training=False, name='iaf_vocoder', z=None, verify=None):
        """wav is unused by the forward (as in the reference); melspec [N, t_mel, n_mels] on the
        GPU.  ``z`` (optional, [N, length, 1]) replaces the logistic noise sampled at
        models.py:32-33 so results are reproducible against the oracle.

        By default the call returns a VERIFIED result (engine.verified_call): it waits for its launches, and a forward whose
        persistent launch gave up or whose operands left the range of the split-fp16 arithmetic is rerun (per-layer launches /
        exact fp32) on the same noise before anything is handed back.  ``verify=False`` (or PWV_ASYNC=1) only enqueues, like
        the C ABI; the caller then calls ``verify()`` before it reads the result."""
        store = self.store or get_default_store()
        engine.raise_if_range_flag('an earlier call')       # sticky words of un-verified forwards that have completed since
        engine.raise_if_persist_failed()
        melspec = engine._require_cuda_f32(melspec, 'melspec')
        if melspec.dim() != 3 or melspec.shape[1] != self.t_mel or melspec.shape[2] != hp.signal.n_mels:
            raise ValueError('melspec must be [N, %d, %d], got %s' % (self.t_mel, hp.signal.n_mels, tuple(melspec.shape)))
        n = melspec.shape[0]
        if z is None:   # Logistic(0,1) noise, models.py:32-33 (drawn once: a rerun sees the same noise)
            noise = self.sample_noise(n, melspec.device)
        else:
            noise = engine._require_cuda_f32(z, 'z')
            if tuple(noise.shape) != (n, self.length, 1):
                raise ValueError('z must be [%d, %d, 1], got %s' % (n, self.length, tuple(noise.shape)))
        return engine.verified_call(lambda prec: self._forward(store, melspec, noise, is_training, name, prec or self.precision), verify)

    def _forward(self, store, melspec, input, is_training, name, precision):
        """models.py:23-78: condition, then the flows; only enqueues."""
        shared = bool(hp.model.get('shared_nets', False))
        with variable_scope(name):
            flows = []
            for i in range(hp.model.n_iaf):
                with variable_scope('iaf{}'.format(i)):
                    kwargs = dict(
                        batch_size=self.batch_size,
                        dilations=hp.model.dilations[i],
                        filter_width=hp.model.filter_width,
                        residual_channels=hp.model.residual_channels,
                        dilation_channels=hp.model.dilation_channels,
                        skip_channels=hp.model.skip_channels,
                        use_biases=hp.model.use_biases,
                        condition_channels=hp.model.condition_channels,
                        use_skip_connection=hp.model.use_skip_connection,
                        is_training=is_training,
                        normalize=hp.model.normalize_wavenet,
                        store=store, precision=precision)
                    if shared:   # build extension: BASELINE.json configs[1]
                        net = WaveNet(quantization_channels=2, input_channels=1, name='shared', **kwargs)
                        iaf = SharedIAFLayer(batch_size=hp.train.batch_size, net=net)
                    else:
                        # quantization_channels=1: the output is a real value, models.py:42,57
                        scaler = WaveNet(quantization_channels=1, name='scalar', **kwargs)
                        shifter = WaveNet(quantization_channels=1, name='shifter', **kwargs)
                        iaf = LinearIAFLayer(batch_size=hp.train.batch_size, scaler=scaler, shifter=shifter)
                    flows.append(iaf)
            all_nets = [net for iaf in flows for net in iaf.nets()]
            with variable_scope('cond'):
                # (the flows are set up first -- that opens no variable scope of theirs, models.py:26-29 stays ahead of :36-67 in the
                # variable order -- so that the one-launch prologue can project for all of them)
                condition = self._condition(melspec, is_training, strides=[4, 4, 5], store=store, precision=precision, nets=all_nets)   # (n, t, h)
                if hp.model.normalize_cond and condition is not None:
                    if isinstance(condition, RepeatedCondition):
                        condition = condition.materialize()
                    with variable_scope('normalize'):
                        condition = normalize(condition, is_training, hp.model.normalize_cond, store=store)
            # the frame-rate projections of every net depend on the mel only: one GEMM for all flows, ahead of the first
            engine.project_all(all_nets, condition, precision=precision)
            for i, iaf in enumerate(flows):
                input = iaf(input, condition)  # (n, t, h)
                # normalization (identity at the default hparams), models.py:70
                input = normalize(input, is_training, hp.model.normalize, name='normalize{}'.format(i), store=store)
        return input

    def verify(self):
        """For callers of the enqueue-only form (verify=False / PWV_ASYNC=1): wait for the enqueued forwards and raise
        PwvPersistError if a persistent launch gave up (the engine is on per-layer launches from then on: rerun) or
        PwvRangeError if one left the range of the split-fp16 arithmetic (rerun with precision='f32'; the reference computes
        in fp32, models.py:81-82; include/pwv_hip.h "Range guard")."""
        engine.verify_enqueued()

    def _mel_limit(self, weights, store):
        """Largest |mel| for which every operand of the conditioning GEMMs stays inside fp16's range: each stage is
        relu(x @ w), so |out| <= |x|max * max column sum of |w|.  Cached per store version."""
        import torch
        key = (store.uid, store.version, len(weights))
        if getattr(self, '_mel_limit_key', None) != key:
            norms = torch.stack([w.abs().sum(dim=0).max() for w in weights]).cpu().tolist()
            bound, worst = 1.0, 1.0
            for nrm in norms:
                bound *= nrm
                worst = max(worst, bound)
            self._mel_limit_key, self._mel_limit_val = key, engine.F16_LIMIT / worst
        return self._mel_limit_val

    # -- condition upsampling (models.py:105-136) ----------------------------------------------------
    def _condition(self, melspec, is_training, strides, store, precision=None, nets=None):
        """The condition in the form the kernels want: a lazy RepeatedCondition for 'repeat'
        (projected at frame rate inside the nets), a materialised [N, T, C] tensor for
        'transposed_conv', None otherwise."""
        precision = precision or self.precision
        hop = hp.signal.hop_length
        assert (np.prod(np.array(strides)) == hop)                              # models.py:106
        if self.length % hop != 0:
            raise ValueError('length (%d) must be a multiple of hop_length (%d): the crop at models.py:124,133 '
                             'yields (t_mel-1)*hop samples' % (self.length, hop))
        method = hp.model.cond_upsample_method
        n, t_mel, n_mels = melspec.shape
        C = hp.model.condition_channels
        if method == 'transposed_conv':
            cond = melspec.reshape(n * t_mel, n_mels)
            length = t_mel
            input_channels = n_mels
            ws = [get_variable('transposed_conv_{}_weights'.format(i), (1, stride, C, input_channels if i == 0 else C), store=store)
                  for i, stride in enumerate(strides)]
            # kernel width == stride: out[t*s + j, co] = sum_ci in[t, ci] * w[0, j, co, ci]  (a GEMM); the [Cin, s*C] operand
            # is re-laid-out once per weight version, not per forward
            key = (store.uid, store.version, tuple(strides))
            if getattr(self, '_tconv_key', None) != key:
                self._tconv_key = key
                self._tconv_mats = [w[0].permute(2, 0, 1).reshape(w.shape[3], stride * C).contiguous() for w, stride in zip(ws, strides)]
            wmats = self._tconv_mats
            if (precision or engine.DEFAULT_PRECISION) == 'f16x3':
                engine.range_check_op(melspec, self._mel_limit(wmats, store))
            for i, stride in enumerate(strides):
                wmat = wmats[i]
                cond = engine.linear_op(cond, wmat, None, relu=True, precision=precision)   # models.py:118-120
                input_channels = C
                length *= stride
                cond = cond.reshape(n * length, C)
                if hp.model.normalize_cond:
                    cond = normalize(cond.reshape(n, length, C), is_training, hp.model.normalize_cond,
                                     name='normalize_transposed_conv_{}'.format(i), store=store).reshape(n * length, C)
            cond = cond.reshape(n, length, C)
            return engine.crop_time_op(cond, length - hop, hop // 2)            # models.py:124
        elif method == 'repeat':
            w = get_variable('dense', [1, n_mels, C], store=store)
            split = (precision or engine.DEFAULT_PRECISION) == 'f16x3'
            if nets and not hp.model.normalize_cond:
                # range check + dense / relu + the projections of every net of the forward: ONE launch (bit-identical to the three)
                fused = engine.repeat_condition_with_projections(nets, melspec, w[0], hop, self.length, precision,
                                                                 self._mel_limit([w[0]], store) if split else None)
                if fused is not None:
                    return fused
            if split:
                engine.range_check_op(melspec, self._mel_limit([w[0]], store))
            frames = engine.linear_op(melspec.reshape(n * t_mel, n_mels), w[0], None, relu=True,
                                      precision=precision)                              # models.py:128-130
            return RepeatedCondition(frames.reshape(n, t_mel, C), hop, hop // 2, self.length)      # models.py:131-133
        return None

    def _upsample_cond(self, melspec, is_training, strides):
        """Reference signature (models.py:105): returns the upsampled condition [N, T, C] tensor."""
        melspec = engine._require_cuda_f32(melspec, 'melspec')
        with variable_scope('iaf_vocoder'):
            with variable_scope('cond'):
                cond = self._condition(melspec, is_training, strides, self.store or get_default_store())
        if isinstance(cond, RepeatedCondition):
            cond = cond.materialize()
        return cond
