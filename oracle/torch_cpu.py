"""PyTorch-CPU fp32 restatement of the generation path.  TEST INFRASTRUCTURE ONLY.

Used (a) as a third, independent implementation of the dilated causal convolution
(``torch.nn.functional.conv1d``) in the oracle tests and (b) as the timed
``cpu_baseline`` ("port") leg of ``bench.py`` (TensorFlow is not
installable here, so the CPU baseline is this restatement of modules.py:11-259 /
models.py:23-136 on the host cores).  PARITY UNPINNED upstream, see iaf_oracle.py.

Never imported by the product package.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .iaf_oracle import ModelConfig, net_names


def causal_conv_torch(x: torch.Tensor, f: torch.Tensor, dilation: int) -> torch.Tensor:
    """modules.py:11-43 via conv1d.  x [N,T,Cin], f [W,Cin,Cout] (TF layout) -> [N,T,Cout]."""
    w = f.shape[0]
    xc = x.transpose(1, 2)                                  # NCT
    xc = F.pad(xc, ((w - 1) * dilation, 0))
    y = F.conv1d(xc, f.permute(2, 1, 0).contiguous(), dilation=dilation)
    return y.transpose(1, 2)


def _wavenet(W: Dict[str, torch.Tensor], scope: str, x: torch.Tensor, cond: Optional[torch.Tensor],
             dilations: Sequence[int], use_biases: bool, use_skip: bool) -> torch.Tensor:
    g = lambda n: W[scope + '/' + n]
    cur = causal_conv_torch(x, g('causal_layer/filter'), 1)               # modules.py:179-180
    total = None
    for j, d in enumerate(dilations):
        p = 'dilated_stack/layer%d/' % j
        # filter and gate convs share their input: one conv with 2*D outputs (same sums)
        fg = torch.cat([g(p + 'filter'), g(p + 'gate')], dim=2)
        y = causal_conv_torch(cur, fg, d)                                  # modules.py:213-214
        if cond is not None:                                               # :216-222
            y = y + cond @ torch.cat([g(p + 'gc_filter')[0], g(p + 'gc_gate')[0]], dim=1)
        if use_biases:                                                     # :224-228
            y = y + torch.cat([g(p + 'filter_bias'), g(p + 'gate_bias')])
        D = y.shape[-1] // 2
        out = torch.tanh(y[..., :D]) * torch.sigmoid(y[..., D:])           # :236
        last = j == len(dilations) - 1
        if use_skip or last:
            skip = out @ g(p + 'skip')[0]                                  # :243-244
            if use_biases:
                skip = skip + g(p + 'skip_bias')
            total = skip if (total is None or not use_skip) else total + skip
        if not last:
            tr = out @ g(p + 'dense')[0]                                   # :239-240
            if use_biases:
                tr = tr + g(p + 'dense_bias')
            cur = cur + tr                                                 # :251
    pp = 'postprocessing/'
    t1 = torch.relu(total)                                                 # :148
    c1 = t1 @ g(pp + 'postprocess1')[0]
    if use_biases:
        c1 = c1 + g(pp + 'postprocess1_bias')
    c2 = torch.relu(c1) @ g(pp + 'postprocess2')[0]
    if use_biases:
        c2 = c2 + g(pp + 'postprocess2_bias')
    return c2


def iaf_vocoder_forward_torch(weights: Dict[str, np.ndarray], mel: np.ndarray, z: np.ndarray,
                              cfg: ModelConfig, dtype=torch.float32) -> np.ndarray:
    """models.py:23-78 on torch-CPU (normalisers must be off: default hparams)."""
    assert not (cfg.normalize or cfg.normalize_cond or cfg.normalize_wavenet)
    W = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in weights.items()}
    melt = torch.from_numpy(mel).to(dtype)
    x = torch.from_numpy(z).to(dtype)
    hop = cfg.hop_length
    with torch.no_grad():
        if cfg.cond_upsample_method == 'repeat':                           # models.py:127-133
            c = torch.relu(melt @ W['iaf_vocoder/cond/dense'][0])
            cond = c.repeat_interleave(hop, dim=1)[:, hop // 2: -(hop // 2), :]
        elif cfg.cond_upsample_method == 'transposed_conv':                # models.py:109-124
            cond = melt
            for i, s in enumerate(cfg.strides):
                w = W['iaf_vocoder/cond/transposed_conv_%d_weights' % i][0]    # [s, Cout, Cin]
                n, t, _ = cond.shape
                o = torch.einsum('ntc,joc->ntjo', cond, w)
                cond = torch.relu(o.reshape(n, t * s, w.shape[1]))
            cond = cond[:, hop // 2: -(hop // 2), :]
        else:
            cond = None
        for i in range(cfg.n_iaf):
            outs = [_wavenet(W, 'iaf_vocoder/iaf%d/%s' % (i, net), x, cond, cfg.dilations[i],
                             cfg.use_biases, cfg.use_skip_connection) for net in net_names(cfg)]
            if cfg.shared_nets:
                x = x * outs[0][..., 0:1] + outs[0][..., 1:2]
            else:
                x = x * outs[0] + outs[1]                                  # modules.py:59
    return x.numpy()
