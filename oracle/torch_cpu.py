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


def flow_halo(cfg: ModelConfig, i: int) -> int:
    """Past input samples flow i's WaveNets see: (W-1) * (sum of dilations + 1) (modules.py:168-172; default.yaml:12-21 -> 1023 / 3069)."""
    return (cfg.filter_width - 1) * (sum(cfg.dilations[i]) + 1)


_G: dict = {}      # what the chunk workers of iaf_vocoder_forward_torch_chunked read (inherited by fork, or shared by threads)


def _chunk_task(t):
    """One (flow, chunk): the flow's two nets + affine on x[lo:b], of which [a:b] is kept.  `cond` is either the frame-rate mel
    rows the slice needs ('repeat': dense + relu + gather happen here) or the per-sample condition slice itself."""
    i, lo, a, xs, kind, payload, frame0 = t
    W, cfg = _G['W'], _G['cfg']
    hop = cfg.hop_length
    with torch.no_grad():
        xs = torch.from_numpy(xs)
        cs = None
        if kind == 'frames':
            frames = torch.relu(torch.from_numpy(payload) @ W['iaf_vocoder/cond/dense'][0])
            idx = (torch.arange(lo, lo + xs.shape[1]) + hop // 2) // hop - frame0
            cs = frames[:, idx, :]
        elif kind == 'samples':
            cs = torch.from_numpy(payload)
        outs = [_wavenet(W, 'iaf_vocoder/iaf%d/%s' % (i, net), xs, cs, cfg.dilations[i], cfg.use_biases, cfg.use_skip_connection)
                for net in net_names(cfg)]
        y = xs * outs[0][..., 0:1] + outs[0][..., 1:2] if cfg.shared_nets else xs * outs[0] + outs[1]
        return a, y[:, a - lo:].numpy().copy()


class ChunkedForward:
    """The same function as iaf_vocoder_forward_torch, evaluated flow by flow in TIME CHUNKS of `chunk` output samples, each
    preceded by the flow's own look-back (flow_halo, recomputed and discarded: every net is a causal FIR system and the affine is
    pointwise, so the kept samples are the unchunked ones up to the conv primitive's summation order).  Two reasons, both about
    being a FAIR cpu baseline (bench.py `cpu_baseline`): a chunk's working set ([chunk + halo, 128] floats
    per op) stays in a core's cache instead of streaming 41 MB tensors through DRAM at 160000 samples, and the chunks of a flow
    are independent, so `workers` cores each take whole chunks with single-threaded ops instead of splitting every small conv
    across all cores.  mode 'process': a fork()ed pool (no GIL; the caller must not have initialised HIP or an OpenMP team in
    this process -- bench.py runs it in a clean child); 'thread': a thread pool (the GIL is released inside the ops)."""

    def __init__(self, weights: Dict[str, np.ndarray], cfg: ModelConfig, chunk: int = 4000, workers: int = 1, mode: str = 'thread'):
        assert not (cfg.normalize or cfg.normalize_cond or cfg.normalize_wavenet)
        self.cfg, self.chunk, self.workers, self.mode = cfg, int(chunk), int(workers), mode
        torch.set_num_threads(1)
        _G['W'] = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in weights.items()}
        _G['cfg'] = cfg
        self.pool = None
        if workers > 1:
            if mode == 'process':
                import multiprocessing as mp
                self.pool = mp.get_context('fork').Pool(workers)
                self._map = lambda f, it: self.pool.map(f, it, chunksize=1)
            else:
                from concurrent.futures import ThreadPoolExecutor
                self.pool = ThreadPoolExecutor(max_workers=workers)
                self._map = lambda f, it: list(self.pool.map(f, it))
        else:
            self._map = lambda f, it: [f(t) for t in it]

    def close(self):
        if self.pool is not None:
            if self.mode == 'process':
                self.pool.close()
                self.pool.join()
            else:
                self.pool.shutdown()
            self.pool = None

    def __call__(self, mel: np.ndarray, z: np.ndarray) -> np.ndarray:
        cfg, chunk = self.cfg, self.chunk
        W = _G['W']
        hop = cfg.hop_length
        x = np.ascontiguousarray(z, dtype=np.float32)
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        length = x.shape[1]
        full_cond = None
        if cfg.cond_upsample_method == 'transposed_conv':
            with torch.no_grad():
                c = torch.from_numpy(mel)
                for i, s in enumerate(cfg.strides):
                    w = W['iaf_vocoder/cond/transposed_conv_%d_weights' % i][0]
                    nn, t, _ = c.shape
                    c = torch.relu(torch.einsum('ntc,joc->ntjo', c, w).reshape(nn, t * s, w.shape[1]))
                full_cond = c[:, hop // 2: -(hop // 2), :].numpy()
        for i in range(cfg.n_iaf):
            h = flow_halo(cfg, i)
            tasks = []
            for a in range(0, length, chunk):
                b, lo = min(a + chunk, length), max(0, a - h)
                if cfg.cond_upsample_method == 'repeat':
                    f0, f1 = (lo + hop // 2) // hop, (b - 1 + hop // 2) // hop
                    tasks.append((i, lo, a, x[:, lo:b], 'frames', mel[:, f0:f1 + 1], f0))
                elif full_cond is not None:
                    tasks.append((i, lo, a, x[:, lo:b], 'samples', full_cond[:, lo:b], 0))
                else:
                    tasks.append((i, lo, a, x[:, lo:b], 'none', None, 0))
            out = np.empty_like(x)
            for a, y in self._map(_chunk_task, tasks):
                out[:, a:a + y.shape[1]] = y
            x = out
        return x


def iaf_vocoder_forward_torch_chunked(weights: Dict[str, np.ndarray], mel: np.ndarray, z: np.ndarray, cfg: ModelConfig,
                                      chunk: int = 4000, workers: int = 1, mode: str = 'thread') -> np.ndarray:
    """One call of ChunkedForward (tests; bench.py keeps the object, and with it the worker pool, across calls)."""
    prev = torch.get_num_threads()
    f = ChunkedForward(weights, cfg, chunk, workers, mode)
    try:
        return f(mel, z)
    finally:
        f.close()
        torch.set_num_threads(prev)
