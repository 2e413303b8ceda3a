"""Exact time-axis sharding of one long utterance (section 5 "long-context").

Every net is a strictly causal FIR system: a WaveNet with filter width W and dilations d_j sees
(W-1)*sum(d_j) + (W-1) past input samples (modules.py:168-172), the IAF affine is pointwise
(modules.py:59) and the condition enters pointwise in time (modules.py:216-222).  The whole chain
of flows therefore has a finite halo H = sum over flows of that number (6142 samples for
hparams/default.yaml).  A shard that recomputes H leading samples (rounded up to a multiple of the
hop so that mel frames stay aligned: sample t uses frame (t + hop/2)//hop, models.py:131-133) and
discards them reproduces the unsharded result BIT FOR BIT -- zero communication beyond the initial
scatter of mel / z slices.  This is how a 60 s utterance (BASELINE config 5) spreads over several
GPUs, or is tiled on one.  It breaks if a normaliser with a global time reduction is enabled
(normalize* = 'in', modules.py:279).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch


def chain_halo(dilations: Sequence[Sequence[int]], filter_width: int, n_iaf: int, hop: int) -> int:
    """Samples of left context the whole flow chain needs, rounded up to a multiple of hop."""
    h = sum((filter_width - 1) * sum(dilations[i]) + (filter_width - 1) for i in range(n_iaf))
    return -(-h // hop) * hop


def shard_plan(length: int, n_shards: int, halo: int, hop: int) -> List[Tuple[int, int, int]]:
    """[(compute_start, out_start, out_end)] with every boundary a multiple of hop; shards are
    balanced in OUTPUT samples; compute_start = max(0, out_start - halo)."""
    if length % hop != 0:
        raise ValueError('length must be a multiple of hop')
    frames = length // hop
    n_shards = max(1, min(n_shards, frames))
    plan = []
    for s in range(n_shards):
        a = (frames * s // n_shards) * hop
        b = (frames * (s + 1) // n_shards) * hop
        plan.append((max(0, a - halo), a, b))
    return plan


def generate_time_sharded(forward: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], mel: torch.Tensor,
                          z: torch.Tensor, hop: int, halo: int, n_shards: int,
                          shard_ids: Optional[Sequence[int]] = None) -> torch.Tensor:
    """mel [N, 1 + L/hop, M], z [N, L, 1]; ``forward(mel_shard, z_shard) -> [N, len, 1]`` runs the model
    on a shard (len = z_shard.shape[1], mel_shard has 1 + len/hop frames).  Returns [N, L, 1] (or, with
    ``shard_ids``, only those shards concatenated -- the per-rank piece of a multi-GPU run)."""
    length = z.shape[1]
    plan = shard_plan(length, n_shards, halo, hop)
    outs = []
    for s, (c0, a, b) in enumerate(plan):
        if shard_ids is not None and s not in shard_ids:
            continue
        mel_s = mel[:, c0 // hop: b // hop + 1].contiguous()
        z_s = z[:, c0:b].contiguous()
        y = forward(mel_s, z_s)
        outs.append(y[:, a - c0:])
    return torch.cat(outs, dim=1)


def vocoder_forward_factory(store, precision=None):
    """forward(mel_shard, z_shard) built on IAFVocoder with the CURRENT hparams and a shared variable store
    (one model object per distinct shard length; the weights and packed plans are shared)."""
    from .models import IAFVocoder
    cache = {}

    def forward(mel_s, z_s):
        key = (z_s.shape[0], z_s.shape[1])
        if key not in cache:
            cache[key] = IAFVocoder(batch_size=key[0], length=key[1], store=store, precision=precision)
        return cache[key](None, mel_s, is_training=False, z=z_s)

    return forward
