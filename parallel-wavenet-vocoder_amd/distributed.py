"""Utterance-parallel generation across the GPUs of one node (one process per GPU).

The generation path has no exchange step in its math: utterances are
independent, so they shard across ranks and every rank runs the whole 4-flow stack on its own
utterances with replicated weights (19.4 MB).  The only collectives are the trivial batch
scatter of mel (and optional z) from rank 0 and the gather of waveforms back -- a few MB,
latency-bound -- over RCCL/xGMI (`backend="nccl"` on ROCm) or gloo in the CPU tests.
The reference itself is single-device for generation (generate.py:47-49).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_items: int, world: int) -> List[int]:
    return [shard_bounds(n_items, world, r)[1] - shard_bounds(n_items, world, r)[0] for r in range(world)]


def scatter_batch(full: Optional[torch.Tensor], item_shape, dtype, device, group=None, src: int = 0) -> torch.Tensor:
    """Rank `src` holds `full` [N, *item_shape]; every rank gets its contiguous shard.
    N is broadcast first; shards are padded to equal size for the collective."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([full.shape[0] if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src, group=group)
    n_items = int(n.item())
    sizes = shard_sizes(n_items, world)
    pad = max(sizes) if sizes else 0
    recv = torch.empty((pad,) + tuple(item_shape), dtype=dtype, device=device)
    chunks = None
    if rank == src:
        chunks = []
        for r in range(world):
            lo, hi = shard_bounds(n_items, world, r)
            c = torch.zeros((pad,) + tuple(item_shape), dtype=dtype, device=device)
            c[:hi - lo] = full[lo:hi].to(device)
            chunks.append(c)
    dist.scatter(recv, chunks, src=src, group=group)
    return recv[:sizes[rank]].contiguous()


def gather_batch(local: torch.Tensor, n_items: int, group=None, dst: int = 0) -> Optional[torch.Tensor]:
    """Inverse of scatter_batch: rank `dst` gets [N, *item_shape] in the original order."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(n_items, world)
    pad = max(sizes) if sizes else 0
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, outs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], dim=0)


def generate_sharded(forward_fn: Callable[[torch.Tensor, Optional[torch.Tensor]], torch.Tensor],
                     mel: Optional[torch.Tensor], mel_item_shape, length: int, device,
                     z: Optional[torch.Tensor] = None, group=None) -> Optional[torch.Tensor]:
    """Rank 0 passes all mels [N, t_mel, n_mels] (and optionally all z [N, length, 1]); every rank
    runs ``forward_fn(mel_shard, z_shard_or_None) -> wav [n_local, length, 1]`` on its shard; rank 0
    returns all waveforms [N, length, 1] in input order (other ranks return None)."""
    rank = dist.get_rank(group)
    has_z = torch.tensor([1 if (rank == 0 and z is not None) else 0], dtype=torch.int64, device=device)
    dist.broadcast(has_z, src=0, group=group)
    n = torch.tensor([mel.shape[0] if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=0, group=group)
    n_items = int(n.item())
    mel_local = scatter_batch(mel, mel_item_shape, torch.float32, device, group)
    z_local = scatter_batch(z, (length, 1), torch.float32, device, group) if int(has_z.item()) else None
    if mel_local.shape[0] > 0:
        wav_local = forward_fn(mel_local, z_local)
    else:
        wav_local = torch.empty((0, length, 1), dtype=torch.float32, device=device)
    return gather_batch(wav_local.to(torch.float32), n_items, group)
