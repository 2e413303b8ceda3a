"""Utterance-parallel (and, for one long utterance, time-parallel) generation across the GPUs of one node, one process per GPU.

The generation path has no exchange step in its math: utterances are
independent, so they shard across ranks and every rank runs the whole 4-flow stack on its own
utterances with replicated weights (19.4 MB).  The only collectives are the trivial batch
scatter of mel (and optional z) from rank 0 and the gather of waveforms back -- a few MB,
latency-bound -- over RCCL/xGMI (`backend="nccl"` on ROCm) or gloo in the CPU tests.
The reference itself is single-device for generation (generate.py:47-49).

A batch smaller than the number of ranks (BASELINE config 5: ONE 60 s utterance) shards in TIME instead
(generate_time_sharded_ranks): the flow chain is causal with a finite look-back (6142 samples for hparams/default.yaml,
timeshard.chain_halo), so rank r computes the r-th slice of every utterance plus that many leading samples and throws
the lead away -- exactly the unsharded result, again with no collective in the data path.
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


def generate_time_sharded_ranks(forward_fn: Callable[[torch.Tensor, Optional[torch.Tensor], int], torch.Tensor],
                                mel: Optional[torch.Tensor], n_mels: int, length: int, hop: int, halo: int, device,
                                z: Optional[torch.Tensor] = None, group=None, tiles_per_rank: int = 1) -> Optional[torch.Tensor]:
    """One (or a few) LONG utterances over all ranks, sharded in time with overlap-and-discard (timeshard.py).
    Rank 0 passes mel [N, 1 + length/hop, n_mels] (and optionally z [N, length, 1]).  Rank r runs
    ``forward_fn(mel_window, z_window_or_None, first_sample) -> [N, window, 1]`` on the r-th window -- its share
    [a, b) of the time axis preceded by `halo` samples of lead (a multiple of hop; clipped at 0) -- and keeps the last
    b - a samples; rank 0 returns [N, length, 1].  `first_sample` lets a forward that samples its own noise draw the
    window's slice of ONE counter-based stream, so that the lead of a window repeats what the window before it saw.
    `tiles_per_rank` > 1 cuts a rank's share into that many tiles run one after the other (memory), each with its own lead.
    Scatter and gather are the only collectives (windows are padded to equal size for them)."""
    from .timeshard import shard_plan
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    meta = torch.tensor([mel.shape[0] if rank == 0 else 0, 1 if (rank == 0 and z is not None) else 0], dtype=torch.int64, device=device)
    dist.broadcast(meta, src=0, group=group)
    n, has_z = int(meta[0].item()), bool(int(meta[1].item()))
    plan = shard_plan(length, world, halo, hop)                 # (window start, share start, share end) per rank; ranks beyond get nothing
    max_win = max(b - c0 for c0, _, b in plan)
    max_out = max(b - a for _, a, b in plan)

    def scatter_windows(full, per_sample, item_dim):
        width = max_win // hop + 1 if not per_sample else max_win
        recv = torch.empty((n, width, item_dim), dtype=torch.float32, device=device)
        chunks = None
        if rank == 0:
            chunks = []
            for r in range(world):
                c = torch.zeros((n, width, item_dim), dtype=torch.float32, device=device)
                if r < len(plan):
                    c0, _, b = plan[r]
                    piece = full[:, c0:b] if per_sample else full[:, c0 // hop: b // hop + 1]
                    c[:, :piece.shape[1]] = piece.to(device)
                chunks.append(c)
        dist.scatter(recv, chunks, src=0, group=group)
        return recv

    mel_w = scatter_windows(mel, False, n_mels)
    z_w = scatter_windows(z, True, 1) if has_z else None
    if rank < len(plan):
        c0, a, b = plan[rank]
        frames = length // hop
        k = max(1, min(int(tiles_per_rank), (b - a) // hop))
        pieces = []
        for i in range(k):                                        # tiles of the share [a, b), each with its own lead inside the window
            ta = a + ((b - a) // hop * i // k) * hop
            tb = a + ((b - a) // hop * (i + 1) // k) * hop
            t0 = max(c0, ta - halo)
            m = mel_w[:, (t0 - c0) // hop: (tb - c0) // hop + 1].contiguous()
            zz = z_w[:, t0 - c0: tb - c0].contiguous() if z_w is not None else None
            pieces.append(forward_fn(m, zz, t0)[:, ta - t0:])
        y = torch.cat(pieces, dim=1).to(torch.float32)
        assert y.shape[1] == b - a and frames * hop == length
    else:
        y = torch.empty((n, 0, 1), dtype=torch.float32, device=device)
    buf = torch.zeros((n, max_out, 1), dtype=torch.float32, device=device)
    buf[:, :y.shape[1]] = y
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, outs, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([outs[r][:, :plan[r][2] - plan[r][1]] for r in range(len(plan))], dim=1)
