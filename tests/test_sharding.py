"""CPU, world_size 2, gloo: the N>1 utterance-sharding path (scatter mel/z from rank 0, per-rank
forward, gather waveforms).  The per-rank forward here is the oracle -- this test checks the
partition / collective plumbing, which is identical under RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import iaf_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_utts, with_z, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pwv_amd.distributed import generate_sharded, shard_bounds
        cfg = O.ModelConfig(dilations=[[1, 2]], n_iaf=1)
        w = O.init_weights(cfg, seed=3)
        length = 160
        mel, z = O.synthetic_inputs(n_utts, length, cfg)
        seen = []

        def forward(mel_local, z_local):
            seen.append(mel_local.shape[0])
            zz = z_local.numpy() if z_local is not None else np.zeros((mel_local.shape[0], length, 1), np.float32)
            y = O.iaf_vocoder_forward(w, mel_local.numpy(), zz, cfg, dtype=np.float32)
            return torch.from_numpy(y)

        out = generate_sharded(forward, torch.from_numpy(mel) if rank == 0 else None, (mel.shape[1], mel.shape[2]), length,
                               torch.device('cpu'), z=torch.from_numpy(z) if (rank == 0 and with_z) else None)
        lo, hi = shard_bounds(n_utts, world, rank)
        assert seen == ([hi - lo] if hi > lo else [])
        if rank == 0:
            want = O.iaf_vocoder_forward(w, mel, z if with_z else np.zeros_like(z), cfg, dtype=np.float32)
            ret['err'] = float(np.abs(out.numpy() - want).max())
            ret['shape'] = tuple(out.shape)
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_utts,with_z', [(5, True), (4, False), (1, True)])
def test_generate_sharded_gloo_world2(n_utts, with_z):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), n_utts, with_z, ret), nprocs=2, join=True)
    assert ret['shape'] == (n_utts, 160, 1)
    assert ret['err'] == 0.0            # same fp32 oracle on the same rows: bit-identical, order preserved


def test_shard_bounds_cover_and_balance():
    from pwv_amd.distributed import shard_bounds, shard_sizes
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1


# ---- one long utterance over the ranks, sharded in TIME ------------------------------------------
def _time_worker(rank, world, port, n_utts, length, tiles, with_z, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pwv_amd.distributed import generate_time_sharded_ranks
        from pwv_amd.timeshard import chain_halo
        cfg = O.ModelConfig(dilations=[[1, 2, 4], [1, 2]], n_iaf=2)
        w = O.init_weights(cfg, seed=4)
        hop = cfg.hop_length
        halo = chain_halo(cfg.dilations, cfg.filter_width, cfg.n_iaf, hop)
        mel, z = O.synthetic_inputs(n_utts, length, cfg)
        windows = []

        def forward(mel_w, z_w, t0):
            windows.append((t0, mel_w.shape[1]))
            if z_w is None:                 # a forward that draws its own noise takes the window's slice of the one stream
                z_w = torch.from_numpy(z[:, t0:t0 + (mel_w.shape[1] - 1) * hop])
            return torch.from_numpy(O.iaf_vocoder_forward(w, mel_w.numpy(), z_w.numpy().astype(np.float64), cfg).astype(np.float32))

        out = generate_time_sharded_ranks(forward, torch.from_numpy(mel) if rank == 0 else None, cfg.n_mels, length, hop, halo,
                                          torch.device('cpu'), z=torch.from_numpy(z) if (rank == 0 and with_z) else None,
                                          tiles_per_rank=tiles)
        n_shards = min(world, length // hop)          # (a rank beyond the number of frames gets nothing)
        assert len(windows) == (tiles if rank < n_shards else 0) and all(t0 % hop == 0 for t0, _ in windows)
        if rank == 0:
            want = O.iaf_vocoder_forward(w, mel, z.astype(np.float64), cfg).astype(np.float32)
            ret['err'] = float(np.abs(out.numpy() - want).max())
            ret['shape'] = tuple(out.shape)
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_utts,length,tiles,with_z', [(1, 4000, 1, True), (2, 2400, 3, False), (1, 80, 1, True)])
def test_time_sharded_ranks_gloo_world2(n_utts, length, tiles, with_z):
    """Two ranks, each with its half of the time axis plus the flow chain's look-back as lead, reproduce the unsharded
    forward (fp64 oracle per window, compared after rounding to fp32: the same numbers up to the last bit of the fp64
    sums; the BIT-exact statement is made on the GPU, tests/test_gpu_fullsize.py)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_time_worker, args=(2, _free_port(), n_utts, length, tiles if length > 80 else 1, with_z, ret), nprocs=2, join=True)
    assert ret['shape'] == (n_utts, length, 1)
    assert ret['err'] <= 1e-6


def _job_worker(rank, world, port, batch, length, ret, allow_time=True):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pwv_amd.generate import forward_over_ranks
        hop, n_mels, halo = 80, 4, 160
        g = torch.Generator().manual_seed(11)
        mel = torch.rand((batch, 1 + length // hop, n_mels), generator=g)
        calls = []

        def noise_window(n, first_sample, window, first_item):      # counter (first_item + i) * length + first_sample + t
            idx = (torch.arange(n).reshape(n, 1) + first_item) * length + first_sample + torch.arange(window).reshape(1, window)
            return torch.sin(idx.to(torch.float64) * 0.37).to(torch.float32).reshape(n, window, 1)

        def make_model(n, window):                                   # a causal stand-in with a look-back of 100 samples
            def model(m, z):
                calls.append((n, window))
                c = m.mean(dim=2).repeat_interleave(hop, dim=1)[:, hop // 2: hop // 2 + window].reshape(n, window, 1)
                zp = torch.nn.functional.pad(z, (0, 0, 100, 0))
                return z + 0.5 * zp[:, :window] + c
            return model

        out = forward_over_ranks(mel if rank == 0 else None, batch, length, torch.device('cpu'), make_model, noise_window, n_mels, hop, halo,
                                 allow_time_shards=allow_time)
        ret['calls%d' % rank] = list(calls)
        if rank == 0:
            want = make_model(batch, length)(mel, noise_window(batch, 0, length, 0))
            ret['equal'] = bool(torch.equal(out, want))
            ret['shape'] = tuple(out.shape)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('batch,length', [(3, 800), (1, 4000)])
def test_generate_under_a_launcher_shards_utterances_or_time(batch, length):
    """generate.forward_over_ranks (what generate() runs when WORLD_SIZE > 1, generate.py:27-38 being single-device): a batch
    of at least `world` utterances shards by utterance, a smaller one in time; either way rank 0 gets exactly what one
    process would have produced from the same noise stream (stand-in forward: causal, look-back 100 < halo)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_job_worker, args=(2, _free_port(), batch, length, ret), nprocs=2, join=True)
    assert ret['shape'] == (batch, length, 1) and ret['equal']
    if batch >= 2:
        assert ret['calls0'] == [(2, length)] and ret['calls1'] == [(1, length)]
    else:
        assert ret['calls0'] == [(1, 2000)] and ret['calls1'] == [(1, 2000 + 160)]


def test_a_time_global_normaliser_keeps_a_small_batch_on_utterance_shards():
    """normalize* = 'in' reduces over the whole time axis (modules.py:274-284): overlap-and-discard is not exact for it, so
    generate() passes allow_time_shards=False and ONE utterance on two ranks runs whole on rank 0 while rank 1 idles."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_job_worker, args=(2, _free_port(), 1, 4000, ret, False), nprocs=2, join=True)
    assert ret['shape'] == (1, 4000, 1) and ret['equal']
    assert ret['calls0'] == [(1, 4000)] and ret['calls1'] == []
