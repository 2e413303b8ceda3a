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
