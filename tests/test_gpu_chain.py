"""-m gpu: the forward with the two nets' launch chains kept on their streams across the flows (engine.run_flow_chain: one
fork, one join, a device-memory handshake + the IAF affine per chain at every flow boundary) against the flow-by-flow form
with stream-level joins -- BIT-identical (same kernels, same operations), eager and under HIP-graph replay; and the loud
fallback when a chain waits in vain.  Reference: the flow loop models.py:34-70, the affine modules.py:59."""
import ctypes

import numpy as np
import pytest

from oracle import iaf_oracle as O
from tests.util import TOL_F32, set_hparams

pytestmark = pytest.mark.gpu


@pytest.fixture()
def chain_knobs():
    from pwv_amd import engine
    saved = (engine.CHAIN_FLOWS, engine.PERSIST)
    yield engine
    engine.CHAIN_FLOWS, engine.PERSIST = saved


def _model(gpu, cfg, length, precision=None):
    import torch
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    set_hparams(cfg)
    w = O.init_weights(cfg, seed=2)
    store = VariableStore(device=gpu)
    store.load_dict(w)
    mel, z = O.synthetic_inputs(cfg_batch(cfg), length, cfg)
    return IAFVocoder(batch_size=mel.shape[0], length=length, store=store, precision=precision), w, mel, z, torch.from_numpy(mel).to(gpu), torch.from_numpy(z).to(gpu)


def cfg_batch(cfg):
    return getattr(cfg, '_batch', 1)


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
@pytest.mark.parametrize('length,batch,cond', [(160000, 1, 'repeat'), (8000, 3, 'repeat'), (4000, 2, 'none')])
def test_chained_flows_are_bit_identical_to_joined_flows(gpu, chain_knobs, length, batch, cond, precision):
    import torch
    from pwv_amd.graph import GraphedVocoder
    engine = chain_knobs
    engine.PERSIST = False           # (short inputs would take the persistent launch, which has no chains)
    cfg = O.ModelConfig() if cond == 'repeat' else O.ModelConfig(cond_upsample_method='none')
    cfg._batch = batch
    model, w, mel, z, mel_t, z_t = _model(gpu, cfg, length, precision)
    engine.CHAIN_FLOWS = False
    model(None, mel_t, is_training=False, z=z_t)                      # (creates the variables, packs the plans)
    y0 = model(None, mel_t, is_training=False, z=z_t).clone()
    engine.CHAIN_FLOWS = True
    before = engine.CHAIN_FORWARDS
    for _ in range(3):
        y1 = model(None, mel_t, is_training=False, z=z_t)
        model.verify()
        assert torch.equal(y0, y1)
    assert engine.CHAIN_FORWARDS == before + 3                        # it really was the chained form
    graphed = GraphedVocoder(model)
    for _ in range(3):
        y2 = graphed(mel_t, z=z_t)
        model.verify()
        assert torch.equal(y0, y2)
    K = min(length, 4000)
    want = O.iaf_vocoder_forward(w, mel[:, :K // 80 + 1], z[:, :K], cfg)
    assert np.abs(y1.cpu().numpy()[:, :K - 40] - want[:, :K - 40]).max() <= TOL_F32


def test_a_chain_that_waits_in_vain_is_loud_and_falls_back(gpu, chain_knobs):
    """The handshake reports a wait that ran into its bound through a sticky word in pinned host memory; the host then
    raises and uses stream-level joins from then on.  (The word is poked from the host here: a real one needs the two
    streams NOT to run concurrently, e.g. a profiler that serialises kernels.)"""
    from pwv_amd._lib import PwvPersistError
    engine = chain_knobs
    engine.CHAIN_FLOWS = True
    assert engine.sync_status() == 0
    ctypes.c_int.from_address(engine._sync_status_addr).value = 7
    with pytest.raises(PwvPersistError, match='in vain'):
        engine.raise_if_sync_failed()
    assert engine.CHAIN_FLOWS is False and engine.sync_status() == 0


def test_affine_sync_op_alone(gpu):
    """pwv_iaf_affine_sync_f32 without flags is the plain affine (the same fmaf as pwv_iaf_front_f32); with its own flag set
    by the call itself and the other flag already raised it must not wait."""
    import torch
    from pwv_amd import _lib, engine
    lib = _lib.lib()
    g = torch.Generator().manual_seed(5)
    z, s, b = (torch.randn((2, 1000, 1), generator=g).to(gpu) for _ in range(3))
    want = engine.iaf_affine_op(z, s, b)
    out = torch.empty_like(z)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.pwv_iaf_affine_sync_f32(z.data_ptr(), s.data_ptr(), b.data_ptr(), 1, out.data_ptr(), z.numel(), None, None, 0, st))
    assert torch.equal(out, want)
    flags = torch.zeros((64,), dtype=torch.int32, device=gpu)
    flags[32] = 1
    out2 = torch.empty_like(z)
    _lib.check(lib.pwv_iaf_affine_sync_f32(z.data_ptr(), s.data_ptr(), b.data_ptr(), 1, out2.data_ptr(), z.numel(), flags.data_ptr(),
                                           flags.data_ptr() + 128, 3, st))
    torch.cuda.synchronize()
    assert torch.equal(out2, want) and int(flags[0]) == 1 and engine.sync_status() == 0
