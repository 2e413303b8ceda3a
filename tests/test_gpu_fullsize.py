"""-m gpu: BASELINE.json configurations 2, 4 and 5 at FULL size, still against the fp64 oracle.

The oracle cannot run 160000..960000 samples of the 120-layer model in test time, but the network is strictly causal
with a finite receptive field (modules.py:168-172; 6142 samples for the default chain, rounded to 6160 = a multiple of
the hop), so two windows of a full-size run are functions of a few thousand input samples only and the oracle can
restate them exactly:

  * prefix: the first K samples depend on z[:K] and on mel frames (t+40)//80 <= K/80 -- the oracle run on the K-sample
    prefix must match hip[:, :K-40] (the last hop/2 samples of the short run see a clamped last frame);
  * late window: the samples t >= s + 6160 of a run on the window [s, L) (zeros for t < s, as at an utterance start)
    equal those of the full run -- the oracle run on the window must match the END of the full-size run.  This is the
    check that touches row indices > 160000 / the last tile32 block / the last mel frames (P stride at 12001 frames).

plus the size-independent properties of tests/test_gpu_golden.py (bitwise repeatability, causality in z and mel at a late
index, batch independence, finiteness).  Reference call sites: models.py:109-124 (C5), default.yaml:47-48 (C4 batch),
models.py:36-65 (flow chain).
"""
import numpy as np
import pytest

from oracle import iaf_oracle as O
from tests.util import TOL_F32, set_hparams

pytestmark = pytest.mark.gpu

HALO = 6160        # timeshard.chain_halo(default dilations, W=2, 4 flows, hop 80)
# tolerance of the reduced-precision 'f16' build extension against the fp64 oracle (tests/test_gpu_f16.py)
TOL_F16_MAX, TOL_F16_RMS = 5e-3, 1e-3


def _model(gpu, cfg, n, length, precision=None, seed=2):
    import torch
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    set_hparams(cfg)
    w = O.init_weights(cfg, seed=seed)
    store = VariableStore(device=gpu)
    store.load_dict(w)
    model = IAFVocoder(batch_size=n, length=length, store=store, precision=precision)
    mel, z = O.synthetic_inputs(n, length, cfg)
    return model, w, mel, z, torch.from_numpy(mel).to(gpu), torch.from_numpy(z).to(gpu)


def _check(got, want, precision):
    err = np.abs(got - want)
    if precision == 'f16':
        assert err.max() <= TOL_F16_MAX and np.sqrt((err ** 2).mean()) <= TOL_F16_RMS, (err.max(), np.sqrt((err ** 2).mean()))
    else:
        assert err.max() <= TOL_F32, err.max()


def _oracle_prefix(cfg, w, mel, z, y, K, precision, utts=None):
    """hip[:, :K-40] of the full-size run == fp64 oracle on the K-sample prefix."""
    hop = cfg.hop_length
    sel = slice(None) if utts is None else utts
    want = O.iaf_vocoder_forward(w, mel[sel, :K // hop + 1], z[sel, :K], cfg)
    _check(y[sel, :K - hop // 2], want[:, :K - hop // 2], precision)


def _oracle_late_window(cfg, w, mel, z, y, keep, precision, utt):
    """The last `keep` samples of utterance `utt` of the full-size run == fp64 oracle on the window
    [L - keep - HALO, L) (its first HALO outputs, which miss their history, are discarded)."""
    hop = cfg.hop_length
    L = z.shape[1]
    s = L - keep - HALO
    assert s > 0 and s % hop == 0
    want = O.iaf_vocoder_forward(w, mel[utt:utt + 1, s // hop:], z[utt:utt + 1, s:], cfg)
    _check(y[utt:utt + 1, L - keep:], want[:, HALO:], precision)


def _properties(model, mel_t, z_t, y0, t0, frame):
    """Bitwise repeatability and strict causality w.r.t. z[t0] and mel[frame] on the last utterance."""
    import torch
    y1 = model(None, mel_t, is_training=False, z=z_t)
    assert torch.equal(y0, y1)
    assert torch.isfinite(y0).all() and float(y0.abs().max()) < 50.0
    n = z_t.shape[0] - 1
    z2 = z_t.clone()
    z2[n, t0, 0] += 1.0
    y2 = model(None, mel_t, is_training=False, z=z2)
    assert torch.equal(y0[:n], y2[:n])                                   # other utterances untouched
    assert torch.equal(y0[n, :t0], y2[n, :t0]) and not torch.equal(y0[n, t0:], y2[n, t0:])
    mel2 = mel_t.clone()
    mel2[n, frame, :] += 0.5                                             # frame f -> samples f*80 - 40 onwards
    y3 = model(None, mel2, is_training=False, z=z_t)
    assert torch.equal(y0[:n], y3[:n])
    assert torch.equal(y0[n, :frame * 80 - 40], y3[n, :frame * 80 - 40]) and not torch.equal(y0[n], y3[n])


@pytest.mark.parametrize('precision', ['f16x3', 'f16'])
def test_c5_transposed_conv_60s_full_size(gpu, precision):
    """BASELINE config 5: transposed-conv upsampling (models.py:109-124), 1 x 960000 samples (12001 frames), the default
    split-fp16 arithmetic and the fp16 storage mode the config names."""
    import torch
    cfg = O.ModelConfig(cond_upsample_method='transposed_conv')
    L = 960000
    model, w, mel, z, mel_t, z_t = _model(gpu, cfg, 1, L, precision)
    y_t = model(None, mel_t, is_training=False, z=z_t)
    y = y_t.cpu().numpy()
    assert y.shape == (1, L, 1)
    _oracle_prefix(cfg, w, mel, z, y, 4000, precision)
    _oracle_late_window(cfg, w, mel, z, y, 2400, precision, 0)
    _properties(model, mel_t, z_t, y_t, 900001, 11900)
    del model, y_t
    torch.cuda.empty_cache()


@pytest.mark.parametrize('method', ['transposed_conv', 'repeat'])
def test_c5_60s_in_eight_time_shards_is_bit_identical(gpu, method):
    """SURVEY 8 f-3 at BASELINE config 5's size: the 60 s utterance (1 x 960000) cut into 8 time shards with the flow chain's
    look-back (6160 samples) recomputed per shard and discarded is torch.equal to the unsharded forward -- what 8 GPUs
    compute under `bench.py --case bench/c5 --gpus 8 --shard time` / generate() with WORLD_SIZE = 8, here tiled on one
    GPU (pwv_amd/timeshard.py; 'repeat' runs the persistent stack launches, 'transposed_conv' the per-layer ones)."""
    import torch
    from pwv_amd.timeshard import chain_halo, generate_time_sharded, shard_plan, vocoder_forward_factory
    cfg = O.ModelConfig(cond_upsample_method=method)
    L = 960000
    model, w, mel, z, mel_t, z_t = _model(gpu, cfg, 1, L)
    want = model(None, mel_t, is_training=False, z=z_t).clone()
    halo = chain_halo(cfg.dilations, cfg.filter_width, cfg.n_iaf, cfg.hop_length)
    assert halo == HALO
    got = generate_time_sharded(vocoder_forward_factory(model.store), mel_t, z_t, cfg.hop_length, halo, n_shards=8)
    assert torch.equal(got, want)
    plan = shard_plan(L, 8, halo, cfg.hop_length)
    assert [b - a for _, a, b in plan] == [120000] * 8 and [a - c0 for c0, a, _ in plan] == [0] + [halo] * 7
    # the pieces ranks 2 and 5 of an 8-rank job would return
    for r in (2, 5):
        piece = generate_time_sharded(vocoder_forward_factory(model.store), mel_t, z_t, cfg.hop_length, halo, n_shards=8, shard_ids=[r])
        assert torch.equal(piece, want[:, plan[r][1]:plan[r][2]])
    del model, want, got
    torch.cuda.empty_cache()


def test_c4_share_8_utterances_full_size(gpu):
    """BASELINE config 4, one GPU's share: 8 utterances x 64000 samples (default.yaml:47-48 batch semantics, every
    utterance exactly generate.length: data_load.py:45-50).  EVERY utterance's prefix against the oracle, the end of
    the LAST utterance (rows 448000..512000) against the oracle, batch independence bit for bit."""
    import torch
    cfg = O.ModelConfig()
    n, L = 8, 64000
    model, w, mel, z, mel_t, z_t = _model(gpu, cfg, n, L)
    y_t = model(None, mel_t, is_training=False, z=z_t)
    y = y_t.cpu().numpy()
    _oracle_prefix(cfg, w, mel, z, y, 4000, 'f16x3')
    _oracle_late_window(cfg, w, mel, z, y, 2400, 'f16x3', n - 1)
    _properties(model, mel_t, z_t, y_t, 60001, 700)
    # a batched run equals per-utterance runs bit for bit (x[t-d] = 0 at every utterance start)
    from pwv_amd.models import IAFVocoder
    one = IAFVocoder(batch_size=1, length=L, store=model.store)
    for i in (0, 3, 7):
        yi = one(None, mel_t[i:i + 1].contiguous(), is_training=False, z=z_t[i:i + 1].contiguous())
        assert torch.equal(yi[0], y_t[i])


def test_c2_shared_nets_full_size(gpu):
    """BASELINE config 2 (build extension: one 2-output net per flow), 4 flows with the default dilations, 160000 samples."""
    cfg = O.ModelConfig(shared_nets=True)
    L = 160000
    model, w, mel, z, mel_t, z_t = _model(gpu, cfg, 1, L)
    y_t = model(None, mel_t, is_training=False, z=z_t)
    y = y_t.cpu().numpy()
    _oracle_prefix(cfg, w, mel, z, y, 8000, 'f16x3')
    _oracle_late_window(cfg, w, mel, z, y, 2400, 'f16x3', 0)
    _properties(model, mel_t, z_t, y_t, 150001, 1900)


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
def test_c3_default_model_full_size_vs_oracle(gpu, precision):
    """BASELINE config 3 (the headline workload, 1 x 160000): prefix and late window against the oracle in both
    fp32-parity arithmetics (the property set lives in tests/test_gpu_golden.py::test_full_size_properties_c3)."""
    cfg = O.ModelConfig()
    L = 160000
    model, w, mel, z, mel_t, z_t = _model(gpu, cfg, 1, L, precision)
    y = model(None, mel_t, is_training=False, z=z_t).cpu().numpy()
    _oracle_prefix(cfg, w, mel, z, y, 8000, precision)
    _oracle_late_window(cfg, w, mel, z, y, 2400, precision, 0)


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
def test_ragged_many_workgroups(gpu, precision):
    """A length whose last 32-row unit is half empty and whose units do not divide over the workgroups (3 x 100080 samples:
    9383 units, the last of every utterance 16 rows short of a tile32 block boundary shared with the next utterance): the
    write-through stores address a workgroup's own units through a bounded descriptor, so the ends of the ranges are where
    a mistake would show.  Prefix and end of the LAST utterance against the oracle, batched == single bit for bit."""
    import torch
    cfg = O.ModelConfig()
    n, L = 3, 100080
    model, w, mel, z, mel_t, z_t = _model(gpu, cfg, n, L, precision)
    y_t = model(None, mel_t, is_training=False, z=z_t)
    y = y_t.cpu().numpy()
    _oracle_prefix(cfg, w, mel, z, y, 4000, precision, utts=slice(1, 2))
    _oracle_late_window(cfg, w, mel, z, y, 2400, precision, n - 1)
    from pwv_amd.models import IAFVocoder
    one = IAFVocoder(batch_size=1, length=L, store=model.store, precision=precision)
    yi = one(None, mel_t[1:2].contiguous(), is_training=False, z=z_t[1:2].contiguous())
    assert torch.equal(yi[0], y_t[1])
