"""generate(case, ckpt, debug): mel-spectrogram -> waveform with the IAF-WaveNet student.

Counterpart of /root/reference/generate.py:16-78 with the same arguments.  Differences that
follow from the platform, not from the math:
  * no TF graph/session: the forward is one call of IAFVocoder on the GPU (the reference's single
    sess.run, generate.py:68);
  * the checkpoint is read by variable NAME (EMA shadows preferred when hp.train.use_ema,
    generate.py:55-66) from a TensorFlow V2 checkpoint (tf_checkpoint.py, no TF needed) or an
    .npz of TF-named arrays; with no checkpoint the model runs with random init exactly like
    the reference ("No checkpoint found", generate.py:65-66);
  * the result is written as .wav / .npy files into hp.logdir (the reference only writes
    TensorBoard audio summaries, generate.py:71-73);
  * `data_path: 'synthetic'` (bench cases) or a glob of .npy mel files replaces the wav dataset;
    wav input uses the torch STFT front-end in audio_frontend.py.
CLI (python-fire style, fire itself is not installed):  python -m pwv_amd.generate <case> [--ckpt=..] [--debug]
"""
from __future__ import absolute_import, division, print_function

import glob
import os
import sys

import numpy as np
import torch

from .hparam import hparam as hp
from .models import IAFVocoder
from .variables import reset_default_store


def _latest_checkpoint(logdir):
    """tf.train.latest_checkpoint: the TF V2 checkpoint named by <logdir>/checkpoint (or the newest
    *.index), else the newest .npz of TF-named arrays."""
    from .tf_checkpoint import latest_checkpoint
    tf_ck = latest_checkpoint(logdir)
    if tf_ck:
        return tf_ck
    cands = sorted(glob.glob(os.path.join(logdir, '*.npz')), key=os.path.getmtime)
    return cands[-1] if cands else None


def _load_mels(data_path, batch_size, length, device):
    """(gt_wav or None, melspec[N, t_mel, n_mels]) for the first `batch_size` items."""
    hop, n_mels = hp.signal.hop_length, hp.signal.n_mels
    t_mel = 1 + length // hop
    if data_path == 'synthetic':
        g = torch.Generator().manual_seed(0)
        return None, (torch.rand((batch_size, t_mel, n_mels), generator=g) * 2 - 1).to(device)
    files = sorted(glob.glob(data_path))
    if not files:
        raise FileNotFoundError('no input files match data_path %r (use data_path: synthetic for seeded noise mel)' % data_path)
    # like data_load.py:22-25: the last (1 - dataset_ratio) share of the files is the generation split
    split = int(len(files) * hp.train.dataset_ratio)
    files = (files[split:] or files)[:batch_size]
    print('dataset size is {}'.format(len(files)))
    if all(not f.endswith('.npy') for f in files):
        # wav input: host reads / trims / pads (data_load.py:42-50), then the spectrogram is computed ON THE DEVICE
        # (audio_frontend.wav_to_mel_device) -- the mel never visits the host between the front-end and the network
        from .audio_frontend import load_wav_fixed, wav_to_mel_device
        wavs = [load_wav_fixed(f, length) for f in files]
        while len(wavs) < batch_size:
            wavs.append(wavs[-1])
        gt = np.stack(wavs)
        return gt[..., None], wav_to_mel_device(torch.from_numpy(gt).to(device))
    mels, wavs = [], []
    for f in files:
        if f.endswith('.npy'):
            m = np.load(f).astype(np.float32)
        else:
            from .audio_frontend import wav_to_normalized_mel
            wav, m = wav_to_normalized_mel(f, length)
            wavs.append(wav)
        if m.shape[0] < t_mel:
            m = np.pad(m, [(0, t_mel - m.shape[0]), (0, 0)])
        mels.append(m[:t_mel])
    while len(mels) < batch_size:
        mels.append(mels[-1])
        if wavs:
            wavs.append(wavs[-1])
    gt = np.stack(wavs)[..., None] if wavs else None
    return gt, torch.from_numpy(np.stack(mels)).to(device)


def forward_over_ranks(mel, batch_size, length, device, make_model, noise_window, n_mels, hop, halo, group=None, allow_time_shards=True):
    """The forward of a job started with one process per GPU (WORLD_SIZE > 1; the reference is single-device,
    generate.py:47-49).  Rank 0 passes all mels [batch_size, t_mel, n_mels]; it gets all waveforms [batch_size, length, 1]
    back (other ranks: None).  Utterances shard across the ranks (distributed.generate_sharded); a batch smaller than the
    world shards in TIME instead, exactly (distributed.generate_time_sharded_ranks, halo = timeshard.chain_halo).  Every
    rank draws its share of ONE logistic-noise stream -- utterance i, sample t is counter i * length + t -- so the result
    does not depend on how the job was cut.  allow_time_shards=False (a model with a time-global normaliser, for which
    overlap-and-discard is not exact) keeps a small batch on utterance shards: ranks beyond the batch idle.
      make_model(n, window) -> callable(mel [n, 1 + window/hop, n_mels], z [n, window, 1]) -> [n, window, 1]
      noise_window(n, first_sample, window, first_item) -> z [n, window, 1]"""
    import torch.distributed as dist
    from .distributed import generate_sharded, generate_time_sharded_ranks, shard_bounds
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    models = {}

    def model_for(n, window):
        if (n, window) not in models:
            models[(n, window)] = make_model(n, window)
        return models[(n, window)]

    if batch_size >= world or not allow_time_shards:
        lo, _ = shard_bounds(batch_size, world, rank)
        return generate_sharded(lambda m, zz: model_for(m.shape[0], length)(m, noise_window(m.shape[0], 0, length, lo)),
                                mel, (1 + length // hop, n_mels), length, device, group=group)
    return generate_time_sharded_ranks(lambda m, zz, t0: model_for(m.shape[0], (m.shape[1] - 1) * hop)(m, noise_window(m.shape[0], t0, (m.shape[1] - 1) * hop, 0)),
                                       mel, n_mels, length, hop, halo, device, group=group)


def generate(case='default', ckpt=None, debug=False):
    '''
    :param case: experiment case name
    :param ckpt: checkpoint to load model
    :param debug: print per-stage timing (the reference hooks tfdbg here).
    '''
    hp.set_hparam_yaml(case)
    if not torch.cuda.is_available():
        raise RuntimeError('generate() needs an MI355X: the HIP path has no CPU fallback')
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))       # one process per GPU
    device = torch.device('cuda', torch.cuda.current_device())
    logdir = os.environ.get('PWV_LOGDIR', hp.logdir)

    store = reset_default_store(device=device)              # fresh "graph"
    batch_size, length = hp.generate.batch_size, hp.generate.length
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        # one process per GPU (torchrun): utterances -- or, for a batch smaller than the world, time slices -- shard over
        # the ranks; rank 0 reads the inputs and writes the outputs
        return _generate_over_ranks(store, batch_size, length, device, logdir, ckpt, debug)
    gt_wav, melspec = _load_mels(hp.data_path, batch_size, length, device)

    model = IAFVocoder(batch_size=batch_size, length=length, store=store)

    # load model
    ckpt = '{}/{}'.format(logdir, ckpt) if ckpt else (_latest_checkpoint(logdir) if os.path.isdir(logdir) else None)
    if ckpt:
        n = store.load_checkpoint(ckpt, use_ema=bool(hp.train.use_ema))
    else:
        print('No checkpoint found at {}.'.format(logdir))

    if debug:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    # feed forward (generate.py:68).  The call returns a verified result: a persistent launch that gave up is rerun on per-layer
    # launches, a forward that left the range of the split-fp16 arithmetic in exact fp32 -- on the same noise
    # (engine.verified_call); what comes back is what the reference's fp32 sess.run would have produced, or an exception.
    # verify=True is EXPLICIT: it outranks PWV_ASYNC=1 (whose default is enqueue-only) -- nothing unverified is written to disk
    pred = model(gt_wav, melspec, is_training=False, verify=True)
    if ckpt:
        # tf.train.Saver.restore fails on a variable the checkpoint lacks (generate.py:59-63); here variables are
        # created lazily by the forward, so the coverage check comes after it
        missing = store.not_restored()
        if missing:
            raise KeyError('checkpoint %s does not hold %d of the model\'s %d variables (they would be random-initialised): %s%s'
                           % (ckpt, len(missing), len(store.vars), ', '.join(missing[:6]), ' ...' if len(missing) > 6 else ''))
        no_shadow = store.ema_missing() if hp.train.use_ema else []
        if no_shadow:
            # with use_ema the reference restores EVERY trainable variable of 'iaf_vocoder' from its shadow (generate.py:59-63);
            # a checkpoint without one fails there, it does not fall back to the raw variable
            raise KeyError('checkpoint %s has no ExponentialMovingAverage shadow for %d model variable(s) although train.use_ema is '
                           'set: %s%s' % (ckpt, len(no_shadow), ', '.join(no_shadow[:6]), ' ...' if len(no_shadow) > 6 else ''))
        print('Successfully loaded checkpoint {} ({} variables)'.format(ckpt, n))
    if debug:
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print('forward: %.2f ms, %.3g samples/s (first call includes weight packing)' % (ms, batch_size * length / ms * 1e3))
    pred_wav = pred.cpu().numpy()
    _write_outputs(pred_wav, logdir)
    print('Done.')
    return pred_wav


def _write_outputs(pred_wav, logdir):
    try:
        os.makedirs(logdir, exist_ok=True)
        from scipy.io import wavfile
        for i in range(pred_wav.shape[0]):
            wavfile.write(os.path.join(logdir, 'pred_%d.wav' % i), hp.signal.sr, np.clip(pred_wav[i, :, 0], -1, 1))
        np.save(os.path.join(logdir, 'pred_wav.npy'), pred_wav)
        print('wrote %d waveform(s) to %s' % (pred_wav.shape[0], logdir))
    except OSError as e:
        print('could not write outputs to %s: %s' % (logdir, e))


def _generate_over_ranks(store, batch_size, length, device, logdir, ckpt, debug):
    """generate() under a launcher (WORLD_SIZE > 1): see forward_over_ranks."""
    import torch.distributed as dist
    from . import engine
    from .timeshard import chain_halo
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if not dist.is_initialized():
        dist.init_process_group(backend='nccl', device_id=device)      # 'nccl' IS RCCL on ROCm (xGMI)
    rank = dist.get_rank()
    hop, n_mels = hp.signal.hop_length, hp.signal.n_mels
    melspec = _load_mels(hp.data_path, batch_size, length, device)[1] if rank == 0 else None
    ckpt = '{}/{}'.format(logdir, ckpt) if ckpt else (_latest_checkpoint(logdir) if os.path.isdir(logdir) else None)
    if ckpt:
        store.load_checkpoint(ckpt, use_ema=bool(hp.train.use_ema))
    elif rank == 0:
        print('No checkpoint found at {}.'.format(logdir))
    seed = torch.tensor([int(os.environ.get('PWV_NOISE_SEED') or int.from_bytes(os.urandom(7), 'little'))], dtype=torch.int64, device=device)
    dist.broadcast(seed, src=0)                                         # one noise stream for the whole job
    seed = int(seed.item())
    halo = chain_halo(hp.model.dilations, hp.model.filter_width, hp.model.n_iaf, hop)

    errors = []

    def make_model(n, window):
        # A failure anywhere in this rank's share -- the constructor (length not a multiple of the hop), the forward, its
        # verification -- must not leave the other ranks alone in the gather: hand back zeros, keep the collectives paired,
        # and let the failure bit below stop every rank before anything is written
        try:
            m = IAFVocoder(batch_size=n, length=window, store=store)
        except Exception as e:
            errors.append(e)
            m = None

        def run(mel, z):
            # a verified call (explicitly: verify=True outranks PWV_ASYNC=1): this rank's share is complete and checked --
            # rerun on per-layer launches / in exact fp32 if need be -- BEFORE it is gathered
            try:
                if m is None:
                    raise errors[0]
                return m(None, mel, is_training=False, z=z, verify=True)
            except Exception as e:
                if not errors or errors[-1] is not e:
                    errors.append(e)
                return torch.zeros((mel.shape[0], window, 1), dtype=torch.float32, device=device)
        return run

    def noise_window(n, first_sample, window, first_item):
        try:
            return engine.logistic_noise_window(n, length, first_sample, window, device, seed, first_item)
        except Exception as e:
            errors.append(e)
            return torch.zeros((n, window, 1), dtype=torch.float32, device=device)

    # a normaliser that reduces over time (modules.py:274-284) makes a time slice depend on the whole utterance: such a model
    # shards by utterance only (ranks beyond the batch idle) -- timeshard.py is exact for causal FIR structure, nothing else
    time_ok = 'in' not in (hp.model.normalize, hp.model.normalize_cond, hp.model.normalize_wavenet)
    pred = forward_over_ranks(melspec, batch_size, length, device, make_model, noise_window, n_mels, hop, halo, allow_time_shards=time_ok)
    flag = torch.tensor([1 if errors else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)          # rank 0 writes only what EVERY rank has verified
    if int(flag.item()):
        if errors:
            raise errors[0]
        raise RuntimeError('generate(): the forward failed on another rank; nothing was written')
    if ckpt and store.not_restored():
        missing = store.not_restored()
        raise KeyError('checkpoint %s does not hold %d of the model\'s %d variables: %s' % (ckpt, len(missing), len(store.vars), ', '.join(missing[:6])))
    if ckpt and hp.train.use_ema and store.ema_missing():
        raise KeyError('checkpoint %s has no ExponentialMovingAverage shadow for: %s' % (ckpt, ', '.join(store.ema_missing()[:6])))
    if rank != 0:
        return None
    pred_wav = pred.cpu().numpy()
    _write_outputs(pred_wav, logdir)
    print('Done.')
    return pred_wav


def _fire(fn, argv):
    """Minimal python-fire work-alike: positionals, --name=value, --name value, --flag (a bare flag may be followed by
    another option: `generate c --debug --ckpt foo`)."""
    pos, kw = [], {}
    i = 0
    while i < len(argv):
        a = argv[i]
        i += 1
        if not a.startswith('--'):
            pos.append(a)
            continue
        k, eq, v = a[2:].partition('=')
        k = k.replace('-', '_')
        if not eq:
            if i < len(argv) and not argv[i].startswith('--'):
                v = argv[i]
                i += 1
            else:
                kw[k] = True           # bare flag; the next token (if any) is an option of its own
                continue
        kw[k] = {'True': True, 'False': False}.get(v, v)
    return fn(*pos, **kw)


if __name__ == '__main__':
    _fire(generate, sys.argv[1:])
