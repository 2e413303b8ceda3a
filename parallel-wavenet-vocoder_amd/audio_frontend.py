"""wav -> normalised dB mel-spectrogram, the input side of the generation path.

Restates, in numpy, the exact recipe `Dataset._get_wav_and_melspec` applies at generation time
(/root/reference/data_load.py:37-56 with audio.py:14-37,102-141,232-243,278-286,327-356), whose
arithmetic lives in librosa 0.5.1 (requirements.txt; not installable here):

    read_wav(sr) -> trim_wav (librosa.effects.trim: top_db 60, frame 2048, hop 512, ref max)
    -> first `length` samples -> fix_length (zero pad) -> STFT (n_fft 512, win 400 periodic hann
    zero-padded to n_fft, hop 80, center, reflect) -> |.| -> Slaney mel filterbank (80 bands,
    0..sr/2, area-normalised) -> amplitude_to_db (amin 1e-5, top_db 80) ->
    (clip((db - min_db)/(max_db - min_db), 0, 1) - 0.5) * 2

so the mel has exactly 1 + length/hop frames and values in [-1, 1], which is what the network
was trained on.  File reading, trimming and padding are host-side data preparation; the spectrogram itself
(STFT -> mel -> dB -> normalisation) also exists as a HIP kernel (`wav_to_mel_device`, csrc/pwv_audio.hip) so that
generate() on wav input keeps the mel on the device.  ROLE OF THE NUMPY FUNCTIONS: `wav2melspec_db` and its helpers are a
CPU restatement of the reference recipe that serves (a) as the host path for .npy / CPU-side inputs and (b) as the CHECKER
of the HIP kernel in tests/test_gpu_unfused_and_e2e.py::test_device_mel_frontend_matches_numpy_restatement -- they are test
reference living in the product package, not an independent oracle; `trim_wav` (librosa.effects.trim) and the resampling in
`read_wav` have property tests only (tests/test_audio_frontend.py).  Parity with librosa is by construction from its documented algorithms (no librosa here to diff against; the
STFT is checked against scipy.signal.stft and the filterbank against hand-derived constants in tests/test_audio_frontend.py);
resampling uses scipy's polyphase filter where librosa used resampy (only matters when the file's
rate differs from hp.signal.sr).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .hparam import hparam as hp


def read_wav(path: str, sr: int) -> np.ndarray:
    """audio.py:14-16 (librosa.load, mono, float32 in [-1, 1], resampled to sr)."""
    from scipy.io import wavfile
    rate, data = wavfile.read(path)
    if data.dtype.kind == 'i':
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == 'u':
        data = (data.astype(np.float32) - 128.0) / 128.0
    else:
        data = data.astype(np.float32)
    if data.ndim > 1:
        data = data.mean(axis=1)
    if rate != sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(rate), int(sr))
        data = resample_poly(data, sr // g, rate // g).astype(np.float32)
    return data


def _frame_rmse_db(y: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    ypad = np.pad(y, frame_length // 2, mode='reflect')
    n = 1 + (len(ypad) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n)[:, None]
    mse = np.mean(np.abs(ypad[idx]) ** 2, axis=1)
    ref = max(1e-10, mse.max())
    return 10.0 * np.log10(np.maximum(1e-10, mse)) - 10.0 * np.log10(ref)


def trim_wav(wav: np.ndarray, top_db: float = 60.0, frame_length: int = 2048, hop_length: int = 512) -> np.ndarray:
    """audio.py:29-31 (librosa.effects.trim defaults): drop leading / trailing frames quieter than
    top_db below the loudest frame."""
    if len(wav) == 0:
        return wav
    nonsilent = np.flatnonzero(_frame_rmse_db(wav, frame_length, hop_length) > -top_db)
    if nonsilent.size == 0:
        return wav[:0]
    start = int(nonsilent[0]) * hop_length
    end = min(len(wav), (int(nonsilent[-1]) + 1) * hop_length)
    return wav[start:end]


def fix_length(wav: np.ndarray, length: int) -> np.ndarray:
    """audio.py:34-37 (librosa.util.fix_length: truncate or zero-pad at the end)."""
    if len(wav) >= length:
        return wav[:length]
    return np.pad(wav, (0, length - len(wav)))


def stft_mag(wav: np.ndarray, n_fft: int, win_length: int, hop_length: int) -> np.ndarray:
    """|librosa.stft| (audio.py:133-134): centred, reflect-padded, periodic hann of win_length
    zero-padded to n_fft.  Returns [1 + n_fft/2, 1 + len/hop]."""
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)     # scipy get_window('hann', fftbins=True)
    lpad = (n_fft - win_length) // 2
    window = np.pad(win, (lpad, n_fft - win_length - lpad))
    ypad = np.pad(wav.astype(np.float64), n_fft // 2, mode='reflect')
    n_frames = 1 + (len(ypad) - n_fft) // hop_length
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(n_frames)[:, None]
    return np.abs(np.fft.rfft(ypad[idx] * window[None, :], axis=1)).T


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    mels = f / (200.0 / 3)
    log_t = f >= 1000.0
    return np.where(log_t, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0), mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3) * m)


def mel_filterbank(sr: int, n_fft: int, n_mels: int) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels) (audio.py:241): Slaney scale, fmin 0, fmax sr/2,
    triangles normalised to unit area.  [n_mels, 1 + n_fft/2]."""
    fft_f = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return weights * enorm[:, None]


def amplitude_to_db(s: np.ndarray, amin: float = 1e-5, top_db: float = 80.0) -> np.ndarray:
    """librosa.amplitude_to_db (audio.py:347), ref 1.0."""
    db = 10.0 * np.log10(np.maximum(amin ** 2, np.abs(s) ** 2))
    return np.maximum(db, db.max() - top_db)


def normalize_db(db: np.ndarray, max_db: float, min_db: float) -> np.ndarray:
    """audio.py:254-286: [-1, 1]."""
    return (np.clip((db - min_db) / (max_db - min_db), 0, 1) - 0.5) * 2


def wav2melspec_db(wav, sr, n_fft, win_length, hop_length, n_mels, max_db=None, min_db=None) -> np.ndarray:
    """audio.py:341-356 -> [t, n_mels]."""
    mel = mel_filterbank(sr, n_fft, n_mels) @ stft_mag(wav, n_fft, win_length, hop_length)
    db = amplitude_to_db(mel)
    if max_db and min_db:
        db = normalize_db(db, max_db, min_db)
    return db.T.astype(np.float32)


def analysis_window(n_fft: int, win_length: int) -> np.ndarray:
    """Periodic hann of win_length, centred in n_fft (what librosa.stft builds from win_length < n_fft)."""
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)
    lpad = (n_fft - win_length) // 2
    return np.pad(win, (lpad, n_fft - win_length - lpad))


_device_consts = {}


def wav_to_mel_device(wav, normalise: Optional[bool] = None):
    """wav [N, L] float32 on the GPU -> (normalised) dB mel [N, 1 + L/hop, n_mels] on the GPU (pwv_wav_to_mel_db_f32),
    with the current hparams' signal settings.  Like audio.wav2melspec_db (audio.py:350) the dB range is normalised only
    when BOTH max_db and min_db are set; `normalise` overrides."""
    import torch
    from . import _lib, engine
    s = hp.signal
    wav = engine._require_cuda_f32(wav, 'wav')
    if wav.dim() != 2:
        raise ValueError('wav must be [N, L], got %s' % (tuple(wav.shape),))
    n, length = wav.shape
    key = (wav.device, s.sr, s.n_fft, s.win_length, s.n_mels)
    if key not in _device_consts:
        _device_consts[key] = (torch.from_numpy(analysis_window(s.n_fft, s.win_length).astype(np.float32)).to(wav.device),
                               torch.from_numpy(mel_filterbank(s.sr, s.n_fft, s.n_mels).astype(np.float32)).to(wav.device))
    window, basis = _device_consts[key]
    mel = torch.empty((n, 1 + length // s.hop_length, s.n_mels), dtype=torch.float32, device=wav.device)
    if normalise is None:
        normalise = bool(s.get('max_db', None) and s.get('min_db', None))
    max_db, min_db = (float(s.max_db), float(s.min_db)) if normalise else (1.0, 0.0)      # (unused by the kernel when not normalising)
    _lib.check(_lib.lib().pwv_wav_to_mel_db_f32(wav.data_ptr(), window.data_ptr(), basis.data_ptr(), mel.data_ptr(), n, length, s.n_fft,
                                                s.hop_length, s.n_mels, 1e-5, 80.0, max_db, min_db, int(normalise),
                                                engine._stream()), 'pwv_wav_to_mel_db_f32')
    return mel


def load_wav_fixed(path: str, length: int) -> np.ndarray:
    """data_load.py:42-50 at generation time: read, trim, first chunk, zero-pad to exactly `length` samples."""
    wav = trim_wav(read_wav(path, hp.signal.sr))
    return fix_length(wav[:length], length).astype(np.float32)


def wav_to_normalized_mel(path: str, length: int):
    """data_load.py:37-56 at generation time (first chunk): returns (wav [length], mel [1 + length/hop, n_mels])."""
    s = hp.signal
    wav = trim_wav(read_wav(path, s.sr))
    wav = fix_length(wav[:length], length)
    mel = wav2melspec_db(wav, s.sr, s.n_fft, s.win_length, s.hop_length, s.n_mels, max_db=s.max_db, min_db=s.min_db)
    return wav.astype(np.float32), mel
