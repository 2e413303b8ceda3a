"""CPU: the wav -> mel front-end against analytic properties of the librosa
recipe it restates (librosa itself is not installable here)."""
import numpy as np
import pytest

from pwv_amd import audio_frontend as A


def test_stft_frames_and_sine_peak():
    sr, n_fft, win, hop = 16000, 512, 400, 80
    t = np.arange(16000) / sr
    wav = 0.5 * np.sin(2 * np.pi * 1000.0 * t).astype(np.float32)
    mag = A.stft_mag(wav, n_fft, win, hop)
    assert mag.shape == (257, 1 + 16000 // hop)                 # == t_mel of models.py:20
    assert np.all(mag[:, 20:180].argmax(axis=0) == 32)         # 1000 Hz / (16000/512) = bin 32
    # a hann window of length 400 has coherent gain 200: amplitude 0.5 -> peak 0.5 * 200 / 2 = 50
    assert abs(mag[32, 100] - 50.0) < 0.5


def test_mel_filterbank_slaney():
    fb = A.mel_filterbank(16000, 512, 80)
    assert fb.shape == (80, 257) and np.all(fb >= 0)
    assert np.all(fb.sum(axis=1) > 0)                           # no empty band at 80 mels / 512 fft
    hz = np.linspace(0, 8000, 257)
    centers = (fb * hz[None, :]).sum(axis=1) / fb.sum(axis=1)
    assert np.all(np.diff(centers) > 0) and centers[0] < 100 and 7000 < centers[-1] < 8000
    # Slaney area normalisation: integral of each triangle over frequency == 1 (up to grid sampling)
    area = fb.sum(axis=1) * (8000 / 256)
    assert np.allclose(area, 1.0, atol=0.15)
    # mel scale is linear below 1 kHz and log above (Slaney)
    assert np.allclose(A._mel_to_hz(A._hz_to_mel([100.0, 999.0, 1000.0, 4000.0])), [100.0, 999.0, 1000.0, 4000.0])
    assert abs(float(A._hz_to_mel(1000.0)) - 15.0) < 1e-9


def test_db_and_normalisation_range():
    s = np.array([[1e-9, 1e-3, 1.0, 100.0]])
    db = A.amplitude_to_db(s)
    assert np.allclose(db, [[-40.0, -40.0, 0.0, 40.0]])         # amin 1e-5 -> -100 dB / -60 dB, then clipped to max (40) - top_db (80)
    n = A.normalize_db(np.array([-100.0, -55.0, -10.0, 35.0, 90.0]), 35, -55)
    assert np.allclose(n, [-1.0, -1.0, 0.0, 1.0, 1.0])


def test_trim_and_fix_length():
    rng = np.random.RandomState(0)
    wav = np.concatenate([np.zeros(4000), 0.3 * rng.randn(8000), np.zeros(6000)]).astype(np.float32)
    tr = A.trim_wav(wav)
    # frames are 2048 long on a 512 grid: a frame is non-silent as soon as its window touches the signal
    assert 8000 <= len(tr) <= 8000 + 2048 + 2 * 512
    assert np.abs(tr).max() > 0.5
    assert len(A.trim_wav(np.zeros(3000, np.float32))) in (0, 3000)
    assert np.array_equal(A.fix_length(np.arange(5.0), 8), [0, 1, 2, 3, 4, 0, 0, 0])
    assert np.array_equal(A.fix_length(np.arange(5.0), 3), [0, 1, 2])


def test_wav_file_to_mel(tmp_path):
    from scipy.io import wavfile
    from pwv_amd.hparam import hparam as hp
    hp.set_hparam_yaml('default')
    sr = 16000
    t = np.arange(2 * sr) / sr
    wav = (0.4 * np.sin(2 * np.pi * 440 * t) * (t > 0.25)).astype(np.float32)
    path = str(tmp_path / 'a.wav')
    wavfile.write(path, sr, (wav * 32767).astype(np.int16))
    w, mel = A.wav_to_normalized_mel(path, 16000)
    assert w.shape == (16000,) and mel.shape == (201, 80) and mel.dtype == np.float32
    assert mel.min() >= -1.0 and mel.max() <= 1.0 and mel.max() > 0.0
    assert abs(np.abs(w).max() - 0.4) < 0.01 and np.abs(w[:1600]).max() > 0.01     # 0.25 s of leading silence trimmed (to the frame grid)
    # resampling path: an 8 kHz file comes back at hp.signal.sr
    wavfile.write(path, 8000, (wav[::2] * 32767).astype(np.int16))
    assert abs(len(A.read_wav(path, 16000)) - 2 * sr) <= 2


def test_stft_matches_scipy_signal_stft():
    """The restated librosa.stft (centred, reflect padding, hann(400) zero-padded to 512, hop 80; audio.py:133-134) against
    scipy.signal.stft configured the same way (boundary='even' is numpy's 'reflect'; scipy divides by the window sum)."""
    from scipy.signal import stft
    rng = np.random.RandomState(3)
    sr, n_fft, win, hop = 16000, 512, 400, 80
    wav = (0.3 * rng.randn(8000) + 0.2 * np.sin(2 * np.pi * 700 * np.arange(8000) / sr)).astype(np.float32)
    window = A.analysis_window(n_fft, win)
    _, _, z = stft(wav.astype(np.float64), fs=sr, window=window, nperseg=n_fft, noverlap=n_fft - hop, nfft=n_fft, boundary='even',
                   padded=False, return_onesided=True)
    want = np.abs(z) * window.sum()
    got = A.stft_mag(wav, n_fft, win, hop)
    assert got.shape == (257, 101)
    assert want.shape[1] >= got.shape[1] and np.abs(got - want[:, :got.shape[1]]).max() <= 1e-9 * max(1.0, want.max())


def test_slaney_filterbank_against_hand_derived_constants():
    """tests/golden/slaney_mel_16k_512_80.json holds the 82 band edges and sample filter weights derived with plain Python
    floats from the Slaney formulas (linear below 1 kHz, log above, unit-area triangles): librosa.filters.mel's documented
    construction (audio.py:241)."""
    import json
    import os
    fix = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'slaney_mel_16k_512_80.json')))
    assert abs(float(A._hz_to_mel(8000.0)) - fix['mel_max']) < 1e-9 and abs(fix['mel_max'] - (15 + 27 * np.log(8) / np.log(6.4))) < 1e-9
    pts = A._mel_to_hz(np.linspace(0.0, fix['mel_max'], 82))
    assert np.abs(pts - np.array(fix['mel_points_hz'])).max() < 1e-7
    fb = A.mel_filterbank(fix['sr'], fix['n_fft'], fix['n_mels'])
    n = 0
    for m, pairs in fix['weights'].items():
        for b, w in pairs:
            assert abs(fb[int(m), int(b)] - w) < 1e-12
            n += 1
    assert n >= 25


def test_amplitude_to_db_reference_semantics():
    """librosa.amplitude_to_db(S) = power_to_db(S**2, amin=amin**2) with top_db relative to the array maximum."""
    s = np.array([[3.0, 1e-7], [0.02, 0.5]])
    db = A.amplitude_to_db(s)
    want = np.array([[20 * np.log10(3.0), 0.0], [20 * np.log10(0.02), 20 * np.log10(0.5)]])
    want[0, 1] = max(-100.0, want.max() - 80.0)
    assert np.allclose(db, want, atol=1e-12)
