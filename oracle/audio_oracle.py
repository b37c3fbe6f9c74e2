"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the reference's mel -> waveform inversion
(src/audio_tacotron.py:99-116 inv_mel_spectrogram with use_lws=False: denormalise, dB -> amplitude, pseudo-inverse mel
basis, power 1.5, Griffin-Lim 60 iterations, inverse pre-emphasis; hparams from src/hparams_tacotron.py:77-117).

The reference delegates to librosa (filters.mel, stft, istft) and scipy.signal.lfilter.  librosa is absent here and on the
GPU box, so its published algorithms are restated (librosa 0.6.x, the version contemporary with the reference):
  filters.mel  -- Slaney mel scale (htk=False), triangular filters, area ("slaney") normalisation;
  stft         -- centred frames with reflect padding, periodic Hann window, n_fft = win_length;
  istft        -- windowed overlap-add divided by the window's sum of squares where it exceeds tiny, trimmed by n_fft/2.
parity unpinned: none of the reference's files holds a waveform or spectrogram this could be checked against.
"""
from __future__ import annotations

import numpy as np

MIN_LEVEL_DB, REF_LEVEL_DB, MAX_ABS_VALUE = -100.0, 20.0, 1.0     # hparams_tacotron.py:99,110,111
POWER, GRIFFIN_LIM_ITERS, PREEMPHASIS = 1.5, 60, 0.97            # :116,117,107
FMIN, FMAX = 125.0, 7600.0                                        # :112,113


def hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sample_rate, n_fft, n_mels, fmin=FMIN, fmax=FMAX):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax): (n_mels, 1 + n_fft/2) float32."""
    fftfreqs = np.linspace(0, sample_rate / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def denormalize(D):
    """audio_tacotron.py:242-248, allow_clipping, asymmetric."""
    return np.clip(D, 0, MAX_ABS_VALUE) * -MIN_LEVEL_DB / MAX_ABS_VALUE + MIN_LEVEL_DB


def db_to_amp(x):
    return np.power(10.0, x * 0.05)


def mel_to_linear(mel, inv_basis):
    """audio_tacotron.py:202-206."""
    return np.maximum(1e-10, inv_basis @ mel)


def hann(n):
    return (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)).astype(np.float64)     # periodic (fftbins=True)


def stft(y, n_fft, hop):
    yp = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop
    w = hann(n_fft)
    frames = np.stack([yp[t * hop:t * hop + n_fft] * w for t in range(n_frames)], axis=1)
    return np.fft.rfft(frames, axis=0)            # (1 + n_fft/2, n_frames)


def istft(S, hop):
    n_fft = 2 * (S.shape[0] - 1)
    n_frames = S.shape[1]
    w = hann(n_fft)
    L = n_fft + hop * (n_frames - 1)
    y = np.zeros(L)
    wss = np.zeros(L)
    frames = np.fft.irfft(S, n=n_fft, axis=0)
    for t in range(n_frames):
        y[t * hop:t * hop + n_fft] += w * frames[:, t]
        wss[t * hop:t * hop + n_fft] += w * w
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2:L - n_fft // 2]


def griffin_lim(S, n_fft, hop, iters=GRIFFIN_LIM_ITERS, angles0=None):
    """audio_tacotron.py:142-153.  S (1 + n_fft/2, T) magnitudes; angles0: the uniform [0,1) numbers behind the random
    initial phases (the reference draws them with np.random.rand)."""
    if angles0 is None:
        angles0 = np.random.rand(*S.shape)
    angles = np.exp(2j * np.pi * angles0)
    Sc = np.abs(S).astype(np.complex128)
    y = istft(Sc * angles, hop)
    for _ in range(iters):
        X = stft(y, n_fft, hop)
        angles = np.exp(1j * np.angle(X))
        y = istft(Sc * angles, hop)
    return y


def inv_preemphasis(x, k=PREEMPHASIS):
    """scipy.signal.lfilter([1], [1, -k], x): y[n] = x[n] + k y[n-1]."""
    y = np.empty_like(x, dtype=np.float64)
    acc = 0.0
    for n in range(len(x)):
        acc = x[n] + k * acc
        y[n] = acc
    return y


def linear_from_mel(mel, sample_rate, fft_size, n_mels):
    """The magnitudes Griffin-Lim starts from: (1 + fft_size/2, T) = mel_to_linear(db_to_amp(denorm + ref))^power."""
    inv = np.linalg.pinv(mel_basis(sample_rate, fft_size, n_mels).astype(np.float64))
    S = mel_to_linear(db_to_amp(denormalize(mel.astype(np.float64)) + REF_LEVEL_DB), inv)
    return S ** POWER


def inv_mel_spectrogram(mel, sample_rate, fft_size, hop_size, n_mels, iters=GRIFFIN_LIM_ITERS, angles0=None):
    """audio_tacotron.py:99-116 (use_lws=False).  mel (n_mels, T) normalised to [0, 1]."""
    S = linear_from_mel(mel, sample_rate, fft_size, n_mels)
    return inv_preemphasis(griffin_lim(S, fft_size, hop_size, iters, angles0))
