"""Mel -> waveform inversion, the audio export of the reference's epoch loop (src/main.py:164-197 calling
src/audio_tacotron.py:99-116): denormalise, dB -> amplitude, pseudo-inverse mel basis, power 1.5, Griffin-Lim (60
iterations), inverse pre-emphasis.  SURVEY.md section 8f row 4.

    inv_mel_spectrogram(mel_spectrogram, sample_rate, fft_size, hop_size, n_mel) -> waveform        (same signature)

The arithmetic runs in libnsg.so (csrc/audio.hip: in-LDS FFT, fused STFT + phase update, overlap-add, recurrence); the mel
filterbank and its pseudo-inverse are constants built once on the host with numpy (librosa.filters.mel restated: Slaney
scale, area normalisation).  hparams are the reference's (src/hparams_tacotron.py:77-117: use_lws=False, power 1.5,
60 iterations, pre-emphasis 0.97, min_level_db -100, ref_level_db 20, fmin 125, fmax 7600, clipped [0, 1] normalisation).

parity unpinned: librosa is absent (here, on the GPU box, and from the reference's own tree), and no file of the reference
holds a waveform; tests compare with the numpy restatement in oracle/audio_oracle.py (same initial phases) and check the
transform identities (istft(stft(y)) == y, the spectral error falls over the iterations).
"""
from __future__ import annotations

import functools
from ctypes import c_float, c_int32, c_size_t

import numpy as np
import torch

from . import _lib
from .ops import WS, _chk, _p, _stream

MIN_LEVEL_DB, REF_LEVEL_DB, MAX_ABS_VALUE = -100.0, 20.0, 1.0     # hparams_tacotron.py:99,110,111
POWER, GRIFFIN_LIM_ITERS, PREEMPHASIS = 1.5, 60, 0.97            # :116,117,107
FMIN, FMAX = 125.0, 7600.0                                        # :112,113


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


@functools.lru_cache(maxsize=8)
def mel_basis(sample_rate: int, fft_size: int, n_mels: int, fmin: float = FMIN, fmax: float = FMAX) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) (audio_tacotron.py:208-219): (n_mels, 1 + fft_size/2) float32."""
    fftfreqs = np.linspace(0, sample_rate / 2.0, 1 + fft_size // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, 1 + fft_size // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


@functools.lru_cache(maxsize=8)
def _inv_mel_basis(sample_rate, fft_size, n_mels):
    return np.ascontiguousarray(np.linalg.pinv(mel_basis(sample_rate, fft_size, n_mels).astype(np.float64)).astype(np.float32))


def mel_to_linear(mel: torch.Tensor, sample_rate=22050, fft_size=1024, n_mels=80, power=POWER) -> torch.Tensor:
    """mel (B, n_mels, T) normalised to [0, 1] on the GPU -> Griffin-Lim's magnitudes S (B, T, 1 + fft_size/2)."""
    _chk(mel, "mel")
    B, M, T = mel.shape
    if M != n_mels:
        raise _lib.NsgError(f"mel_to_linear: expected {n_mels} mel bins, got {M}")
    F = fft_size // 2 + 1
    inv = torch.from_numpy(_inv_mel_basis(sample_rate, fft_size, n_mels)).to(mel.device)
    S = torch.empty(B, T, F, dtype=torch.float32, device=mel.device)
    _lib.call("nsg_audio_mel_to_linear", _p(mel), _p(inv), _p(S), c_int32(B), c_int32(M), c_int32(T), c_int32(F), c_float(MIN_LEVEL_DB),
              c_float(REF_LEVEL_DB), c_float(MAX_ABS_VALUE), c_float(power), _stream())
    return S


def griffin_lim(S: torch.Tensor, fft_size=1024, hop_size=256, iters=GRIFFIN_LIM_ITERS, angles0: torch.Tensor | None = None) -> torch.Tensor:
    """S (B, T, F) magnitudes -> y (B, hop*(T-1)).  angles0: uniform [0,1) numbers for the initial phases (drawn if None)."""
    _chk(S, "S")
    B, T, F = S.shape
    if F != fft_size // 2 + 1:
        raise _lib.NsgError(f"griffin_lim: S has {F} bins, fft_size {fft_size} needs {fft_size // 2 + 1}")
    u = torch.rand(B, T, F, device=S.device) if angles0 is None else _chk(angles0.contiguous(), "angles0")
    y = torch.empty(B, hop_size * (T - 1), dtype=torch.float32, device=S.device)
    nb = _lib.query("nsg_audio_griffin_lim_workspace_bytes", c_int32(B), c_int32(T), c_int32(fft_size))
    ws = WS.get(nb, S.device)
    _lib.call("nsg_audio_griffin_lim", _p(S), _p(u), _p(y), c_int32(B), c_int32(T), c_int32(fft_size), c_int32(hop_size), c_int32(iters),
              _p(ws), c_size_t(nb), _stream())
    return y


def stft(y: torch.Tensor, fft_size=1024, hop_size=256) -> torch.Tensor:
    """y (B, L) -> complex64 (B, 1 + L // hop, F): librosa.stft per row (frame-major)."""
    _chk(y, "y")
    B, L = y.shape
    X = torch.empty(B, 1 + L // hop_size, fft_size // 2 + 1, 2, dtype=torch.float32, device=y.device)
    _lib.call("nsg_audio_stft", _p(y), _p(X), c_int32(B), c_int32(L), c_int32(fft_size), c_int32(hop_size), _stream())
    return torch.view_as_complex(X)


def inv_preemphasis(y: torch.Tensor, k=PREEMPHASIS) -> torch.Tensor:
    _chk(y, "y")
    B, L = y.shape
    out = torch.empty_like(y)
    _lib.call("nsg_audio_inv_preemphasis", _p(y), _p(out), c_int32(B), c_int32(L), c_float(k), _stream())
    return out


def inv_mel_spectrogram(mel_spectrogram, sample_rate=22050, fft_size=1024, hop_size=256, n_mel=80, iters=GRIFFIN_LIM_ITERS,
                        angles0=None, device="cuda:0"):
    """audio_tacotron.py:99-116.  mel_spectrogram: numpy (n_mel, T) as in the reference (returns a float32 numpy waveform), or a
    GPU tensor (B, n_mel, T) (returns a (B, hop*(T-1)) tensor)."""
    as_numpy = isinstance(mel_spectrogram, np.ndarray)
    mel = torch.from_numpy(np.ascontiguousarray(mel_spectrogram, dtype=np.float32)).to(device) if as_numpy else mel_spectrogram
    if mel.dim() == 2:
        mel = mel.unsqueeze(0)
    mel = mel.contiguous().float()
    S = mel_to_linear(mel, sample_rate, fft_size, n_mel)
    if angles0 is not None and not torch.is_tensor(angles0):
        angles0 = torch.from_numpy(np.ascontiguousarray(angles0, dtype=np.float32)).to(mel.device)
    y = inv_preemphasis(griffin_lim(S, fft_size, hop_size, iters, angles0))
    return y[0].cpu().numpy() if as_numpy else y


def save_wav(wav, path, sample_rate=22050):
    """audio_tacotron.py:15-18: peak-normalise to int16 and write."""
    from scipy.io import wavfile
    wav = np.asarray(wav, dtype=np.float32)
    wav = wav * (32767 / max(0.01, float(np.max(np.abs(wav)))))
    wavfile.write(path, sample_rate, wav.astype(np.int16))
