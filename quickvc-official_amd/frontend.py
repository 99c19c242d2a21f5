"""Host-side front-end pieces either side of the hot path (SURVEY 8f "next" #2/#3).

``wave_to_mel`` restates mel_processing.py:79-98 of the reference (reflect pad, Hann STFT 1280/320
center=False, sqrt(re^2+im^2+1e-6), Slaney-normalised mel basis, log(clamp 1e-5)) and ``trim`` restates
the energy trim convert.py:65 obtains from librosa.  PARITY UNPINNED: librosa is not installable
here, so the mel basis and the trim are written from the published Slaney/librosa
definitions and are not checked against the reference; they sit outside the HIP path and only
feed the speaker encoder.
"""
from __future__ import annotations

import numpy as np
import torch


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    mel = f / (200.0 / 3)
    log_t = f >= 1000.0
    mel = np.where(log_t, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0), mel)
    return mel


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f = m * (200.0 / 3)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), f)


def mel_basis(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax=None) -> np.ndarray:
    """Slaney-style triangular filters with area normalisation -> (n_mels, n_fft//2+1)."""
    fmax = sr / 2.0 if fmax is None else fmax
    fft_f = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.zeros((n_mels, n_fft // 2 + 1))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def wave_to_mel(wave: torch.Tensor, n_fft: int, n_mels: int, sr: int, hop: int, win: int, fmin: float, fmax) -> torch.Tensor:
    """wave (B, T) in [-1, 1] -> log-mel (B, n_mels, frames)."""
    pad = (n_fft - hop) // 2
    x = torch.nn.functional.pad(wave.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    window = torch.hann_window(win, device=wave.device, dtype=wave.dtype)
    spec = torch.stft(x, n_fft, hop_length=hop, win_length=win, window=window, center=False, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    mag = torch.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-6)
    basis = torch.from_numpy(mel_basis(sr, n_fft, n_mels, fmin, fmax)).to(wave.device, wave.dtype)
    return torch.log(torch.clamp(basis @ mag, min=1e-5))


def trim(wave: np.ndarray, top_db: float = 20.0, frame_length: int = 2048, hop_length: int = 512) -> np.ndarray:
    """Drop leading/trailing frames whose RMS is more than ``top_db`` below the peak frame."""
    if len(wave) < frame_length:
        return wave
    pad = frame_length // 2
    x = np.pad(wave, (pad, pad), mode="constant")
    n = 1 + (len(x) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n)[:, None]
    rms = np.sqrt(np.mean(x[idx] ** 2, axis=1))
    db = 20.0 * np.log10(np.maximum(rms, 1e-10)) - 20.0 * np.log10(max(rms.max(), 1e-10))
    keep = np.nonzero(db > -top_db)[0]
    if keep.size == 0:
        return wave[:0]
    return wave[keep[0] * hop_length: min(len(wave), (keep[-1] + 1) * hop_length)]


def load_wav(path: str, sr: int) -> np.ndarray:
    """16-bit / float wav -> float32 mono at ``sr`` (polyphase resampling if needed)."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    rate, data = wavfile.read(path)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == "u":
        data = (data.astype(np.float32) - 128.0) / 128.0
    data = data.astype(np.float32)
    if data.ndim > 1:
        data = data.mean(axis=1)
    if rate != sr:
        from math import gcd
        g = gcd(int(rate), int(sr))
        data = resample_poly(data, sr // g, rate // g).astype(np.float32)
    return data
