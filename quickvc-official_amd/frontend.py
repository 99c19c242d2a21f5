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


class MelFrontend:
    """``wave_to_mel`` on the GPU through the C ABI (qvc_wave_to_mel, csrc/qvc_mel.hip): fp32 STFT as a GEMM
    against a windowed DFT table on the f32 MFMA, magnitude, sparse mel filters, log -- two launches, no FFT plan.

    Same arguments as mel_processing.wave_to_mel (win must equal n_fft, as in the shipped config); the mel filter
    bank is ``mel_basis`` above (parity-unpinned against librosa, see the module docstring).
    """

    def __init__(self, n_fft: int, n_mels: int, sr: int, hop: int, win: int, fmin: float, fmax, device="cuda"):
        import ctypes
        from . import lib as L
        if win != n_fft:
            raise ValueError("the HIP front-end needs win_length == filter_length")
        self.lib = L.load_library()
        self.n_fft, self.hop, self.n_mels = int(n_fft), int(hop), int(n_mels)
        self.device = torch.device(device)
        basis = np.ascontiguousarray(mel_basis(sr, n_fft, n_mels, fmin, fmax), dtype=np.float32)
        n = int(self.lib.qvc_mel_table_bytes(self.n_fft, self.n_mels))
        if n < 0:
            L.check(self.lib, n, "qvc_mel_table_bytes")
        host = torch.empty(n, dtype=torch.uint8)
        L.check(self.lib, self.lib.qvc_mel_pack_tables(self.n_fft, self.hop, self.n_mels, basis.ctypes.data_as(ctypes.c_void_p),
                                                       host.data_ptr(), n), "qvc_mel_pack_tables")
        self.table = host.to(self.device)
        self._ws = None

    def frames(self, samples: int) -> int:
        pad = (self.n_fft - self.hop) // 2
        return (samples + 2 * pad - self.n_fft) // self.hop + 1

    @torch.no_grad()
    def __call__(self, wave: torch.Tensor) -> torch.Tensor:
        """wave (U, samples) in [-1, 1] -> log-mel (U, n_mels, frames) fp32 on the device."""
        from . import lib as L
        if wave.dim() != 2:
            raise ValueError(f"wave must be (utterances, samples), got {tuple(wave.shape)}")
        wave = wave.to(device=self.device, dtype=torch.float32).contiguous()
        U, N = wave.shape
        n = int(self.lib.qvc_mel_workspace_bytes(self.n_fft, self.hop, U, N))
        if n < 0:
            L.check(self.lib, n, "qvc_mel_workspace_bytes")
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        mel = torch.empty(U, self.n_mels, self.frames(N), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):     # the library keys per-device launch state on the current device
            st = self.lib.qvc_wave_to_mel(self.table.data_ptr(), self.n_fft, self.hop, self.n_mels, wave.data_ptr(), mel.data_ptr(),
                                          U, N, self._ws.data_ptr(), self._ws.numel(),
                                          torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib, st, "qvc_wave_to_mel")
        return mel


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
