#!/usr/bin/env python3
"""Generates tests/golden/mel.npz by importing the UNMODIFIED reference front-end (container only).

/root/reference/mel_processing.py imports librosa.filters.mel at module level and librosa is not installable
here, so a stub ``librosa`` is installed whose ``filters.mel`` returns this repo's Slaney filter bank
(quickvc_official_amd.frontend.mel_basis).  Everything else -- reflect padding, Hann STFT, magnitude with the 1e-6
floor, matmul with the basis, log(clamp 1e-5) -- is the reference's own code (mel_processing.py:15-98).  So the
fixtures pin the STFT / log-mel arithmetic to the reference; the filter bank itself stays PARITY UNPINNED.

Fixtures (data only): for three waveform lengths, ``wave_to_spec`` (strided subsample + sum of squares) and
``wave_to_mel`` (whole), plus the seeds that regenerate the waveforms.

    python tests/golden/make_golden_mel.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

LENGTHS = (4000, 16000, 30001)          # shortest ~ 0.25 s, a whole second, a ragged length
N_FFT, HOP, WIN, N_MELS, SR, FMIN, FMAX = 1280, 320, 1280, 80, 16000, 0.0, None    # logs/quickvc/config.json:25-36


def synth_wave(n: int, seed: int) -> torch.Tensor:
    """Deterministic speech-like test signal in [-1, 1]: a few gliding partials plus noise."""
    rs = np.random.RandomState(seed)
    t = np.arange(n) / SR
    x = np.zeros(n)
    for f0, a in ((110.0, 0.4), (220.0, 0.25), (1333.0, 0.1), (3100.0, 0.05)):
        x += a * np.sin(2 * np.pi * (f0 * t + 15.0 * t * t) + rs.uniform(0, 6.28))
    x += 0.05 * rs.randn(n)
    x *= 0.5 + 0.5 * np.sin(2 * np.pi * 1.5 * t) ** 2
    return torch.from_numpy(np.clip(x, -1.0, 1.0).astype(np.float32)).unsqueeze(0)


def main() -> None:
    from quickvc_official_amd.frontend import mel_basis
    lib = types.ModuleType("librosa")
    filt = types.ModuleType("librosa.filters")
    filt.mel = lambda sr, n_fft, n_mels, fmin, fmax: mel_basis(sr, n_fft, n_mels, fmin, fmax)
    lib.filters = filt
    sys.modules["librosa"], sys.modules["librosa.filters"] = lib, filt
    sys.path.insert(0, REFERENCE)
    import mel_processing as ref                     # the reference module itself
    import qvc_oracle as oracle

    arrays = {}
    for i, n in enumerate(LENGTHS):
        wave = synth_wave(n, 900 + i)
        with torch.no_grad():
            spec = ref.wave_to_spec(wave, N_FFT, HOP, WIN)
            mel = ref.wave_to_mel(wave, N_FFT, N_MELS, SR, HOP, WIN, FMIN, FMAX)
        flat = spec.reshape(-1)
        stride = max(1, -(-flat.numel() // 8192))
        arrays[f"spec{n}"] = flat[::stride].numpy().copy()
        arrays[f"spec{n}::shape"] = np.asarray(spec.shape, dtype=np.int64)
        arrays[f"spec{n}::sumsq"] = np.asarray([float(spec.double().pow(2).sum())])
        arrays[f"mel{n}"] = mel.numpy().copy()
        # the oracle's restatement must reproduce the reference (this pins it)
        o_spec = oracle.wave_to_spec(wave, N_FFT, HOP, WIN)
        o_mel = oracle.wave_to_mel(wave, torch.from_numpy(mel_basis(SR, N_FFT, N_MELS, FMIN, FMAX)), N_FFT, HOP, WIN)
        e_spec = float((o_spec - spec).abs().max() / spec.abs().max())
        e_mel = float((o_mel - mel).abs().max())
        print(f"len {n}: spec {tuple(spec.shape)} mel {tuple(mel.shape)}  oracle vs reference: spec rel {e_spec:.2e}, log-mel abs {e_mel:.2e}")
        assert e_spec < 1e-5 and e_mel < 1e-4
    arrays["lengths"] = np.asarray(LENGTHS, dtype=np.int64)
    arrays["seeds"] = np.asarray([900 + i for i in range(len(LENGTHS))], dtype=np.int64)
    path = os.path.join(HERE, "mel.npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
