"""Deterministic synthetic checkpoint + inputs in the reference's formats.

No pretrained ``quickvc.pth`` and no HuBERT exist offline, so tests and
``bench.py`` run on a synthetic checkpoint with the reference's exact state-dict
keys/shapes, and on synthetic units / speaker embeddings / noise.  Everything is
drawn from ``numpy.random.RandomState`` (a frozen legacy stream: the same seed gives
the same bytes on every numpy version and machine), never from torch's generator,
so the golden vectors made in the build container can be regenerated on the GPU box.

Weight statistics are chosen so that every layer matters numerically (a fresh
reference init would not: SURVEY 0.6/0.7): effective conv weights have variance
gain^2/fan_in, ``weight_g`` is ||v|| times an independent U(0.6,1.4) factor (so the
fold g*v/||v|| is really exercised), and ``flow.*.post`` is non-zero.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch


def _fan_in(name: str, shape: Tuple[int, ...]) -> float:
    if ".ups." in name:                 # ConvTranspose1d (Cin, Cout, K): ~Cin*K/stride terms per output
        return shape[0] * shape[2] / 4.5
    if len(shape) == 3:
        return shape[1] * shape[2]
    if len(shape) == 2:
        return shape[1]
    return 1.0


def make_synthetic_state_dict(model: torch.nn.Module, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """A full ``state_dict`` for ``model`` (reference key order), fp32 CPU tensors."""
    rng = np.random.RandomState(seed)
    ref = model.state_dict()
    out: Dict[str, torch.Tensor] = {}
    pending_v: Dict[str, np.ndarray] = {}
    for name, t in ref.items():
        shape = tuple(t.shape)
        if name.endswith(("updown_filter", "stft.window", "analysis_filter", "synthesis_filter")):
            out[name] = t.detach().float().cpu().clone()          # fixed buffers keep their values
            continue
        draw = rng.standard_normal(size=shape).astype(np.float64)
        if name.endswith(".weight_g"):
            out[name] = None                                       # filled once weight_v is known
            pending_v[name[:-len("_g")] + "_v"] = 0.6 + 0.8 * rng.random_sample(size=shape)
            continue
        if name.endswith(".bias") or ".bias_" in name:
            scale = 0.05
            if name == "enc_p.proj.bias" or name == "enc_q.proj.bias":
                val = draw * scale
                val[shape[0] // 2:] -= 1.0                         # log-sigma half centred at -1
                out[name] = torch.from_numpy(val.astype(np.float32))
                continue
            out[name] = torch.from_numpy((draw * scale).astype(np.float32))
            continue
        gain = 1.0
        if "conv_post" in name and "multistream" not in name:
            gain = 0.5                                             # keeps exp(log-magnitude) in a sane range
        if name.endswith("proj.weight"):
            gain = 0.5
        if "enc_spk.lstm" in name or "enc_spk.linear" in name:
            val = (rng.random_sample(size=shape) * 2.0 - 1.0) / np.sqrt(max(shape[-1], 1))
            out[name] = torch.from_numpy(val.astype(np.float32))
            continue
        val = draw * (gain / np.sqrt(_fan_in(name, shape)))
        out[name] = torch.from_numpy(val.astype(np.float32))
    # weight_g = ||v|| * independent factor (per dim-0 slice)
    for vname, factor in pending_v.items():
        v = out[vname].double().numpy()
        norm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(axis=1)).reshape(factor.shape)
        out[vname[:-len("_v")] + "_g"] = torch.from_numpy((norm * factor).astype(np.float32))
    assert all(v is not None for v in out.values())
    return out


def make_synthetic_inputs(batch: int, frames: int, unit_channels: int, inter_channels: int, gin_channels: int,
                          seed0: int = 0):
    """Per-utterance synthetic inputs, seeds ``seed0 .. seed0+batch-1``.

    unit ~ N(0,1) (B,256,T); g = L2-normalised |N(0,1)| (B,gin) (post-ReLU, unit norm like
    models.py:517-518); noise ~ N(0,1) (B,inter,T).
    """
    units, gs, noises = [], [], []
    for b in range(batch):
        rng = np.random.RandomState(seed0 + b)
        units.append(rng.standard_normal(size=(unit_channels, frames)).astype(np.float32))
        gv = np.abs(rng.standard_normal(size=(gin_channels,)))
        gs.append((gv / np.sqrt((gv ** 2).sum())).astype(np.float32))
        noises.append(rng.standard_normal(size=(inter_channels, frames)).astype(np.float32))
    return (torch.from_numpy(np.stack(units)), torch.from_numpy(np.stack(gs)), torch.from_numpy(np.stack(noises)))


def make_synthetic_posterior_inputs(batch: int, frames: int, spec_channels: int, inter_channels: int, gin_channels: int,
                                    seed0: int = 50):
    """Inputs of the posterior direction (models.py:617-618): a linear-spectrogram-like spec = |N(0,1)| (B, spec, T)
    (magnitudes are non-negative, mel_processing.py:56), g and noise as in ``make_synthetic_inputs``."""
    specs, gs, noises = [], [], []
    for b in range(batch):
        rng = np.random.RandomState(seed0 + b)
        specs.append(np.abs(rng.standard_normal(size=(spec_channels, frames))).astype(np.float32))
        gv = np.abs(rng.standard_normal(size=(gin_channels,)))
        gs.append((gv / np.sqrt((gv ** 2).sum())).astype(np.float32))
        noises.append(rng.standard_normal(size=(inter_channels, frames)).astype(np.float32))
    return (torch.from_numpy(np.stack(specs)), torch.from_numpy(np.stack(gs)), torch.from_numpy(np.stack(noises)))


def make_synthetic_mel(frames: int, n_mel: int = 80, seed: int = 7) -> torch.Tensor:
    """A log-mel-like (1, n_mel, frames) tensor for the speaker-encoder path."""
    rng = np.random.RandomState(seed)
    return torch.from_numpy((rng.standard_normal(size=(1, n_mel, frames)) * 1.5 - 4.0).astype(np.float32))
