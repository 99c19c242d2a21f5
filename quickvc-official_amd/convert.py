"""Inference CLI with the reference's surface (convert.py:19-86): same flags and defaults, same
``title|src|tgt`` list format, writes ``{outdir}/{title}.wav`` as float32 at ``data.sampling_rate``.

Differences forced by the offline environment: HuBERT-soft cannot be fetched, so the
source side takes pre-extracted units -- ``src`` may be a ``.npy`` file of shape (frames, 256) fp32
(the reference's own on-disk unit format, dataset/encode.py:38) or a ``.wav`` that has such a file
next to it (``x.wav`` -> ``x.npy``).  New optional flags: ``--batch`` (utterances converted per
launch), ``--seed`` (noise), ``--dtype``.

Corpus scale (BASELINE.json configs[3]): started under ``torch.distributed.run`` (one process per
GPU) every rank converts its own static shard of the list (length-sorted round-robin,
``dist.shard_indices``); nothing is exchanged between ranks -- each rank loads the checkpoint itself, so
there is not even a start-up broadcast in this mode.

    python -m quickvc_official_amd.convert --hpfile logs/quickvc/config.json --ptfile quickvc.pth
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

from .checkpoint import load_checkpoint
from .dist import env_world, shard_indices
from .config import get_hparams_from_file
from .frontend import MelFrontend, load_wav, trim, wave_to_mel
from .model import SynthesizerTrn


def _load_units(src: str) -> torch.Tensor:
    path = src if src.endswith(".npy") else os.path.splitext(src)[0] + ".npy"
    if not os.path.exists(path):
        raise FileNotFoundError(f"no unit file for {src}: HuBERT-soft is not available offline; provide {path} "
                                "(frames, 256) fp32 as written by the reference's dataset/encode.py")
    u = np.load(path).astype(np.float32)
    if u.ndim != 2 or u.shape[1] != 256:
        raise ValueError(f"{path}: expected (frames, 256), got {u.shape}")
    return torch.from_numpy(u).t().unsqueeze(0)               # (1, 256, frames), data_utils_new_new.py:121-122


def plan_batches(lengths, batch: int):
    """Utterances of equal unit length share a launch (the path has no masks, so padding would change
    the result near the end); returns lists of item indices, longest first."""
    by_len = {}
    for i, n in enumerate(lengths):
        by_len.setdefault(int(n), []).append(i)
    plan = []
    for n in sorted(by_len, reverse=True):
        group = by_len[n]
        plan.extend(group[i:i + batch] for i in range(0, len(group), batch))
    return plan


def main(argv=None) -> None:
    p = argparse.ArgumentParser()
    p.add_argument("--hpfile", type=str, default="logs/quickvc/config.json", help="path to json config file")
    p.add_argument("--ptfile", type=str, default="logs/quickvc/quickvc.pth", help="path to pth file")
    p.add_argument("--txtpath", type=str, default="convert.txt", help="path to txt file")
    p.add_argument("--outdir", type=str, default="output/quickvc", help="path to output dir")
    p.add_argument("--use_timestamp", default=False, action="store_true")
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    args = p.parse_args(argv)

    from scipy.io.wavfile import write
    os.makedirs(args.outdir, exist_ok=True)
    hps = get_hparams_from_file(args.hpfile)
    rank, local_rank, world = env_world()
    torch.cuda.set_device(local_rank)
    print("Loading model...")
    net_g = SynthesizerTrn(hps.data.filter_length // 2 + 1, hps.train.segment_size // hps.data.hop_length,
                           **hps.model, operand_dtype=args.dtype).cuda().eval()
    print("Number of parameter: %.2fM" % (sum(p.nelement() for p in net_g.parameters()) / 1e6))
    print("Loading checkpoint...")
    load_checkpoint(args.ptfile, net_g, None)

    items = []
    with open(args.txtpath, "r") as f:
        for raw in f.readlines():
            if raw.strip():
                title, src, tgt = raw.strip().split("|")
                items.append((title, src, tgt))
    if args.seed is not None:
        torch.manual_seed(args.seed)

    print("Synthesizing...")
    d = hps.data
    with torch.no_grad():
        # speaker embeddings once per distinct target (the reference recomputes them per line)
        g_cache = {}
        prepared = []
        # mel front-end on the GPU (qvc_wave_to_mel) when the config has win == n_fft, else the torch restatement
        front = MelFrontend(d.filter_length, d.n_mel_channels, d.sampling_rate, d.hop_length, d.win_length,
                            d.mel_fmin, d.mel_fmax) if d.win_length == d.filter_length and d.hop_length % 16 == 0 else None
        for title, src, tgt in items:
            if tgt not in g_cache:
                wav = torch.from_numpy(trim(load_wav(tgt, d.sampling_rate), top_db=20)).unsqueeze(0).cuda()
                mel = front(wav) if front is not None else wave_to_mel(wav, d.filter_length, d.n_mel_channels, d.sampling_rate,
                                                                       d.hop_length, d.win_length, d.mel_fmin, d.mel_fmax)
                g_cache[tgt] = net_g.speaker_embed(mel)                 # (1, 80, F') -> (1, gin), HIP LSTM
            prepared.append((title, _load_units(src), g_cache[tgt]))
        # this rank's shard of the list, then equal-length utterances share a launch
        mine = shard_indices(len(prepared), rank, world, [p[1].shape[-1] for p in prepared])
        prepared = [prepared[i] for i in mine]
        for idxs in plan_batches([p[1].shape[-1] for p in prepared], args.batch):
            chunk = [prepared[i] for i in idxs]
            unit = torch.cat([u for _, u, _ in chunk], 0).cuda()
            g = torch.cat([gg for _, _, gg in chunk], 0)
            audio = net_g.infer_batch(unit, g)
            for (title, _, _), a in zip(chunk, audio):
                name = f"{time.strftime('%m-%d_%H-%M', time.localtime())}_{title}.wav" if args.use_timestamp else f"{title}.wav"
                write(os.path.join(args.outdir, name), d.sampling_rate, a[0].float().cpu().numpy())


if __name__ == "__main__":
    main()
