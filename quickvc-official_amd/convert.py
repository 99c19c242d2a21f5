"""Inference CLI with the reference's surface (convert.py:19-86): same flags and defaults, same
``title|src|tgt`` list format, writes ``{outdir}/{title}.wav`` as float32 at ``data.sampling_rate``.

Differences forced by the offline environment: HuBERT-soft cannot be fetched, so the
source side takes pre-extracted units -- ``src`` may be a ``.npy`` file of shape (frames, 256) fp32
(the reference's own on-disk unit format, dataset/encode.py:38) or a ``.wav`` that has such a file
next to it (``x.wav`` -> ``x.npy``).  New optional flags: ``--batch`` (utterances converted per
launch), ``--seed`` (noise), ``--dtype``.

Corpus scale (BASELINE.json configs[3]): started under ``torch.distributed.run`` (one process per
GPU) every rank converts its own static shard of the list (length-sorted round-robin,
``dist.shard_indices``, decided from the .npy headers alone: a rank never loads another rank's units or
targets); nothing is exchanged between ranks -- each rank loads the checkpoint itself, so there is not
even a start-up broadcast in this mode.  Utterances of different lengths share launches through the
ragged path (``qvc_infer_batch_ragged``): batches are cut from the length-sorted shard.

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
from .frontend import MelFrontend, load_wav, trim
from .model import SynthesizerTrn


def _unit_path(src: str) -> str:
    path = src if src.endswith(".npy") else os.path.splitext(src)[0] + ".npy"
    if not os.path.exists(path):
        raise FileNotFoundError(f"no unit file for {src}: HuBERT-soft is not available offline; provide {path} "
                                "(frames, 256) fp32 as written by the reference's dataset/encode.py")
    return path


def unit_frames(src: str) -> int:
    """Unit-frame count of a source from the .npy HEADER only (memory-mapped: no payload is read)."""
    u = np.load(_unit_path(src), mmap_mode="r")
    if u.ndim != 2 or u.shape[1] != 256:
        raise ValueError(f"{_unit_path(src)}: expected (frames, 256), got {u.shape}")
    return int(u.shape[0])


def _load_units(src: str) -> torch.Tensor:
    u = np.load(_unit_path(src)).astype(np.float32)
    if u.ndim != 2 or u.shape[1] != 256:
        raise ValueError(f"{_unit_path(src)}: expected (frames, 256), got {u.shape}")
    return torch.from_numpy(u).t().unsqueeze(0)               # (1, 256, frames), data_utils_new_new.py:121-122


def plan_batches(lengths, batch: int, max_pad: float = 0.25):
    """Length-bucketed batches for the ragged path: items sorted by decreasing length, then cut into runs of at most
    ``batch`` whose shortest member is at least (1 - max_pad) of the longest (padding is skipped work per tile, but
    a batch still runs as long as its longest utterance).  Returns lists of item indices, longest first."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    plan, cur = [], []
    for i in order:
        if cur and (len(cur) >= batch or int(lengths[i]) < (1.0 - max_pad) * int(lengths[cur[0]])):
            plan.append(cur)
            cur = []
        cur.append(i)
    if cur:
        plan.append(cur)
    return plan


def rank_plan(items, rank: int, world: int, batch: int):
    """This rank's work for a ``title|src|tgt`` list: (lengths of ALL sources -- headers only --, this rank's item
    indices, its ragged batches as lists of GLOBAL item indices).  Pure host logic: nothing but .npy headers is read,
    so every rank can plan the whole corpus while loading only its own share."""
    lengths = [unit_frames(src) for _, src, _ in items]
    mine = shard_indices(len(items), rank, world, lengths)
    batches = [[mine[i] for i in idxs] for idxs in plan_batches([lengths[i] for i in mine], batch)]
    return lengths, mine, batches


def main(argv=None) -> None:
    p = argparse.ArgumentParser()
    p.add_argument("--hpfile", type=str, default="logs/quickvc/config.json", help="path to json config file")
    p.add_argument("--ptfile", type=str, default="logs/quickvc/quickvc.pth", help="path to pth file")
    p.add_argument("--txtpath", type=str, default="convert.txt", help="path to txt file")
    p.add_argument("--outdir", type=str, default="output/quickvc", help="path to output dir")
    p.add_argument("--use_timestamp", default=False, action="store_true")
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--dtype", default="f16", choices=["f16", "bf16", "bf16x"])
    p.add_argument("--device", type=int, default=None, help="GPU ordinal (default: LOCAL_RANK); rehearsals of several ranks on one GPU pass 0")
    args = p.parse_args(argv)

    from scipy.io.wavfile import write
    os.makedirs(args.outdir, exist_ok=True)
    hps = get_hparams_from_file(args.hpfile)
    rank, local_rank, world = env_world()
    torch.cuda.set_device(local_rank if args.device is None else args.device)
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
        torch.manual_seed(args.seed + rank)

    print("Synthesizing...")
    d = hps.data
    with torch.no_grad():
        # Shard FIRST: the split only needs every source's length, read from the .npy headers; units are loaded
        # and targets embedded for this rank's own items only (O(corpus / world) work and memory per rank).
        _lengths, _mine, batches = rank_plan(items, rank, world, args.batch)
        # mel front-end on the GPU (qvc_wave_to_mel; raises for configs it does not cover -- there is no CPU path)
        front = MelFrontend(d.filter_length, d.n_mel_channels, d.sampling_rate, d.hop_length, d.win_length,
                            d.mel_fmin, d.mel_fmax)
        g_cache = {}                                # speaker embeddings once per distinct target (the reference recomputes per line)
        for idxs in batches:
            chunk = [items[i] for i in idxs]
            for _, _, tgt in chunk:
                if tgt not in g_cache:
                    wav = torch.from_numpy(trim(load_wav(tgt, d.sampling_rate), top_db=20)).unsqueeze(0).cuda()
                    g_cache[tgt] = net_g.speaker_embed(front(wav))          # (1, 80, F') -> (1, gin), HIP mel + HIP LSTM
            units = [_load_units(src) for _, src, _ in chunk]
            g = torch.cat([g_cache[tgt] for _, _, tgt in chunk], 0)
            audio = net_g.infer_ragged(units, g)                              # every utterance at its own length
            for (title, _, _), a in zip(chunk, audio):
                name = f"{time.strftime('%m-%d_%H-%M', time.localtime())}_{title}.wav" if args.use_timestamp else f"{title}.wav"
                write(os.path.join(args.outdir, name), d.sampling_rate, a[0].float().cpu().numpy())


if __name__ == "__main__":
    main()
