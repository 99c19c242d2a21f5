"""Inference CLI with the reference's surface (convert.py:19-86): same flags and defaults, same
``title|src|tgt`` list format, writes ``{outdir}/{title}.wav`` as float32 at ``data.sampling_rate``.

Differences forced by the offline environment: HuBERT-soft cannot be fetched, so the
source side takes pre-extracted units -- ``src`` may be a ``.npy`` file of shape (frames, 256) fp32
(the reference's own on-disk unit format, dataset/encode.py:38) or a ``.wav`` that has such a file
next to it (``x.wav`` -> ``x.npy``).  New optional flags: ``--batch`` (utterances converted per
launch), ``--seed`` (noise), ``--dtype``.

Corpus scale (BASELINE.json configs[3]): the per-line loop of the reference (convert.py:58-86: load, infer, write, one
utterance at a time) becomes a three-stage pipeline (``CorpusPipeline``): a loader thread reads the next batches' unit
files straight into pinned, pre-padded upload buffers (native thread pool, include/qvc_io.h), the main thread uploads,
converts (ragged batch, side stream) and downloads asynchronously, a writer thread writes the wav files of finished
batches -- disk, PCIe and GPU work overlap.  Started under ``torch.distributed.run`` (one process per
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
import queue
import threading
import time

import numpy as np
import torch

from .checkpoint import load_checkpoint
from .dist import env_world, shard_indices
from .config import get_hparams_from_file
from .frontend import MelFrontend, load_wav, trim
from .model import SynthesizerTrn


def _unit_path(src: str, check: bool = True) -> str:
    path = src if src.endswith(".npy") else os.path.splitext(src)[0] + ".npy"
    if check and not os.path.exists(path):
        raise FileNotFoundError(f"no unit file for {src}: HuBERT-soft is not available offline; provide {path} "
                                "(frames, 256) fp32 as written by the reference's dataset/encode.py")
    return path


def unit_frames(src: str) -> int:
    """Unit-frame count of a source from the .npy HEADER only (qvc_io_npy_shape: no payload is read)."""
    from .fileio import npy_shape
    frames, cols = npy_shape(_unit_path(src))
    if cols != 256:
        raise ValueError(f"{_unit_path(src)}: expected (frames, 256), got ({frames}, {cols})")
    return frames


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


def rank_plan(items, rank: int, world: int, batch: int, pool=None):
    """This rank's work for a ``title|src|tgt`` list: (lengths of ALL sources -- headers only --, this rank's item
    indices, its ragged batches as lists of GLOBAL item indices).  Pure host logic: nothing but .npy headers is read
    (on ``pool``, a fileio.IoPool, in parallel when given), so every rank can plan the whole corpus while loading only
    its own share."""
    if pool is not None and items:
        paths = [_unit_path(src, check=False) for _, src, _ in items]     # a missing file fails in the native open, with its error
        try:
            shapes = pool.npy_shapes(paths)
        except Exception:
            for _, src, _ in items:
                _unit_path(src)                                           # says WHICH unit file is missing (and why there must be one)
            raise
        for path, (_f, c) in zip(paths, shapes):
            if c != 256:
                raise ValueError(f"{path}: expected (frames, 256), got ({_f}, {c})")
        lengths = [int(f) for f, _ in shapes]
    else:
        lengths = [unit_frames(src) for _, src, _ in items]
    mine = shard_indices(len(items), rank, world, lengths)
    batches = [[mine[i] for i in idxs] for idxs in plan_batches([lengths[i] for i in mine], batch)]
    return lengths, mine, batches


def batch_noise(seed: int, key: int, n: int, inter: int, frames: int, device) -> torch.Tensor:
    """The N(0,1) draw of models.py:94 for one batch of the pipeline: (n, inter, frames) from a generator seeded with
    (seed, key) -- key = the global index of the batch's first list line -- so a run is reproducible per batch whatever
    the sharding, and a test can regenerate the draw of any utterance."""
    gen = torch.Generator(device=device)
    gen.manual_seed((int(seed) * 1000003 + int(key)) & 0x7FFFFFFFFFFFFFFF)
    return torch.randn((n, inter, frames), generator=gen, device=device, dtype=torch.float32)


class CorpusPipeline:
    """Corpus-scale conversion of pre-planned ragged batches (BASELINE.json configs[3]; reference loop: convert.py:58-86).

    ``slots`` buffer sets rotate through three stages that run concurrently:
      load   (loader thread + native I/O pool): unit files -> pinned (B, Tmax, 256) fp32, frame-major as on disk;
      convert (caller's thread, side streams):  async upload, qvc_infer_batch_ragged_fm, async download into pinned memory;
      write  (writer thread + native I/O pool): float32 wav files, byte-identical to scipy.io.wavfile.write.
    ``lanes`` batches are converted at the same time, each on a compute stream and workspace of its own: the kernels of
    batch k+1 fill the launch tails and dependency gaps of batch k (about -6 % per batch at 32 x 5 s, bench.py
    --in-flight); the results do not depend on it.
    """

    def __init__(self, net_g, batch: int, max_frames: int, sampling_rate: int, slots: int = 4, io_threads: int = 8,
                 seed: int = 0, lanes: int = 2):
        from .fileio import IoPool
        self.net_g, self.engine = net_g, net_g.engine()
        self.dev = self.engine.device
        self.batch, self.max_frames, self.rate, self.seed = int(batch), int(max_frames), int(sampling_rate), int(seed)
        self.spf = net_g.samples_per_frame
        self.inter = net_g.model_config["inter_channels"]
        self.uc = net_g.model_config.get("unit_channels", 256)
        self.load_pool, self.write_pool = IoPool(io_threads), IoPool(io_threads)
        B, Tm = self.batch, self.max_frames
        self.slots = []
        self.lanes = max(1, int(lanes))
        for _ in range(max(self.lanes + 1, int(slots))):
            self.slots.append(dict(
                unit_pin=torch.empty(B * Tm * self.uc, dtype=torch.float32).pin_memory(),
                lens_pin=torch.empty(B, dtype=torch.int32).pin_memory(),
                out_pin=torch.empty(B * Tm * self.spf, dtype=torch.float32).pin_memory(),
                unit_dev=torch.empty(B * Tm * self.uc, dtype=torch.float32, device=self.dev),
                lens_dev=torch.empty(B, dtype=torch.int32, device=self.dev),
                out_dev=torch.empty(B * Tm * self.spf, dtype=torch.float32, device=self.dev),
                done=torch.cuda.Event()))
        for s in self.slots:
            s["up"], s["comp"] = torch.cuda.Event(), torch.cuda.Event()
        self.ws = [self.engine.alloc_workspace(B, Tm) for _ in range(self.lanes)]
        # uploads, kernels and downloads on streams of their own: batch k+1 travels to the GPU and batch k-1 back to
        # the host while batch k computes (one stream would serialise 1 GB of PCIe traffic with the kernels)
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(self.lanes)]
        self.up_stream, self.down_stream = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        # where the threads spend their time (seconds): *_wait = blocked on the neighbouring stage
        self.stats = {"utterances": 0, "samples": 0, "batches": 0, "load_s": 0.0, "load_wait_s": 0.0, "enqueue_s": 0.0,
                      "enqueue_wait_s": 0.0, "write_s": 0.0, "write_wait_s": 0.0, "gpu_wait_s": 0.0, "embed_s": 0.0}

    def close(self) -> None:
        self.load_pool.close()
        self.write_pool.close()

    def run(self, batches, src_paths, out_paths, lengths, g_rows) -> None:
        """``batches``: lists of item indices (longest utterance first in each); ``src_paths[i]`` / ``out_paths[i]`` /
        ``lengths[i]`` (unit frames) per item; ``g_rows``: device tensor (n_items, gin) of speaker embeddings per item,
        or a callable returning it -- called AFTER the loader thread has started, so that embedding the targets
        overlaps reading the first batches."""
        free_q, ready_q, done_q = queue.Queue(), queue.Queue(), queue.Queue()
        for s in self.slots:
            free_q.put(s)
        errors = []

        def loader():
            try:
                for idxs in batches:
                    tw = time.perf_counter()
                    s = free_q.get()
                    tl = time.perf_counter()
                    self.stats["load_wait_s"] += tl - tw
                    n, tmax = len(idxs), max(int(lengths[i]) for i in idxs)
                    if n > self.batch or tmax > self.max_frames:
                        raise ValueError(f"batch of {n} x {tmax} frames exceeds the pipeline's buffers ({self.batch} x {self.max_frames})")
                    dst = s["unit_pin"][:n * tmax * self.uc].view(n, tmax, self.uc)
                    self.load_pool.load_units([src_paths[i] for i in idxs], dst, s["lens_pin"])
                    if [int(v) for v in s["lens_pin"][:n]] != [int(lengths[i]) for i in idxs]:
                        raise ValueError("a unit file changed its length since the run was planned")
                    self.stats["load_s"] += time.perf_counter() - tl
                    ready_q.put((s, idxs, n, tmax))
            except Exception as exc:                          # noqa: BLE001 -- reported by run()
                errors.append(exc)
            finally:
                ready_q.put(None)

        def writer():
            try:
                while True:
                    tw = time.perf_counter()
                    item = done_q.get()
                    if item is None:
                        return
                    s, idxs, n, tmax = item
                    tg = time.perf_counter()
                    s["done"].synchronize()
                    t0w = time.perf_counter()
                    out = s["out_pin"][:n * tmax * self.spf].view(n, tmax * self.spf)
                    self.write_pool.write_wavs([out_paths[i] for i in idxs], out, [int(lengths[i]) * self.spf for i in idxs], self.rate)
                    free_q.put(s)
                    t1w = time.perf_counter()
                    self.stats["write_wait_s"] += tg - tw
                    self.stats["gpu_wait_s"] += t0w - tg
                    self.stats["write_s"] += t1w - t0w
            except Exception as exc:                          # noqa: BLE001
                errors.append(exc)
                while done_q.get() is not None:               # drain so that the main thread never blocks on us
                    pass

        lt, wt = threading.Thread(target=loader, daemon=True), threading.Thread(target=writer, daemon=True)
        lt.start(); wt.start()
        if callable(g_rows):
            te = time.perf_counter()
            g_rows = g_rows()
            torch.cuda.synchronize(self.dev)
            self.stats["embed_s"] = time.perf_counter() - te
        # every batch's item indices on the device up front: a per-batch host -> device copy of a Python list is a
        # SYNCHRONOUS copy on the compute stream -- it made the host wait for the previous batch's kernels before it could
        # enqueue the next ones (2.75 ms per batch against 2.2 ms of kernels)
        flat = [i for idxs in batches for i in idxs]
        idx_dev = torch.tensor(flat, dtype=torch.int64).pin_memory().to(self.dev, non_blocking=True)
        offs, o = [], 0
        for idxs in batches:
            offs.append(o)
            o += len(idxs)
        bi = 0
        for st in self.streams:
            st.wait_stream(torch.cuda.current_stream(self.dev))          # g_rows may still be pending there
        try:
            with torch.no_grad():
                while True:
                    tw = time.perf_counter()
                    item = ready_q.get()
                    te = time.perf_counter()
                    self.stats["enqueue_wait_s"] += te - tw
                    if item is None or errors:
                        break
                    s, idxs, n, tmax = item
                    unit = s["unit_dev"][:n * tmax * self.uc].view(n, tmax, self.uc)
                    lens = s["lens_dev"][:n]
                    with torch.cuda.stream(self.up_stream):
                        unit.copy_(s["unit_pin"][:n * tmax * self.uc].view(n, tmax, self.uc), non_blocking=True)
                        lens.copy_(s["lens_pin"][:n], non_blocking=True)
                        s["up"].record(self.up_stream)
                    lane = bi % self.lanes                    # a stream's order keeps a lane's workspace to one batch at a time
                    stream = self.streams[lane]
                    with torch.cuda.stream(stream):
                        noise = batch_noise(self.seed, idxs[0], n, self.inter, tmax, self.dev)
                        g = g_rows.index_select(0, idx_dev[offs[bi]:offs[bi] + n])
                        out = s["out_dev"][:n * tmax * self.spf].view(n, 1, tmax * self.spf)
                        stream.wait_event(s["up"])
                        self.engine.infer_batch_ragged(unit, g, noise, lens, out=out, ws=self.ws[lane], unit_fm=True)
                        s["comp"].record(stream)
                        # the caching allocator hands noise / g back to THIS stream's pool once they go out of scope
                    bi += 1
                    with torch.cuda.stream(self.down_stream):
                        self.down_stream.wait_event(s["comp"])
                        s["out_pin"][:n * tmax * self.spf].copy_(out.view(-1), non_blocking=True)
                        s["done"].record(self.down_stream)
                    done_q.put(item)
                    self.stats["enqueue_s"] += time.perf_counter() - te
                    self.stats["utterances"] += n
                    self.stats["samples"] += sum(int(lengths[i]) for i in idxs) * self.spf
                    self.stats["batches"] += 1
        finally:
            done_q.put(None)
            wt.join()
            # a loader stuck on a full free-queue cannot happen (slots only return); if it is still reading, let it finish
            lt.join(timeout=60)
        if errors:
            raise errors[0]


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
    p.add_argument("--io-threads", type=int, default=8, help="native I/O worker threads per direction")
    p.add_argument("--lanes", type=int, default=2, help="batches converted at the same time (compute streams); the output does not depend on it")
    args = p.parse_args(argv)

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
    seed = args.seed if args.seed is not None else int.from_bytes(os.urandom(4), "little")

    print("Synthesizing...")
    convert_items(net_g, hps.data, items, args.outdir, rank, world, args.batch, seed, args.use_timestamp, args.io_threads,
                  lanes=args.lanes)


def convert_items(net_g, d, items, outdir: str, rank: int = 0, world: int = 1, batch: int = 32, seed: int = 0,
                  use_timestamp: bool = False, io_threads: int = 8, timings: dict = None, lanes: int = 2):
    """Convert this rank's shard of ``items`` = [(title, src, tgt)] into ``outdir`` (the body of convert.py:58-86)."""
    from .fileio import IoPool
    t0 = time.perf_counter()
    with torch.no_grad():
        # Shard FIRST: the split only needs every source's length, read from the .npy headers; units are loaded
        # and targets embedded for this rank's own items only (O(corpus / world) work and memory per rank).
        plan_pool = IoPool(io_threads)
        try:
            lengths, mine, batches = rank_plan(items, rank, world, batch, pool=plan_pool)
        finally:
            plan_pool.close()
        if not mine:
            return {"utterances": 0, "samples": 0, "batches": 0}
        t1 = time.perf_counter()
        n_targets = [0]

        def embed_targets():
            # mel front-end on the GPU (qvc_wave_to_mel; raises for configs it does not cover -- there is no CPU path);
            # speaker embeddings once per distinct target of this shard (the reference recomputes them per line, convert.py:64-77)
            front = MelFrontend(d.filter_length, d.n_mel_channels, d.sampling_rate, d.hop_length, d.win_length, d.mel_fmin, d.mel_fmax)
            tgt_row, rows = {}, []
            for i in mine:
                tgt = items[i][2]
                if tgt not in tgt_row:
                    wav = torch.from_numpy(trim(load_wav(tgt, d.sampling_rate), top_db=20)).unsqueeze(0).cuda()
                    tgt_row[tgt] = len(rows)
                    rows.append(net_g.speaker_embed(front(wav)))          # (1, 80, F') -> (1, gin), HIP mel + HIP LSTM
            table = torch.cat(rows, 0)
            g_rows = torch.zeros(len(items), table.shape[1], device=table.device)
            g_rows[torch.as_tensor(mine, device=table.device)] = table[torch.as_tensor([tgt_row[items[i][2]] for i in mine], device=table.device)]
            n_targets[0] = len(rows)
            return g_rows

        stamp = time.strftime('%m-%d_%H-%M', time.localtime())
        out_paths = [os.path.join(outdir, f"{stamp}_{t}.wav" if use_timestamp else f"{t}.wav") for t, _, _ in items]
        mine_set = set(mine)
        src_paths = [_unit_path(src, check=False) if i in mine_set else None for i, (_, src, _) in enumerate(items)]
        # the pipeline's buffers (pinned host memory, workspace, I/O pools: tens of ms) are set up on a helper thread
        # while this one embeds the target speakers
        net_g.engine()                                                    # packed once, before two threads ask for it
        box = {}

        def make_pipe():
            try:
                with torch.cuda.device(dev_index):
                    box["pipe"] = CorpusPipeline(net_g, min(batch, max(len(b) for b in batches)), max(lengths[i] for i in mine),
                                                 d.sampling_rate, io_threads=io_threads, seed=seed, lanes=lanes)
            except Exception as exc:                                      # noqa: BLE001 -- re-raised below
                box["error"] = exc

        dev_index = torch.cuda.current_device()
        th = threading.Thread(target=make_pipe, daemon=True)
        th.start()
        g_rows = embed_targets()
        th.join()
        if "error" in box:
            raise box["error"]
        pipe = box["pipe"]
        try:
            t2 = time.perf_counter()
            pipe.run(batches, src_paths, out_paths, lengths, g_rows)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
        finally:
            pipe.close()
    if timings is not None:
        timings.update(plan_s=t1 - t0, setup_and_embed_s=t2 - t1, pipeline_s=t3 - t2, targets=n_targets[0])
    return pipe.stats


if __name__ == "__main__":
    main()
