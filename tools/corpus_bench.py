#!/usr/bin/env python3
"""tools/corpus_bench.py -- BASELINE.json configs[3] on one GPU: a synthetic corpus of unit files converted end to end.

Writes N synthetic utterances (unit .npy files of random lengths, the reference's on-disk format, dataset/encode.py:38)
and a few target wavs into a scratch directory, then times
  * end-to-end: list -> plan (headers) -> target speakers embedded while the pipeline's buffers are set up ->
    CorpusPipeline (load | upload, convert, download | write wavs);
  * kernel-only: the same ragged batches with inputs resident in HBM (no file I/O, no PCIe), replayed back to back.
Prints ONE JSON line.  The scratch directory is removed afterwards.

    python tools/corpus_bench.py [--n 2048] [--targets 8] [--min-frames 40] [--max-frames 400] [--batch 32] [--dtype bf16x]
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--targets", type=int, default=8)
    ap.add_argument("--min-frames", type=int, default=40)
    ap.add_argument("--max-frames", type=int, default=400)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16x", choices=["f16", "bf16", "bf16x"])
    ap.add_argument("--io-threads", type=int, default=8)
    ap.add_argument("--lanes", type=int, default=2, help="batches converted at the same time in the pipeline")
    ap.add_argument("--scratch", default=None, help="parent of the scratch directory (default: the system temp dir)")
    ap.add_argument("--repeat", type=int, default=2, help="end-to-end passes; the last one is reported (the first warms the page cache)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from scipy.io import wavfile
    import quickvc_official_amd as q
    from quickvc_official_amd import convert as cli
    from quickvc_official_amd.config import HParams
    from quickvc_official_amd.synth import make_synthetic_state_dict

    torch.cuda.set_device(0)
    cfg = dict(q.DEFAULT_MODEL_CONFIG)
    net_g = q.SynthesizerTrn(641, 32, **cfg, operand_dtype=args.dtype)
    net_g.load_state_dict(make_synthetic_state_dict(net_g, 1234))
    net_g = net_g.cuda().eval()
    d = HParams(**q.DEFAULT_DATA_CONFIG)

    td = tempfile.mkdtemp(prefix="qvc_corpus_", dir=args.scratch)
    try:
        rng = np.random.RandomState(5)
        sr = d.sampling_rate
        t = np.arange(int(2.0 * sr)) / sr
        for k in range(args.targets):
            wav = 0.3 * np.sin(2 * np.pi * (110.0 + 17.0 * k) * t) + 0.02 * rng.randn(len(t))
            wavfile.write(os.path.join(td, f"tgt{k}.wav"), sr, (np.clip(wav, -1, 1) * 32767).astype(np.int16))
        lens = rng.randint(args.min_frames, args.max_frames + 1, size=args.n)
        t0 = time.perf_counter()
        for i, n in enumerate(lens):
            np.save(os.path.join(td, f"u{i:05d}.npy"), rng.randn(int(n), 256).astype(np.float32))
        gen_s = time.perf_counter() - t0
        items = [(f"o{i:05d}", os.path.join(td, f"u{i:05d}.npy"), os.path.join(td, f"tgt{i % args.targets}.wav")) for i in range(args.n)]
        outdir = os.path.join(td, "out")
        os.makedirs(outdir)

        # ---- end to end (everything after the model is on the GPU)
        e2e = []
        for rep in range(max(1, args.repeat)):
            timings = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            stats = cli.convert_items(net_g, d, items, outdir, 0, 1, args.batch, seed=7, io_threads=args.io_threads, timings=timings, lanes=args.lanes)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            e2e.append(dict(wall_s=wall, **timings, stages={k: round(v, 4) for k, v in stats.items() if k.endswith("_s")}))
        written = len(os.listdir(outdir))
        sizes_ok = all(os.path.getsize(os.path.join(outdir, f"o{i:05d}.wav")) == 58 + int(lens[i]) * 320 * 4 for i in range(0, args.n, 97))

        # ---- kernel only: the same batches, inputs already in HBM, back to back on one stream
        lengths, mine, batches = cli.rank_plan(items, 0, 1, args.batch)
        eng = net_g.engine()
        dev = eng.device
        tmax_all = max(lengths)
        B = max(len(b) for b in batches)
        unit = torch.randn(B, tmax_all, 256, device=dev)
        g = torch.nn.functional.normalize(torch.rand(B, cfg["gin_channels"], device=dev), dim=1)
        out = torch.empty(B * tmax_all * 320, device=dev)
        ws = eng.alloc_workspace(B, tmax_all)
        prepared = []
        for idxs in batches:
            n, tm = len(idxs), max(lengths[i] for i in idxs)
            prepared.append((n, tm, torch.tensor([lengths[i] for i in idxs], dtype=torch.int32, device=dev),
                             torch.randn(n, cfg["inter_channels"], tm, device=dev)))

        def kernel_pass():
            for n, tm, lens_dev, noise in prepared:
                eng.infer_batch_ragged(unit.view(-1)[:n * tm * 256].view(n, tm, 256), g[:n], noise, lens_dev,
                                       out=out[:n * tm * 320].view(n, 1, tm * 320), ws=ws, unit_fm=True)
        kernel_pass()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            kernel_pass()
        torch.cuda.synchronize()
        kernel_s = (time.perf_counter() - t0) / 3

        samples = int(lens.sum()) * 320
        last = e2e[-1]
        res = {
            "workload": f"{args.n} synthetic utterances, {args.min_frames}-{args.max_frames} unit frames (uniform), {args.targets} targets, "
                        f"shipped config, {args.dtype}, batches of <= {args.batch} (BASELINE.json configs[3], one GPU)",
            "utterances": int(stats["utterances"]), "batches": int(stats["batches"]), "audio_seconds": samples / sr,
            "files_written": written, "file_sizes_ok": bool(sizes_ok),
            "end_to_end": {"wall_s": last["wall_s"], "samples_per_s": samples / last["wall_s"], "utterances_per_s": args.n / last["wall_s"],
                           "plan_s": last["plan_s"], "setup_and_embed_s": last["setup_and_embed_s"], "pipeline_s": last["pipeline_s"],
                           "pipeline_samples_per_s": samples / last["pipeline_s"], "passes": e2e},
            "kernel_only": {"wall_s": kernel_s, "samples_per_s": samples / kernel_s},
            "pipeline_vs_kernel_only": kernel_s / last["pipeline_s"], "end_to_end_vs_kernel_only": kernel_s / last["wall_s"],
            "lanes": args.lanes,
            "io": {"input_MB": float(lens.sum()) * 1024 / 1e6, "output_MB": samples * 4 / 1e6, "io_threads": args.io_threads,
                   "scratch": os.path.dirname(td), "corpus_generation_s": gen_s},
        }
        print(json.dumps(res))
    finally:
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
