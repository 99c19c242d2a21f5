"""BASELINE.json configs[4]: streaming conversion, 64 concurrent streams, hipGraph replay per step.
Times one step of (a) StreamConverter -- incremental, segment rings (qvc_stream_step) -- and (b) ChunkedConverter --
exact windows over the whole path -- against (c) the offline whole-batch conversion of the same number of frames.
usage: python tools/stream_bench.py [streams] [hop_frames ...]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickvc_official_amd as q  # noqa: E402
from quickvc_official_amd.streaming import ChunkedConverter, StreamConverter  # noqa: E402
from quickvc_official_amd.synth import make_synthetic_inputs, make_synthetic_state_dict  # noqa: E402


def timed_replays(conv, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(conv._stream):
        # random inputs (zero operands let the chip clock ~20 % higher: DESIGN.md), and enough warm-up steps for the
        # rings of an incremental converter to fill with real data
        conv._unit.normal_()
        conv._noise.normal_()
        conv._g.copy_(torch.nn.functional.normalize(torch.rand_like(conv._g), dim=1))
        for _ in range(12):
            conv._graph.replay()
        e0.record(conv._stream)
        for _ in range(n):
            conv._graph.replay()
        e1.record(conv._stream)
    conv._stream.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    streams = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    hops = [int(x) for x in sys.argv[2:]] or [320, 16]
    model = q.SynthesizerTrn(641, 32, **q.DEFAULT_MODEL_CONFIG)
    model.load_state_dict(make_synthetic_state_dict(model, 1234))
    model = model.cuda().eval()
    spf = model.samples_per_frame
    # offline cost per frame: the whole batch of `streams` x 320 frames in one call (graph replay)
    eng = model.engine()
    unit, g, noise = make_synthetic_inputs(streams, 320, 256, 192, 256, seed0=500)
    unit, g, noise = unit.cuda(), g.cuda(), noise.cuda()
    out = torch.empty(streams, 1, 320 * spf, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ws = eng.alloc_workspace(streams, 320)
        eng.infer_batch(unit, g, noise, out, ws=ws)
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            eng.infer_batch(unit, g, noise, out, ws=ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10):
            graph.replay()
        e1.record(s)
    s.synchronize()
    offline_us_per_frame = e0.elapsed_time(e1) / 10 * 1e3 / (streams * 320)
    res = {"streams": streams, "offline_us_per_frame": offline_us_per_frame, "offline_ms_per_320_frames": e0.elapsed_time(e1) / 10, "hops": {}}
    for hop in hops:
        inc = StreamConverter(model, streams, hop_frames=hop)
        ms_inc = timed_replays(inc)
        row = {"incremental_ms_per_step": ms_inc, "incremental_us_per_frame": ms_inc * 1e3 / (streams * hop),
               "incremental_vs_offline": ms_inc * 1e3 / (streams * hop) / offline_us_per_frame, "lag_frames": inc.lag,
               "incremental_useful_samples_per_s": streams * hop * spf / (ms_inc * 1e-3),
               "audio_seconds_per_step_per_stream": hop * spf / 16000.0}
        del inc
        try:
            ch = ChunkedConverter(model, streams, hop_frames=hop)
            ms_ch = timed_replays(ch)
            row.update({"windowed_ms_per_step": ms_ch, "windowed_vs_offline": ms_ch * 1e3 / (streams * hop) / offline_us_per_frame,
                        "windowed_window_frames": ch.window})
            del ch
        except Exception as exc:                      # e.g. out of memory at a tiny hop x many streams
            row["windowed_error"] = str(exc)[:120]
        res["hops"][str(hop)] = row
        torch.cuda.empty_cache()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
