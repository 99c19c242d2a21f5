"""BASELINE.json configs[4]: chunked streaming conversion, 320-frame hop, 64 concurrent streams, hipGraph replay.
Reports the time of one chunk step (all streams advance by one hop) and the useful-sample throughput.
usage: python tools/stream_bench.py [streams] [hop_frames] [utterance_frames]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickvc_official_amd as q  # noqa: E402
from quickvc_official_amd.streaming import ChunkedConverter  # noqa: E402
from quickvc_official_amd.synth import make_synthetic_inputs, make_synthetic_state_dict  # noqa: E402


def main():
    streams = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    hop = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 1600          # 32 s per stream
    model = q.SynthesizerTrn(641, 32, **q.DEFAULT_MODEL_CONFIG)
    model.load_state_dict(make_synthetic_state_dict(model, 1234))
    model = model.cuda().eval()
    conv = ChunkedConverter(model, streams, hop_frames=hop)
    unit, g, noise = make_synthetic_inputs(streams, frames, 256, 192, 256, seed0=500)
    unit, g, noise = unit.cuda(), g.cuda(), noise.cuda()
    conv.convert(unit, g, noise)                                      # warm-up
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        out = conv.convert(unit, g, noise)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    n_chunks = len(list(conv.windows(frames)))
    # graph replay alone (no window copies): the device time of one chunk step
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(conv._stream):
        e0.record(conv._stream)
        for _ in range(20):
            conv._graph.replay()
        e1.record(conv._stream)
    conv._stream.synchronize()
    spf = model.samples_per_frame
    print(json.dumps({
        "workload": f"chunked streaming, {streams} streams x {frames} frames, hop {hop} + 2x{conv.context} context frames",
        "chunk_steps": n_chunks, "ms_per_chunk_step_incl_copies": wall / n_chunks * 1e3,
        "ms_per_graph_replay": e0.elapsed_time(e1) / 20,
        "useful_samples_per_s": streams * frames * spf / wall,
        "rtf": wall / (streams * frames * spf / 16000.0),
        "audio_seconds_per_chunk_per_stream": hop * spf / 16000.0,
        "window_overhead": conv.window / hop, "out_shape": list(out.shape)}))


if __name__ == "__main__":
    main()
