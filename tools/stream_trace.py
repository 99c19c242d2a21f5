"""Per-kernel cost of ONE incremental streaming step (developer tool): run under
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stream -o s --output-format csv -- python3 tools/stream_trace.py [streams] [hop]
and read the kernel stats: only the incremental converter runs here (tools/stream_bench.py also times the offline and
windowed paths)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickvc_official_amd as q  # noqa: E402
from quickvc_official_amd.streaming import StreamConverter  # noqa: E402
from quickvc_official_amd.synth import make_synthetic_state_dict  # noqa: E402

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hop = int(sys.argv[2]) if len(sys.argv) > 2 else 16
model = q.SynthesizerTrn(641, 32, **q.DEFAULT_MODEL_CONFIG)
model.load_state_dict(make_synthetic_state_dict(model, 1234))
model = model.cuda().eval()
conv = StreamConverter(model, streams, hop_frames=hop)
with torch.cuda.stream(conv._stream):
    conv._unit.normal_()
    conv._noise.normal_()
    conv._g.copy_(torch.nn.functional.normalize(torch.rand_like(conv._g), dim=1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(12):
        conv._graph.replay()
    e0.record(conv._stream)
    for _ in range(50):
        conv._graph.replay()
    e1.record(conv._stream)
conv._stream.synchronize()
print(f"{streams} streams, hop {hop}: {e0.elapsed_time(e1) / 50:.4f} ms per step")
