"""Streaming at a small hop: one StreamConverter of S streams against two of S/2 streams replayed on two HIP streams
(developer probe).  At hop 16 a step's WaveNet launches have only S tiles -- a quarter of the CUs at S = 64 -- and cost
their weight stream however few frames they see, so two half-width converters can run side by side.
usage: python tools/stream_lanes_probe.py [streams] [hop]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickvc_official_amd as q  # noqa: E402
from quickvc_official_amd.streaming import StreamConverter  # noqa: E402
from quickvc_official_amd.synth import make_synthetic_state_dict  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hop = int(sys.argv[2]) if len(sys.argv) > 2 else 16
model = q.SynthesizerTrn(641, 32, **q.DEFAULT_MODEL_CONFIG)
model.load_state_dict(make_synthetic_state_dict(model, 1234))
model = model.cuda().eval()


def prime(conv):
    with torch.cuda.stream(conv._stream):
        conv._unit.normal_()
        conv._noise.normal_()
        conv._g.copy_(torch.nn.functional.normalize(torch.rand_like(conv._g), dim=1))
        for _ in range(12):
            conv._graph.replay()
    conv._stream.synchronize()


def run(convs, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for c in convs:
            with torch.cuda.stream(c._stream):
                c._graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


res = {"streams": S, "hop": hop}
for parts in (1, 2, 4):
    convs = [StreamConverter(model, S // parts, hop_frames=hop) for _ in range(parts)]
    for c in convs:
        prime(c)
    run(convs, 10)
    res[f"{parts}_converters_ms_per_step_of_all_streams"] = run(convs, 50)
    del convs
    torch.cuda.empty_cache()
print(json.dumps(res))
