"""Developer helper (GPU box): duration of each of N consecutive graph replays of the benchmark step after an idle
period -- shows how many steps the device needs to reach its steady clock.  usage: python tools/warmup_probe.py [N]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickvc_official_amd as q  # noqa: E402
from quickvc_official_amd.engine import QvcEngine  # noqa: E402
from quickvc_official_amd.synth import make_synthetic_inputs, make_synthetic_state_dict  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
model = q.SynthesizerTrn(641, 32, **q.DEFAULT_MODEL_CONFIG)
eng = QvcEngine(dict(model.model_config, operand_dtype="bf16x"), make_synthetic_state_dict(model, 1234), dev)
unit, g, noise = (t.to(dev) for t in make_synthetic_inputs(32, 250, 256, 192, 256))
out = torch.empty(32, 1, 80000, device=dev)
s = torch.cuda.Stream(dev)
ws = eng.alloc_workspace(32, 250)
with torch.cuda.stream(s):
    eng.infer_batch(unit, g, noise, out, ws=ws)
    s.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        eng.infer_batch(unit, g, noise, out, ws=ws)
    for idle in (0.0, 0.5):
        time.sleep(idle)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
        ev[0].record(s)
        for i in range(N):
            graph.replay()
            ev[i + 1].record(s)
        s.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
        print(f"after {idle:.1f} s idle:", " ".join(f"{x:.3f}" for x in ms[:12]), "...",
              "mean[20:40] %.3f  mean[60:] %.3f" % (sum(ms[20:40]) / 20, sum(ms[60:]) / max(1, len(ms[60:]))))
