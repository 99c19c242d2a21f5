#!/usr/bin/env python3
"""Developer helper (GPU box): per-launch time line of one benchmark step (HIP events, median of 5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quickvc_official_amd as q
from quickvc_official_amd.engine import QvcEngine
from quickvc_official_amd.synth import make_synthetic_state_dict, make_synthetic_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
model = q.SynthesizerTrn(641, 32, **q.DEFAULT_MODEL_CONFIG)
sd = make_synthetic_state_dict(model, 1234)
eng = QvcEngine(model.model_config, sd, dev)
unit, g, noise = make_synthetic_inputs(B, 250, 256, 192, 256)
unit, g, noise = unit.to(dev), g.to(dev), noise.to(dev)
runs = []
for _ in range(6):
    _, recs = eng.infer_batch_timed(unit, g, noise)
    runs.append(recs)
runs = runs[1:]
tot = 0.0
for i, r in enumerate(runs[0]):
    ms = sorted(x[i]["ms"] for x in runs)[len(runs) // 2]
    tot += ms
    tf = r["flops"] / (ms * 1e-3) / 1e12 if ms > 0 else 0
    print(f"{i:3d} {r['name']:34s} {ms * 1e3:8.1f} us {tf:8.1f} TF {r['bytes'] / (ms * 1e-3) / 1e9 if ms > 0 else 0:8.0f} GB/s")
print("sum", tot, "ms")
