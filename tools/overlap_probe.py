#!/usr/bin/env python3
"""Does the chip have room for a second batch in flight?  (developer probe, not part of the product)

Several hipGraphs of the same batch-32 step, each with its own inputs, workspace, output and stream, replayed round
robin with 1, 2, ... of them in flight, so that the kernels of step i+1 may fill whatever step i leaves idle (launch
tails, the gaps between dependent launches, CUs the WaveNet launches leave without a workgroup).  Prints one JSON line.

    python tools/overlap_probe.py [--steps 40] [--lanes 4]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=250)
    ap.add_argument("--lanes", type=int, default=4, help="graphs (batches) built; 1 .. lanes of them are kept in flight")
    args = ap.parse_args()
    import torch
    import quickvc_official_amd as q
    from quickvc_official_amd.engine import QvcEngine
    from quickvc_official_amd.synth import make_synthetic_state_dict, make_synthetic_inputs

    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cfg = dict(q.DEFAULT_MODEL_CONFIG)
    model = q.SynthesizerTrn(641, 32, **cfg, operand_dtype="f16")
    engine = QvcEngine(model.model_config, make_synthetic_state_dict(model, 1234), device)
    B, F = args.batch, args.frames
    lanes = []
    for k in range(args.lanes):
        unit, g, noise = make_synthetic_inputs(B, F, 256, cfg["inter_channels"], cfg["gin_channels"], seed0=k * B)
        unit, g, noise = unit.to(device), g.to(device), noise.to(device)
        out = torch.empty(B, 1, F * engine.samples_per_frame, dtype=torch.float32, device=device)
        ws = engine.alloc_workspace(B, F)
        stream = torch.cuda.Stream(device)
        with torch.cuda.stream(stream):
            engine.infer_batch(unit, g, noise, out, ws=ws)
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                engine.infer_batch(unit, g, noise, out, ws=ws)
        lanes.append({"stream": stream, "graph": graph, "out": out, "keep": (unit, g, noise, ws)})

    def run(in_flight: int, steps: int, offset_ms: float = 0.0) -> float:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if offset_ms > 0 and in_flight > 1:                  # hold the second lane back: does the relative phase of the lanes matter?
            with torch.cuda.stream(lanes[1]["stream"]):
                torch.cuda._sleep(int(offset_ms * 1e-3 * 100e6))   # s_memrealtime ticks at 100 MHz on this part
        for i in range(steps):
            lane = lanes[i % in_flight]
            with torch.cuda.stream(lane["stream"]):
                lane["graph"].replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    ref = [lane["out"].clone() for lane in lanes]
    res = {}
    for rnd in range(2):
        for n in range(1, args.lanes + 1):
            run(n, 6)
            res[f"in_flight_{n}_ms_per_step_run{rnd}"] = run(n, args.steps)
    if args.lanes >= 2:
        for off in (0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0):
            run(2, 6)
            res[f"in_flight_2_offset_{off}ms"] = run(2, 5 * args.steps, off)
    res["outputs_unchanged"] = all(torch.equal(r, lane["out"]) for r, lane in zip(ref, lanes))
    res["steps"] = args.steps
    print(json.dumps(res))


if __name__ == "__main__":
    main()
