"""Experiment: does staggering sub-batches (the WaveNet part of one sub-batch -- L2-bound -- under the decoder of the
previous one -- MFMA/power-bound) beat one full-batch pass?  Uses the stage entry points on several streams, captured
in one hipGraph.  usage: python tools/pipeline_probe.py [parts ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickvc_official_amd as q  # noqa: E402
from quickvc_official_amd.engine import QvcEngine  # noqa: E402
from quickvc_official_amd.synth import make_synthetic_inputs, make_synthetic_state_dict  # noqa: E402


def run(parts, B=32, T=250, staggered=True, steps=20):
    dev = torch.device("cuda:0")
    model = q.SynthesizerTrn(641, 32, **q.DEFAULT_MODEL_CONFIG)
    sd = make_synthetic_state_dict(model, 1234)
    engines = [QvcEngine(model.model_config, sd, dev) for _ in range(parts)]
    unit, g, noise = (t.to(dev) for t in make_synthetic_inputs(B, T, 256, 192, 256, seed0=0))
    n = B // parts
    sl = [slice(p * n, (p + 1) * n) for p in range(parts)]
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    main = torch.cuda.Stream(dev)
    outs = [None] * parts

    def body():
        start = torch.cuda.Event(); start.record(main)
        flow_done = []
        for p in range(parts):
            s = streams[p]
            s.wait_event(start)
            with torch.cuda.stream(s):
                if staggered and p > 0:
                    s.wait_event(flow_done[p - 1])
                z = engines[p].enc_p(unit[sl[p]], noise[sl[p]])
                z = engines[p].flow_reverse(z, g[sl[p]])
                e = torch.cuda.Event(); e.record(s); flow_done.append(e)
                post = engines[p].dec_trunk(z, g[sl[p]])
                outs[p] = engines[p].istft_synth(post)
        for p in range(parts):
            e = torch.cuda.Event(); e.record(streams[p]); main.wait_event(e)

    with torch.cuda.stream(main):
        body(); main.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=main):
            body()
        for _ in range(3):
            graph.replay()
        main.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main)
        for _ in range(steps):
            graph.replay()
        b.record(main)
        main.synchronize()
    ms = a.elapsed_time(b) / steps
    ref = engines[0].infer_batch(unit[sl[0]], g[sl[0]], noise[sl[0]])
    ok = torch.equal(ref, outs[0])
    print(f"parts={parts} staggered={staggered}: {ms:.3f} ms per {B} utterances ({B * T * 320 / ms / 1e6:.1f} M samples/s) first part identical={ok}", flush=True)


if __name__ == "__main__":
    for parts in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
        run(parts, staggered=True)
        if parts > 1:
            run(parts, staggered=False)
