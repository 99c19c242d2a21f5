#!/usr/bin/env python3
"""tools/snr_configs.py -- waveform SNR against the fp32 CPU oracle per configuration and operand mode.

For the shipped configuration and the three other configurations of tests/test_gpu_parity.py::test_other_configurations_vs_oracle
(narrow / x4x4 / multiband): min and mean SNR over the utterances for f16, bf16x and bf16.  Prints one JSON line.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import torch
    import quickvc_official_amd as q
    from quickvc_official_amd.engine import QvcEngine
    from quickvc_official_amd.synth import make_synthetic_state_dict, make_synthetic_inputs
    import qvc_oracle as oracle

    configs = {
        "shipped": {},
        "narrow": dict(inter_channels=128, hidden_channels=96, upsample_initial_channel=256, gin_channels=128),
        "x4x4": dict(upsample_rates=[4, 4], upsample_kernel_sizes=[15, 16], resblock_kernel_sizes=[3, 5, 7],
                     resblock_dilation_sizes=[[1, 2, 3], [1, 2, 3], [1, 2, 3]], upsample_initial_channel=384),
        "multiband": dict(ms_istft_vits=False, mb_istft_vits=True, upsample_initial_channel=256, inter_channels=96, hidden_channels=128),
    }
    dev = torch.device("cuda:0")
    res = {}
    for name, over in configs.items():
        cfg = dict(q.DEFAULT_MODEL_CONFIG, **over)
        model = q.SynthesizerTrn(641, 32, **cfg)
        res[name] = {}
        for seed_w, (B, T) in ((311, (3, 41)), (1234, (4, 120))):
            sd = make_synthetic_state_dict(model, seed_w)
            unit, g, noise = make_synthetic_inputs(B, T, 256, cfg["inter_channels"], cfg["gin_channels"], seed0=71)
            ref = oracle.infer_from_g(sd, cfg, unit, g.unsqueeze(-1), noise)
            for dt in ("f16", "bf16x", "bf16"):
                eng = QvcEngine(dict(model.model_config, operand_dtype=dt), sd, dev)
                out = eng.infer_batch(unit.to(dev), g.to(dev), noise.to(dev))
                torch.cuda.synchronize()
                snrs = [oracle.snr_db(ref[b], out[b].cpu()) for b in range(B)]
                res[name][f"weights{seed_w}_T{T}_{dt}"] = {"min": min(snrs), "mean": sum(snrs) / len(snrs)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
