#!/usr/bin/env python3
"""Golden vectors for the posterior direction: enc_q(spec | g) and the forward flow, recorded from
the REFERENCE itself (models.py:617-618).  Build-container only, like make_golden.py (whose import shims it uses):

    python tests/golden/make_golden_q.py

Writes mini_q.npz / odd_q.npz and adds their entries to manifest.json; the other fixtures are left untouched.
Inputs and weights are regenerated from seeds (quickvc-official_amd/synth.py); only reference OUTPUTS are stored.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (install_shims, subsample; sets sys.path for the repo and the oracle)


def main():
    mg.install_shims()
    sys.path.insert(0, mg.REFERENCE)
    import models as ref_models                                       # the reference, unmodified
    import quickvc_official_amd as q
    from quickvc_official_amd.synth import make_synthetic_state_dict, make_synthetic_posterior_inputs
    import qvc_oracle as oracle

    torch.manual_seed(0)
    torch.set_num_threads(8)
    manifest = json.load(open(os.path.join(HERE, "manifest.json")))
    for name, cfg, seed, batch, frames in (("mini_q", q.MINI_MODEL_CONFIG, 1234, 2, 23), ("odd_q", q.ODD_MODEL_CONFIG, 4321, 2, 21)):
        ours = q.SynthesizerTrn(641, 32, **cfg)
        sd = make_synthetic_state_dict(ours, seed)
        net = ref_models.SynthesizerTrn(641, 32, **cfg)
        net.load_state_dict(sd)
        net.eval()
        spec, g, noise = make_synthetic_posterior_inputs(batch, frames, 641, cfg["inter_channels"], cfg["gin_channels"], seed0=50)
        orig = torch.randn_like
        torch.randn_like = lambda t, *a, **k: noise.to(t.dtype)      # models.py:94
        try:
            with torch.no_grad():
                z, m_q, logs_q = net.enc_q(spec, cond=g.unsqueeze(-1))           # models.py:617
                z_p = net.flow(z, g=g.unsqueeze(-1))                              # models.py:618
                back = net.flow(z_p, g=g.unsqueeze(-1), reverse=True)
        finally:
            torch.randn_like = orig
        assert (back - z).abs().max() < 1e-4                          # the flow is invertible
        taps = {"enc_q.m": m_q, "enc_q.logs": logs_q, "enc_q.z": z, "flow.z_p": z_p}
        otaps = {}
        oracle.posterior_encode(sd, cfg, spec, g.unsqueeze(-1), noise, otaps)
        worst = 0.0
        for k, v in taps.items():
            diff = (otaps[k] - v).abs().max().item()
            scale = v.abs().max().item()
            worst = max(worst, diff / max(scale, 1e-9))
            assert diff <= 2e-5 * max(1.0, scale), f"{name}:{k}: oracle differs from the reference by {diff}"
        print(f"== {name}: oracle == reference on {len(taps)} taps (worst rel-to-max diff {worst:.2e})")
        arrays = {}
        for k, v in taps.items():
            arrays[k] = v.numpy()
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **arrays)
        manifest[name] = {"config": cfg, "weights_seed": seed, "inputs_seed0": 50, "batch": batch, "frames": frames,
                          "file": f"{name}.npz", "taps": sorted(taps.keys()), "kind": "posterior"}
        print(f"   wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
