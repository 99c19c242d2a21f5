"""Shared helpers for the tests: golden loading, synthetic model/input regeneration."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_case(name):
    """Returns (entry, arrays) for a golden case."""
    entry = manifest()[name]
    arrays = dict(np.load(os.path.join(GOLDEN, entry["file"])))
    return entry, arrays


def subsample(t, limit):
    flat = t.detach().float().reshape(-1)
    stride = max(1, -(-flat.numel() // limit))
    return flat[::stride].cpu().numpy()


def regenerate(entry):
    """(model, state_dict, unit, g, noise) from the seeds recorded in the manifest."""
    import quickvc_official_amd as q
    from quickvc_official_amd.synth import make_synthetic_state_dict, make_synthetic_inputs
    cfg = entry["config"]
    model = q.SynthesizerTrn(641, 32, **cfg)
    sd = make_synthetic_state_dict(model, entry["weights_seed"])
    unit, g, noise = make_synthetic_inputs(entry["batch"], entry["frames"], 256, cfg["inter_channels"],
                                           cfg["gin_channels"], seed0=entry["inputs_seed0"])
    return model, sd, unit, g, noise


def snr_db(ref, out):
    ref = torch.as_tensor(ref).double().flatten()
    out = torch.as_tensor(out).double().flatten()
    err = float(((ref - out) ** 2).sum())
    return float("inf") if err == 0 else 10.0 * float(np.log10(float((ref ** 2).sum()) / err))
