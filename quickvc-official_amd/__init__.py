"""MI355X-native QuickVC inference hot path (drop-in for tarepan/QuickVC-official's
``SynthesizerTrn.infer`` / ``convert.py``).

The directory name carries a hyphen (it is fixed by the project layout), so it is
imported through the repo-root shim module ``quickvc_official_amd`` which registers
this directory as the package ``quickvc_official_amd``.

Importing the package never touches the GPU and never needs the HIP library; the
library is loaded (and its absence reported loudly) the first time the hot path runs.
"""
from .config import HParams, get_hparams_from_file, DEFAULT_MODEL_CONFIG, DEFAULT_DATA_CONFIG, MINI_MODEL_CONFIG, ODD_MODEL_CONFIG
from .model import SynthesizerTrn, SpeakerEncoder
from .checkpoint import load_checkpoint, save_checkpoint, latest_checkpoint_path

__all__ = [
    "HParams", "get_hparams_from_file", "DEFAULT_MODEL_CONFIG", "DEFAULT_DATA_CONFIG", "MINI_MODEL_CONFIG",
    "ODD_MODEL_CONFIG", "SynthesizerTrn", "SpeakerEncoder", "load_checkpoint", "save_checkpoint",
    "latest_checkpoint_path",
]
