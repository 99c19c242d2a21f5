"""JSON config -> nested attribute object (the reference's ``HParams`` surface).

Mirrors utils.py:105-143 of the reference: ``get_hparams_from_file(path)`` returns an
object whose nested dict sections are attribute-accessible *and* behave like a
mapping (``**hps.model`` is how convert.py:35 builds the model), so configs written
for the reference load unchanged.  Extra keys are kept; nothing is validated here.
"""
from __future__ import annotations

import json
from typing import Any, Iterator


class HParams:
    """Recursive attribute/mapping view over a JSON object (utils.py:114-143)."""

    def __init__(self, **entries: Any) -> None:
        for key, value in entries.items():
            self.__dict__[key] = HParams(**value) if isinstance(value, dict) else value

    # mapping protocol -- needed for ``**hps.model``
    def keys(self):
        return self.__dict__.keys()

    def items(self):
        return self.__dict__.items()

    def values(self):
        return self.__dict__.values()

    def __iter__(self) -> Iterator[str]:
        return iter(self.__dict__)

    def __len__(self) -> int:
        return len(self.__dict__)

    def __getitem__(self, key: str) -> Any:
        return self.__dict__[key]

    def __setitem__(self, key: str, value: Any) -> None:
        self.__dict__[key] = value

    def __contains__(self, key: str) -> bool:
        return key in self.__dict__

    def __repr__(self) -> str:
        return repr(self.__dict__)

    def to_dict(self) -> dict:
        return {k: (v.to_dict() if isinstance(v, HParams) else v) for k, v in self.__dict__.items()}


def get_hparams_from_file(config_path: str) -> HParams:
    """utils.py:105-111."""
    with open(config_path, "r") as f:
        return HParams(**json.loads(f.read()))


# The shipped model section (logs/quickvc/config.json:42-68 of the reference), used by
# bench.py / smoke when no config file is given.  Unused keys are kept on purpose: the
# constructor must swallow them like the reference does (models.py:568,573).
DEFAULT_MODEL_CONFIG = {
    "ms_istft_vits": True, "mb_istft_vits": False, "istft_vits": False, "subbands": 4,
    "gen_istft_n_fft": 16, "gen_istft_hop_size": 4,
    "inter_channels": 192, "hidden_channels": 192, "filter_channels": 768,
    "n_heads": 2, "n_layers": 6, "kernel_size": 3, "p_dropout": 0.1, "resblock": "1",
    "resblock_kernel_sizes": [3, 7, 11], "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    "upsample_rates": [5, 4], "upsample_initial_channel": 512, "upsample_kernel_sizes": [16, 16],
    "n_layers_q": 3, "use_spectral_norm": False, "gin_channels": 256,
    "use_sdp": False, "ssl_dim": 1024, "use_spk": False,
}

DEFAULT_DATA_CONFIG = {
    "max_wav_value": 32768.0, "sampling_rate": 16000, "filter_length": 1280, "hop_length": 320,
    "win_length": 1280, "n_mel_channels": 80, "mel_fmin": 0.0, "mel_fmax": None,
}

# A structurally identical but tiny model (same kernel sizes / strides / dilations / n_fft),
# used for golden vectors whose weights are small enough to commit (SURVEY 8c "mini").
MINI_MODEL_CONFIG = dict(DEFAULT_MODEL_CONFIG, inter_channels=64, hidden_channels=64,
                         upsample_initial_channel=128, gin_channels=64)
# Channel counts that are not multiples of the MFMA K-step (32) nor of 16: exercises every
# zero-padding path of the weight packer and the masked stores of the kernels.
ODD_MODEL_CONFIG = dict(DEFAULT_MODEL_CONFIG, inter_channels=48, hidden_channels=40,
                        upsample_initial_channel=96, gin_channels=24)
