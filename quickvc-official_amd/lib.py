"""ctypes binding of libqvc_hip.so (include/qvc.h).  No torch types cross this boundary:
only raw pointers (``tensor.data_ptr()``), sizes, the POD ``qvc_config`` and a stream handle.

The library is looked up next to this file (built in-tree by build.py).  A missing library
is a hard error: the product has no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict

import torch

QVC_OK = 0
QVC_BF16, QVC_F16 = 0, 1
QVC_DEC_ISTFT, QVC_DEC_MULTIBAND, QVC_DEC_MULTISTREAM = 0, 1, 2
QVC_MAX_UPS = QVC_MAX_RESBLOCKS = 4
QVC_BF16X = 2     # bf16 WaveNet half + f16 generator (include/qvc.h)
DTYPES = {"bf16": QVC_BF16, "f16": QVC_F16, "fp16": QVC_F16, "bf16x": QVC_BF16X}

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libqvc_hip.so")


class QvcConfig(ctypes.Structure):
    _fields_ = [
        ("unit_channels", ctypes.c_int32), ("inter_channels", ctypes.c_int32), ("hidden_channels", ctypes.c_int32),
        ("gin_channels", ctypes.c_int32), ("wn_kernel_size", ctypes.c_int32), ("enc_layers", ctypes.c_int32),
        ("flow_layers", ctypes.c_int32), ("n_flows", ctypes.c_int32), ("upsample_initial_channel", ctypes.c_int32),
        ("n_ups", ctypes.c_int32), ("upsample_rates", ctypes.c_int32 * QVC_MAX_UPS),
        ("upsample_kernel_sizes", ctypes.c_int32 * QVC_MAX_UPS), ("n_resblocks", ctypes.c_int32),
        ("resblock_kernel_sizes", ctypes.c_int32 * QVC_MAX_RESBLOCKS),
        ("resblock_dilations", (ctypes.c_int32 * 3) * QVC_MAX_RESBLOCKS),
        ("n_fft", ctypes.c_int32), ("hop", ctypes.c_int32), ("subbands", ctypes.c_int32), ("decoder", ctypes.c_int32),
        ("fir_taps", ctypes.c_int32), ("operand_dtype", ctypes.c_int32), ("n_mel_channels", ctypes.c_int32), ("spec_channels", ctypes.c_int32),
    ]


class QvcTensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p), ("ndim", ctypes.c_int32),
                ("shape", ctypes.c_int64 * 4)]


class QvcLaunchRecord(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("ms", ctypes.c_float), ("flops", ctypes.c_double),
                ("bytes", ctypes.c_double)]


class QvcError(RuntimeError):
    pass


def declare(lib: ctypes.CDLL, prefix: str = "qvc") -> None:
    """Signatures of include/qvc.h (also used for the test-only emulation library)."""
    P, I, L, V = ctypes.POINTER, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
    cfgp = P(QvcConfig)
    if prefix == "qvc":
        lib.qvc_abi_version.restype = ctypes.c_int
        lib.qvc_status_string.restype = ctypes.c_char_p
        lib.qvc_status_string.argtypes = [ctypes.c_int]
        lib.qvc_device_check.restype = ctypes.c_int
        lib.qvc_debug_set.restype = ctypes.c_int
        lib.qvc_debug_set.argtypes = [ctypes.c_char_p, I]
        lib.qvc_debug_get.restype = ctypes.c_int
        lib.qvc_debug_get.argtypes = [ctypes.c_char_p, P(I)]
        lib.qvc_debug_saturations.restype = ctypes.c_int
        lib.qvc_debug_saturations.argtypes = [P(L), I]
        lib.qvc_blob_bytes.restype = L
        lib.qvc_blob_bytes.argtypes = [cfgp]
        lib.qvc_pack_weights.restype = ctypes.c_int
        lib.qvc_pack_weights.argtypes = [cfgp, P(QvcTensor), I, V, L]
        lib.qvc_plan_info.restype = ctypes.c_int
        lib.qvc_plan_info.argtypes = [cfgp, P(I)]
        lib.qvc_workspace_bytes.restype = L
        lib.qvc_workspace_bytes.argtypes = [cfgp, I, I]
        lib.qvc_infer_batch.restype = ctypes.c_int
        lib.qvc_infer_batch.argtypes = [cfgp, V, V, V, V, V, I, I, V, L, V]
        lib.qvc_infer_batch_ragged.restype = ctypes.c_int
        lib.qvc_infer_batch_ragged.argtypes = [cfgp, V, V, V, V, V, I, I, V, V, L, V]
        lib.qvc_infer_batch_ragged_fm.restype = ctypes.c_int
        lib.qvc_infer_batch_ragged_fm.argtypes = [cfgp, V, V, V, V, V, I, I, V, V, L, V]
        lib.qvc_stream_state_bytes.restype = L
        lib.qvc_stream_state_bytes.argtypes = [cfgp, I, I]
        lib.qvc_stream_workspace_bytes.restype = L
        lib.qvc_stream_workspace_bytes.argtypes = [cfgp, I, I]
        lib.qvc_stream_lag_frames.restype = I
        lib.qvc_stream_lag_frames.argtypes = [cfgp]
        lib.qvc_stream_noise_lag_frames.restype = I
        lib.qvc_stream_noise_lag_frames.argtypes = [cfgp]
        lib.qvc_stream_step.restype = ctypes.c_int
        lib.qvc_stream_step.argtypes = [cfgp, V, V, L, V, V, V, V, I, I, V, V, V, L, V]
        lib.qvc_stream_reset_slot.restype = ctypes.c_int
        lib.qvc_stream_reset_slot.argtypes = [cfgp, V, L, I, I, I, I, V, V, V]
        lib.qvc_aux_create.restype = ctypes.c_int
        lib.qvc_aux_create.argtypes = [P(V)]
        lib.qvc_aux_destroy.restype = ctypes.c_int
        lib.qvc_aux_destroy.argtypes = [V]
        lib.qvc_infer_batch_ex.restype = ctypes.c_int
        lib.qvc_infer_batch_ex.argtypes = [cfgp, V, V, V, V, V, I, I, V, L, V, V]
        lib.qvc_infer_batch_timed.restype = ctypes.c_int
        lib.qvc_infer_batch_timed.argtypes = [cfgp, V, V, V, V, V, I, I, V, L, V, P(QvcLaunchRecord), I, P(I)]
        lib.qvc_enc_p.restype = ctypes.c_int
        lib.qvc_enc_p.argtypes = [cfgp, V, V, V, V, I, I, V, L, V]
        lib.qvc_wn_stack.restype = ctypes.c_int
        lib.qvc_wn_stack.argtypes = [cfgp, V, I, V, V, V, I, I, V, L, V]
        lib.qvc_flow_reverse.restype = ctypes.c_int
        lib.qvc_flow_reverse.argtypes = [cfgp, V, V, V, I, I, V, L, V]
        lib.qvc_dec_trunk.restype = ctypes.c_int
        lib.qvc_dec_trunk.argtypes = [cfgp, V, V, V, V, I, I, V, L, V]
        lib.qvc_istft_synth.restype = ctypes.c_int
        lib.qvc_istft_synth.argtypes = [cfgp, V, V, V, V, I, I, V]
        lib.qvc_conv1d_scratch_bytes.restype = L
        lib.qvc_conv1d_scratch_bytes.argtypes = [I, I, I]
        lib.qvc_conv1d_workspace_bytes.restype = L
        lib.qvc_conv1d_workspace_bytes.argtypes = [I, I, I, I]
        lib.qvc_conv1d.restype = ctypes.c_int
        lib.qvc_conv1d.argtypes = [V, V, V, V, I, I, I, I, I, I, ctypes.c_float, I, V, V, L, V, L, V]
        lib.qvc_spk_blob_bytes.restype = L
        lib.qvc_spk_blob_bytes.argtypes = [cfgp]
        lib.qvc_spk_pack_weights.restype = ctypes.c_int
        lib.qvc_spk_pack_weights.argtypes = [cfgp, P(QvcTensor), I, V, L]
        lib.qvc_spk_workspace_bytes.restype = L
        lib.qvc_spk_workspace_bytes.argtypes = [cfgp, I, I]
        lib.qvc_speaker_embed.restype = ctypes.c_int
        lib.qvc_speaker_embed.argtypes = [cfgp, V, V, V, I, I, V, L, V]
        lib.qvc_encq_blob_bytes.restype = L
        lib.qvc_encq_blob_bytes.argtypes = [cfgp]
        lib.qvc_encq_pack_weights.restype = ctypes.c_int
        lib.qvc_encq_pack_weights.argtypes = [cfgp, P(QvcTensor), I, V, L]
        lib.qvc_enc_q.restype = ctypes.c_int
        lib.qvc_enc_q.argtypes = [cfgp, V, V, V, V, V, I, I, V, L, V]
        lib.qvc_flow_forward.restype = ctypes.c_int
        lib.qvc_flow_forward.argtypes = [cfgp, V, V, V, I, I, V, L, V]
        lib.qvc_mel_table_bytes.restype = L
        lib.qvc_mel_table_bytes.argtypes = [I, I]
        lib.qvc_mel_pack_tables.restype = ctypes.c_int
        lib.qvc_mel_pack_tables.argtypes = [I, I, I, V, V, L]
        lib.qvc_mel_workspace_bytes.restype = L
        lib.qvc_mel_workspace_bytes.argtypes = [I, I, I, I]
        lib.qvc_wave_to_mel.restype = ctypes.c_int
        lib.qvc_wave_to_mel.argtypes = [V, I, I, I, V, V, I, I, V, L, V]


_lib = None


def load_library() -> ctypes.CDLL:
    """Loads libqvc_hip.so or raises -- never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise QvcError(f"{_LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                           "(hipcc --offload-arch=gfx950); the hot path has no CPU fallback")
        lib = ctypes.CDLL(_LIB_PATH)
        declare(lib)
        if lib.qvc_abi_version() != 8:
            raise QvcError("libqvc_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def debug_set(name: str, value: int) -> None:
    """Developer / test switch of the library (include/qvc.h: qvc_debug_set).  The library itself never reads the
    environment; the GPU tests flip launch-shape variants through this call, in-process."""
    lib = load_library()
    check(lib, lib.qvc_debug_set(name.encode(), int(value)), f"qvc_debug_set({name})")


def debug_get(name: str) -> int:
    lib = load_library()
    v = ctypes.c_int32(0)
    check(lib, lib.qvc_debug_get(name.encode(), ctypes.byref(v)), f"qvc_debug_get({name})")
    return int(v.value)


def check(lib, status: int, what: str) -> None:
    if status != QVC_OK:
        raise QvcError(f"{what} failed: {lib.qvc_status_string(int(status)).decode()} ({status})")


def make_config(mc: dict) -> QvcConfig:
    """``SynthesizerTrn.model_config`` (models.py:551-591 hyper-parameters) -> qvc_config."""
    c = QvcConfig()
    c.unit_channels = int(mc.get("unit_channels", 256))
    c.inter_channels, c.hidden_channels = int(mc["inter_channels"]), int(mc["hidden_channels"])
    c.gin_channels = int(mc["gin_channels"])
    c.wn_kernel_size, c.enc_layers, c.flow_layers, c.n_flows = 5, 16, 4, 4          # models.py:582-584
    c.upsample_initial_channel = int(mc["upsample_initial_channel"])
    ups, ks = list(mc["upsample_rates"]), list(mc["upsample_kernel_sizes"])
    rk, rd = list(mc["resblock_kernel_sizes"]), [list(d) for d in mc["resblock_dilation_sizes"]]
    if len(ups) > QVC_MAX_UPS or len(rk) > QVC_MAX_RESBLOCKS or len(ups) != len(ks) or len(rk) != len(rd):
        raise QvcError("unsupported upsample / resblock configuration")
    c.n_ups = len(ups)
    for i, (u, k) in enumerate(zip(ups, ks)):
        c.upsample_rates[i], c.upsample_kernel_sizes[i] = int(u), int(k)
    c.n_resblocks = len(rk)
    for j, (k, d) in enumerate(zip(rk, rd)):
        if len(d) != 3:
            raise QvcError("ResBlock1 needs three dilations")
        c.resblock_kernel_sizes[j] = int(k)
        for q in range(3):
            c.resblock_dilations[j][q] = int(d[q])
    c.n_fft, c.hop = int(mc["gen_istft_n_fft"]), int(mc["gen_istft_hop_size"])
    c.subbands = int(mc["subbands"])
    kind = mc.get("decoder", "multistream")
    c.decoder = {"istft": QVC_DEC_ISTFT, "multiband": QVC_DEC_MULTIBAND, "multistream": QVC_DEC_MULTISTREAM}[kind]
    c.fir_taps = 63
    dt = mc.get("operand_dtype", "f16")
    if dt not in DTYPES:
        raise QvcError(f"operand_dtype must be one of {sorted(DTYPES)}")
    c.operand_dtype = DTYPES[dt]
    c.n_mel_channels = int(mc.get("n_mel_channels", 80))
    c.spec_channels = int(mc.get("spec_channels", 641))
    return c


def pack_weights(lib, cfg: QvcConfig, state_dict: Dict[str, torch.Tensor], which: str = "path") -> torch.Tensor:
    """state_dict (reference keys) -> packed host blob (uint8 tensor, 256-byte aligned storage).

    ``which``: "path" = enc_p / flow / dec (qvc_pack_weights), "spk" = the speaker encoder (qvc_spk_pack_weights).
    """
    bytes_fn, pack_fn = {"path": (lib.qvc_blob_bytes, lib.qvc_pack_weights),
                         "spk": (lib.qvc_spk_blob_bytes, lib.qvc_spk_pack_weights),
                         "encq": (lib.qvc_encq_blob_bytes, lib.qvc_encq_pack_weights)}[which]
    n = int(bytes_fn(ctypes.byref(cfg)))
    if n < 0:
        check(lib, n, bytes_fn.__name__)
    keep = []   # keep fp32 contiguous CPU copies alive during the call
    arr = (QvcTensor * len(state_dict))()
    i = 0
    for name, t in state_dict.items():
        if not torch.is_tensor(t) or not t.is_floating_point() or t.dim() > 4:
            continue
        tc = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
        keep.append(tc)
        arr[i].name = name.encode()
        arr[i].data = tc.data_ptr()
        arr[i].ndim = tc.dim()
        for d in range(tc.dim()):
            arr[i].shape[d] = tc.shape[d]
        i += 1
    raw = torch.empty(n + 256, dtype=torch.uint8)
    shift = (-raw.data_ptr()) % 256
    blob = raw[shift:shift + n]
    check(lib, pack_fn(ctypes.byref(cfg), arr, i, blob.data_ptr(), n), pack_fn.__name__)
    return blob
