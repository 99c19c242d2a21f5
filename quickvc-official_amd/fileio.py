"""ctypes binding of libqvc_io.so (include/qvc_io.h): batch file I/O for corpus-scale conversion.

Unit files in (the reference's ``.npy`` format, dataset/encode.py:33-38), float32 wav files out (byte-identical to
``scipy.io.wavfile.write`` as convert.py:84-86 calls it), each call spread over a pool of native worker threads.  The
calls release the GIL, so a Python loader / writer thread per direction keeps the disk busy while the main thread
feeds the GPU.  Host-only code: nothing here touches the HIP library.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Sequence

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libqvc_io.so")
_lib = None

_ERRORS = {-1: "bad argument", -2: "a file could not be opened / created", -3: "not a little-endian float32 C-ordered 2-D .npy",
           -4: "wrong column count, or more frames than the slot holds", -5: "short read / write"}


class QvcIoError(RuntimeError):
    pass


def load_library() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise QvcIoError(f"{_LIB_PATH} is missing: build it with `python __graft_entry__.py`")
        lib = ctypes.CDLL(_LIB_PATH)
        P, I, L, V = ctypes.POINTER, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
        lib.qvc_io_pool_create.restype = ctypes.c_int
        lib.qvc_io_pool_create.argtypes = [I, P(V)]
        lib.qvc_io_pool_destroy.restype = ctypes.c_int
        lib.qvc_io_pool_destroy.argtypes = [V]
        lib.qvc_io_npy_shape.restype = ctypes.c_int
        lib.qvc_io_npy_shape.argtypes = [ctypes.c_char_p, P(I), P(I)]
        lib.qvc_io_npy_shapes.restype = ctypes.c_int
        lib.qvc_io_npy_shapes.argtypes = [V, P(ctypes.c_char_p), I, P(I), P(I)]
        lib.qvc_io_load_units.restype = ctypes.c_int
        lib.qvc_io_load_units.argtypes = [V, P(ctypes.c_char_p), I, V, I, I, V]
        lib.qvc_io_write_wavs.restype = ctypes.c_int
        lib.qvc_io_write_wavs.argtypes = [V, P(ctypes.c_char_p), I, V, L, V, I]
        _lib = lib
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise QvcIoError(f"{what} failed: {_ERRORS.get(rc, 'unknown error')} ({rc})")


def npy_shape(path: str):
    """(frames, cols) of a unit file from its header alone (no payload is read)."""
    lib = load_library()
    f, c = ctypes.c_int32(0), ctypes.c_int32(0)
    _check(lib.qvc_io_npy_shape(os.fsencode(path), ctypes.byref(f), ctypes.byref(c)), f"qvc_io_npy_shape({path})")
    return int(f.value), int(c.value)


def _paths(paths: Sequence[str]):
    arr = (ctypes.c_char_p * len(paths))()
    for i, p in enumerate(paths):
        arr[i] = os.fsencode(p)
    return arr


class IoPool:
    """A pool of native worker threads; ``load_units`` / ``write_wavs`` return when every file of the call is done."""

    def __init__(self, threads: int = 8):
        self.lib = load_library()
        self._h = ctypes.c_void_p(None)
        _check(self.lib.qvc_io_pool_create(int(threads), ctypes.byref(self._h)), "qvc_io_pool_create")
        self.threads = int(threads)

    def close(self) -> None:
        if self._h is not None and self._h.value:
            self.lib.qvc_io_pool_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def npy_shapes(self, paths: Sequence[str]):
        """[(frames, cols)] of many unit files from their headers alone, read in parallel on the pool."""
        n = len(paths)
        f, c = (ctypes.c_int32 * n)(), (ctypes.c_int32 * n)()
        _check(self.lib.qvc_io_npy_shapes(self._h, _paths(paths), n, f, c), "qvc_io_npy_shapes")
        return list(zip(f, c))

    def load_units(self, paths: Sequence[str], dst: torch.Tensor, lens_out: torch.Tensor) -> None:
        """Read ``len(paths)`` unit files into ``dst`` (n, slot_frames, cols) fp32 CPU (contiguous; pinned for async
        uploads): file i -> dst[i, :frames_i]; rows past an utterance's end are left as they are.  ``lens_out`` (n,)
        int32 CPU receives the frame counts."""
        n = len(paths)
        if dst.device.type != "cpu" or dst.dtype != torch.float32 or not dst.is_contiguous() or dst.dim() != 3 or dst.shape[0] < n:
            raise ValueError(f"dst must be a contiguous fp32 CPU tensor (>= {n}, slot_frames, cols), got {tuple(dst.shape)} {dst.dtype}")
        if lens_out.device.type != "cpu" or lens_out.dtype != torch.int32 or not lens_out.is_contiguous() or lens_out.numel() < n:
            raise ValueError("lens_out must be a contiguous int32 CPU tensor with one entry per file")
        _check(self.lib.qvc_io_load_units(self._h, _paths(paths), n, dst.data_ptr(), int(dst.shape[1]), int(dst.shape[2]),
                                          lens_out.data_ptr()), "qvc_io_load_units")

    def write_wavs(self, paths: Sequence[str], src: torch.Tensor, samples: Sequence[int], rate: int) -> None:
        """Write file i = src[i, :samples[i]] (fp32 CPU, rows contiguous) as a mono float32 wav at ``rate`` Hz."""
        n = len(paths)
        src2 = src.reshape(src.shape[0], -1)
        if src2.device.type != "cpu" or src2.dtype != torch.float32 or src2.stride(1) != 1 or src2.shape[0] < n:
            raise ValueError("src must be an fp32 CPU tensor with contiguous rows, one per file")
        if len(samples) != n or any(int(s) < 0 or int(s) > src2.shape[1] for s in samples):
            raise ValueError("one sample count per file, each within its row")
        cnt = (ctypes.c_int32 * n)(*[int(s) for s in samples])
        _check(self.lib.qvc_io_write_wavs(self._h, _paths(paths), n, src2.data_ptr(), int(src2.stride(0)), cnt, int(rate)),
               "qvc_io_write_wavs")
