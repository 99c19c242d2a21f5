"""QvcEngine: owns the packed weights and workspace on one GPU and calls the C ABI.

PyTorch is used here for device memory and streams only; every FLOP of enc_p / flow / dec
runs inside libqvc_hip.so.  Buffers are PyTorch allocations handed over as raw pointers;
kernels are enqueued on ``torch.cuda.current_stream()`` so the call composes with PyTorch
work (the LSTM speaker encoder) and can be captured in a ``torch.cuda.CUDAGraph``.
"""
from __future__ import annotations

import ctypes
import functools
from typing import Dict, Optional

import torch

from . import lib as L


def aligned_empty(nbytes: int, device) -> torch.Tensor:
    return _aligned_empty(nbytes, device)


def _aligned_empty(nbytes: int, device) -> torch.Tensor:
    raw = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    shift = (-raw.data_ptr()) % 256
    return raw[shift:shift + nbytes]


def _on_device(fn):
    """Run a method with the engine's GPU as the current device: the library keys per-device state (the opt-in for
    more than 64 KiB of dynamic LDS) on hipGetDevice(), and a process may drive several GPUs."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        with torch.cuda.device(self.device):
            return fn(self, *args, **kwargs)
    return wrapper


class QvcEngine:
    def __init__(self, model_config: dict, state_dict: Optional[Dict[str, torch.Tensor]], device, parallel_branches: bool = False,
                 pack: bool = True):
        """``pack=False`` (multi-GPU start-up, ranks other than the one that owns the checkpoint): allocate the blob
        without packing anything -- ``state_dict`` may be None -- and let ``dist.broadcast_blob`` fill it."""
        self.lib = L.load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.QvcError("QvcEngine needs a HIP device (no CPU fallback)")
        with torch.cuda.device(self.device):
            L.check(self.lib, self.lib.qvc_device_check(), "qvc_device_check")
        self.model_config = dict(model_config)
        self.cfg = L.make_config(model_config)
        if pack:
            host_blob = L.pack_weights(self.lib, self.cfg, state_dict)
            self.blob = _aligned_empty(host_blob.numel(), self.device)
            self.blob.copy_(host_blob)
        else:
            n = int(self.lib.qvc_blob_bytes(ctypes.byref(self.cfg)))
            if n < 0:
                L.check(self.lib, n, "qvc_blob_bytes")
            self.blob = _aligned_empty(n, self.device)
            state_dict = state_dict or {}
        # optional fork/join resources: the three ResBlocks of a stage as parallel branches (two aux streams +
        # events).  Measured neutral on MI355X (2.934 vs 2.949 ms/step: each launch already fills the chip), and
        # concurrent kernels blur per-kernel profiles, so it is off by default.
        self._aux = ctypes.c_void_p(None)
        if parallel_branches:
            with torch.cuda.device(self.device):
                L.check(self.lib, self.lib.qvc_aux_create(ctypes.byref(self._aux)), "qvc_aux_create")
        self._ws: Optional[torch.Tensor] = None
        self._ws_key = None
        # speaker encoder: own blob, packed on first use from the same state dict
        self._spk_sd = {k: v for k, v in state_dict.items() if k.startswith("enc_spk.")}
        # posterior encoder: own blob too, packed on first use
        self._encq_sd = {k: v for k, v in state_dict.items() if k.startswith("enc_q.")}
        self._encq_blob: Optional[torch.Tensor] = None
        self._spk_blob: Optional[torch.Tensor] = None
        self._spk_ws: Optional[torch.Tensor] = None
        n = model_config["gen_istft_hop_size"] * model_config["subbands"]
        for u in model_config["upsample_rates"]:
            n *= u
        self.samples_per_frame = n

    def __del__(self):
        try:
            if getattr(self, "_aux", None) is not None and self._aux.value:
                self.lib.qvc_aux_destroy(self._aux)
                self._aux = ctypes.c_void_p(None)
        except Exception:
            pass

    # weights can also arrive from another rank (one broadcast at start-up)
    def load_blob_(self, blob: torch.Tensor) -> None:
        self.blob.copy_(blob)

    def alloc_workspace(self, batch: int, frames: int) -> torch.Tensor:
        """A private workspace for (batch, frames).  Anything that bakes workspace pointers into a captured graph
        must own the buffer it captured (the shared one below is replaced when a larger request arrives)."""
        n = int(self.lib.qvc_workspace_bytes(ctypes.byref(self.cfg), batch, frames))
        if n < 0:
            L.check(self.lib, n, "qvc_workspace_bytes")
        return _aligned_empty(n, self.device)

    def workspace(self, batch: int, frames: int) -> torch.Tensor:
        key = (batch, frames)
        if self._ws is None or self._ws_key != key:
            n = int(self.lib.qvc_workspace_bytes(ctypes.byref(self.cfg), batch, frames))
            if n < 0:
                L.check(self.lib, n, "qvc_workspace_bytes")
            if self._ws is None or self._ws.numel() < n:
                self._ws = _aligned_empty(n, self.device)
            self._ws_key = key
        return self._ws

    @staticmethod
    def _f32(t: torch.Tensor, device) -> torch.Tensor:
        return t.to(device=device, dtype=torch.float32).contiguous()

    @_on_device
    def infer_batch(self, unit: torch.Tensor, g: torch.Tensor, noise: torch.Tensor,
                    out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
        """unit (B,256,T), g (B,gin), noise (B,inter,T) -> (B,1,T*samples_per_frame) fp32.
        ``ws``: a caller-owned workspace from alloc_workspace(B, T) (graph owners); default = the shared one."""
        B, cu, T = unit.shape
        mc = self.model_config
        if cu != mc.get("unit_channels", 256) or g.shape != (B, mc["gin_channels"]) or \
                tuple(noise.shape) != (B, mc["inter_channels"], T):
            raise ValueError(f"bad input shapes: unit {tuple(unit.shape)}, g {tuple(g.shape)}, noise {tuple(noise.shape)}")
        unit, g, noise = (self._f32(t, self.device) for t in (unit, g, noise))
        if out is None:
            out = torch.empty(B, 1, T * self.samples_per_frame, dtype=torch.float32, device=self.device)
        if ws is None:
            ws = self.workspace(B, T)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        st = self.lib.qvc_infer_batch_ex(ctypes.byref(self.cfg), self.blob.data_ptr(), unit.data_ptr(), g.data_ptr(),
                                         noise.data_ptr(), out.data_ptr(), B, T, ws.data_ptr(), ws.numel(), stream,
                                         self._aux)
        L.check(self.lib, st, "qvc_infer_batch")
        return out

    @_on_device
    def infer_batch_ragged(self, unit: torch.Tensor, g: torch.Tensor, noise: torch.Tensor, frames: torch.Tensor,
                           out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None,
                           unit_fm: bool = False) -> torch.Tensor:
        """A batch of utterances of DIFFERENT lengths: unit (B,256,Tmax) / noise (B,inter,Tmax) padded (the padding's
        content is ignored), frames (B,) int32 lengths -> (B,1,Tmax*samples_per_frame); row b holds utterance b's
        waveform in its first frames[b]*samples_per_frame samples and zeros after.
        ``unit_fm``: unit is (B,Tmax,256) -- frame-major, as the reference's unit files are on disk
        (dataset/encode.py:38), so a batch read by fileio.IoPool.load_units uploads as it is."""
        mc = self.model_config
        if unit_fm:
            B, T, cu = unit.shape
        else:
            B, cu, T = unit.shape
        if cu != mc.get("unit_channels", 256) or g.shape != (B, mc["gin_channels"]) or \
                tuple(noise.shape) != (B, mc["inter_channels"], T) or tuple(frames.shape) != (B,):
            raise ValueError(f"bad input shapes: unit {tuple(unit.shape)}, g {tuple(g.shape)}, noise {tuple(noise.shape)}, frames {tuple(frames.shape)}")
        if frames.device.type != "cuda":                       # host-side lengths can be validated for free
            if int(frames.min()) < 2 or int(frames.max()) > T:
                raise ValueError(f"frames must lie in [2, {T}], got [{int(frames.min())}, {int(frames.max())}]")
        unit, g, noise = (self._f32(t, self.device) for t in (unit, g, noise))
        lens = frames.to(device=self.device, dtype=torch.int32).contiguous()
        if out is None:
            out = torch.empty(B, 1, T * self.samples_per_frame, dtype=torch.float32, device=self.device)
        if ws is None:
            ws = self.workspace(B, T)
        fn = self.lib.qvc_infer_batch_ragged_fm if unit_fm else self.lib.qvc_infer_batch_ragged
        st = fn(ctypes.byref(self.cfg), self.blob.data_ptr(), unit.data_ptr(), g.data_ptr(),
                noise.data_ptr(), out.data_ptr(), B, T, lens.data_ptr(), ws.data_ptr(), ws.numel(),
                torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib, st, "qvc_infer_batch_ragged")
        return out

    # ---- streaming (qvc_stream_step): sizes, lags and one step; the state / workspace tensors belong to the caller
    def stream_sizes(self, batch: int, hop: int):
        """(state_bytes, workspace_bytes, lag_frames, noise_lag_frames) for `batch` streams fed `hop` frames per step."""
        vals = (int(self.lib.qvc_stream_state_bytes(ctypes.byref(self.cfg), batch, hop)),
                int(self.lib.qvc_stream_workspace_bytes(ctypes.byref(self.cfg), batch, hop)),
                int(self.lib.qvc_stream_lag_frames(ctypes.byref(self.cfg))),
                int(self.lib.qvc_stream_noise_lag_frames(ctypes.byref(self.cfg))))
        for v in vals:
            if v < 0:
                L.check(self.lib, v, "qvc_stream_*_bytes")
        return vals

    @_on_device
    def stream_step(self, state: torch.Tensor, ws: torch.Tensor, unit_new: torch.Tensor, g: torch.Tensor, noise_new: torch.Tensor,
                    out: torch.Tensor, pos: torch.Tensor, lens: torch.Tensor) -> None:
        """One hop for every stream (all tensors fp32 / int32, contiguous, on this device; see include/qvc.h)."""
        B, _, hop = unit_new.shape
        st = self.lib.qvc_stream_step(ctypes.byref(self.cfg), self.blob.data_ptr(), state.data_ptr(), state.numel(),
                                      unit_new.data_ptr(), g.data_ptr(), noise_new.data_ptr(), out.data_ptr(), B, hop,
                                      pos.data_ptr(), lens.data_ptr(), ws.data_ptr(), ws.numel(),
                                      torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib, st, "qvc_stream_step")

    @_on_device
    def stream_reset_slot(self, state: torch.Tensor, batch: int, hop: int, slot: int, length: int, pos: torch.Tensor,
                          lens: torch.Tensor) -> None:
        """qvc_stream_reset_slot: zero slot `slot`'s ring rows, pos[slot] = 0, lens[slot] = length (on the current stream)."""
        st = self.lib.qvc_stream_reset_slot(ctypes.byref(self.cfg), state.data_ptr(), state.numel(), batch, hop, int(slot),
                                            int(length), pos.data_ptr(), lens.data_ptr(),
                                            torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib, st, "qvc_stream_reset_slot")

    @_on_device
    def speaker_embed(self, mel: torch.Tensor) -> torch.Tensor:
        """SpeakerEncoder.embed_utterance for a batch (models.py:528-546): mel (U, n_mel, F) -> g (U, gin)."""
        if mel.dim() != 3 or mel.shape[1] != int(self.cfg.n_mel_channels) or mel.shape[2] < 1:
            raise ValueError(f"mel must be (U, {int(self.cfg.n_mel_channels)}, frames), got {tuple(mel.shape)}")
        if self._spk_blob is None:
            if not self._spk_sd:
                raise L.QvcError("the state dict holds no enc_spk.* weights")
            host = L.pack_weights(self.lib, self.cfg, self._spk_sd, which="spk")
            self._spk_blob = _aligned_empty(host.numel(), self.device)
            self._spk_blob.copy_(host)
        mel = self._f32(mel, self.device)
        U, _, F = mel.shape
        n = int(self.lib.qvc_spk_workspace_bytes(ctypes.byref(self.cfg), U, F))
        if n < 0:
            L.check(self.lib, n, "qvc_spk_workspace_bytes")
        if self._spk_ws is None or self._spk_ws.numel() < n:
            self._spk_ws = _aligned_empty(n, self.device)
        g = torch.empty(U, self.model_config["gin_channels"], dtype=torch.float32, device=self.device)
        st = self.lib.qvc_speaker_embed(ctypes.byref(self.cfg), self._spk_blob.data_ptr(), mel.data_ptr(), g.data_ptr(),
                                        U, F, self._spk_ws.data_ptr(), self._spk_ws.numel(),
                                        torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib, st, "qvc_speaker_embed")
        return g

    @_on_device
    def enc_q(self, spec: torch.Tensor, g: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        """Posterior encoder, models.py:617: spec (B, spec_channels, T), g (B, gin), noise (B, inter, T) -> z [B][T][inter]."""
        B, cs, T = spec.shape
        mc = self.model_config
        if cs != int(self.cfg.spec_channels) or g.shape != (B, mc["gin_channels"]) or tuple(noise.shape) != (B, mc["inter_channels"], T):
            raise ValueError(f"bad input shapes: spec {tuple(spec.shape)}, g {tuple(g.shape)}, noise {tuple(noise.shape)}")
        if self._encq_blob is None:
            if not self._encq_sd:
                raise L.QvcError("the state dict holds no enc_q.* weights")
            host = L.pack_weights(self.lib, self.cfg, self._encq_sd, which="encq")
            self._encq_blob = _aligned_empty(host.numel(), self.device)
            self._encq_blob.copy_(host)
        spec, g, noise = (self._f32(t, self.device) for t in (spec, g, noise))
        z = torch.empty(B, T, mc["inter_channels"], dtype=torch.float32, device=self.device)
        ws = self.workspace(B, T)
        st = self.lib.qvc_enc_q(ctypes.byref(self.cfg), self._encq_blob.data_ptr(), spec.data_ptr(), g.data_ptr(), noise.data_ptr(),
                                z.data_ptr(), B, T, ws.data_ptr(), ws.numel(), torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib, st, "qvc_enc_q")
        return z

    @_on_device
    def flow_forward(self, z_fm: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
        """ResidualCouplingBlock.forward(reverse=False), models.py:618: z [B][T][inter] -> z_p (a new tensor)."""
        B, T, _ = z_fm.shape
        z = self._f32(z_fm, self.device).clone()
        g = self._f32(g, self.device)
        ws = self.workspace(B, T)
        st = self.lib.qvc_flow_forward(ctypes.byref(self.cfg), self.blob.data_ptr(), z.data_ptr(), g.data_ptr(), B, T,
                                       ws.data_ptr(), ws.numel(), torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib, st, "qvc_flow_forward")
        return z

    @_on_device
    def infer_batch_timed(self, unit, g, noise, out=None, max_records: int = 512):
        """Same launches with per-launch HIP-event timing; returns (out, [dict(name, ms, flops, bytes)])."""
        B, _, T = unit.shape
        unit, g, noise = (self._f32(t, self.device) for t in (unit, g, noise))
        if out is None:
            out = torch.empty(B, 1, T * self.samples_per_frame, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, T)
        rec = (L.QvcLaunchRecord * max_records)()
        n = ctypes.c_int32(0)
        st = self.lib.qvc_infer_batch_timed(ctypes.byref(self.cfg), self.blob.data_ptr(), unit.data_ptr(), g.data_ptr(),
                                            noise.data_ptr(), out.data_ptr(), B, T, ws.data_ptr(), ws.numel(),
                                            torch.cuda.current_stream(self.device).cuda_stream, rec, max_records,
                                            ctypes.byref(n))
        L.check(self.lib, st, "qvc_infer_batch_timed")
        return out, [dict(name=rec[i].name.decode(), ms=float(rec[i].ms), flops=float(rec[i].flops),
                          bytes=float(rec[i].bytes)) for i in range(min(n.value, max_records))]

    # ---- stage entry points (frame-major tensors), used by the stage-level parity tests
    @_on_device
    def enc_p(self, unit, noise):
        B, _, T = unit.shape
        unit, noise = self._f32(unit, self.device), self._f32(noise, self.device)
        z = torch.empty(B, T, self.model_config["inter_channels"], dtype=torch.float32, device=self.device)
        ws = self.workspace(B, T)
        st = self.lib.qvc_enc_p(ctypes.byref(self.cfg), self.blob.data_ptr(), unit.data_ptr(), noise.data_ptr(),
                                z.data_ptr(), B, T, ws.data_ptr(), ws.numel(),
                                torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib, st, "qvc_enc_p")
        return z

    @_on_device
    def wn_stack(self, which: int, x_fm, g=None):
        """WN.forward (modules.py:69-114) of stack ``which`` (0 = enc_p.enc, 1+i = flow.flows[2i].enc): [B][T][hidden] -> same."""
        B, T, _ = x_fm.shape
        x = self._f32(x_fm, self.device)
        gg = self._f32(g, self.device) if g is not None else None
        out = torch.empty_like(x)
        ws = self.workspace(B, T)
        st = self.lib.qvc_wn_stack(ctypes.byref(self.cfg), self.blob.data_ptr(), int(which), x.data_ptr(),
                                   gg.data_ptr() if gg is not None else None, out.data_ptr(), B, T, ws.data_ptr(), ws.numel(),
                                   torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib, st, "qvc_wn_stack")
        return out

    @_on_device
    def flow_reverse(self, z_fm, g):
        B, T, _ = z_fm.shape
        z = self._f32(z_fm, self.device).clone()
        g = self._f32(g, self.device)
        ws = self.workspace(B, T)
        st = self.lib.qvc_flow_reverse(ctypes.byref(self.cfg), self.blob.data_ptr(), z.data_ptr(), g.data_ptr(), B, T,
                                       ws.data_ptr(), ws.numel(), torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib, st, "qvc_flow_reverse")
        return z

    @_on_device
    def dec_trunk(self, z_fm, g):
        B, T, _ = z_fm.shape
        z, g = self._f32(z_fm, self.device), self._f32(g, self.device)
        frames = T * (self.samples_per_frame // (self.model_config["gen_istft_hop_size"] * self.model_config["subbands"])) + 1
        pc = self.model_config["subbands"] * 2 * (self.model_config["gen_istft_n_fft"] // 2 + 1)
        post = torch.empty(B, frames, pc, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, T)
        st = self.lib.qvc_dec_trunk(ctypes.byref(self.cfg), self.blob.data_ptr(), z.data_ptr(), g.data_ptr(),
                                    post.data_ptr(), B, T, ws.data_ptr(), ws.numel(),
                                    torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib, st, "qvc_dec_trunk")
        return post

    @_on_device
    def istft_synth(self, post_fm, want_bands: bool = False):
        B, F, _ = post_fm.shape
        post = self._f32(post_fm, self.device)
        hop, sb = self.model_config["gen_istft_hop_size"], self.model_config["subbands"]
        out = torch.empty(B, 1, sb * hop * (F - 1), dtype=torch.float32, device=self.device)
        ymb = torch.empty(B, sb, hop * (F - 1), dtype=torch.float32, device=self.device) if want_bands else None
        st = self.lib.qvc_istft_synth(ctypes.byref(self.cfg), self.blob.data_ptr(), post.data_ptr(), out.data_ptr(),
                                      ymb.data_ptr() if want_bands else None, B, F,
                                      torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib, st, "qvc_istft_synth")
        return (out, ymb) if want_bands else out
