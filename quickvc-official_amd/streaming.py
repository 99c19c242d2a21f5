"""Exact chunked conversion of many concurrent streams (BASELINE.json configs[4]).

The reference processes a whole utterance in one shot (no chunking, no streaming).  Every
op on the path is a convolution with a bounded receptive field -- measured on the oracle: an input
frame influences output frames within +-84 frames (enc_p WN 32, flow 32, decoder ~20) -- and the speaker
embedding g is global, so a window of ``hop + 2*context`` frames reproduces its central ``hop`` frames
exactly when ``context >= 84``.  Windows are clamped to the utterance, so the first / last window touch
the true sequence edges and stay exact there too (the convs zero-pad activations at the *sequence*
edges at every layer, which a zero-padded window could not reproduce).

All windows have ONE shape, (streams, 256, hop + 2*context), so the launch sequence is captured once
in a hipGraph and replayed per chunk.  Latency of a streamed chunk = ``context`` frames of look-ahead.
"""
from __future__ import annotations

from typing import Iterator, Optional, Tuple

import torch

RECEPTIVE_FRAMES = 84      # measured (tests/test_gpu_parity.py::test_chunked_streaming_is_exact re-checks it)


class ChunkedConverter:
    """Fixed-shape, graph-replayed windowed conversion for ``streams`` concurrent utterances."""

    def __init__(self, model, streams: int, hop_frames: int = 320, context: int = 88, use_graph: bool = True):
        if context < RECEPTIVE_FRAMES:
            raise ValueError(f"context {context} < receptive field {RECEPTIVE_FRAMES}: chunks would not be exact")
        self.model, self.streams, self.hop, self.context = model, streams, hop_frames, context
        self.window = hop_frames + 2 * context
        self.engine = model.engine()
        dev = self.engine.device
        mc = model.model_config
        self.spf = model.samples_per_frame
        self._unit = torch.zeros(streams, mc.get("unit_channels", 256), self.window, device=dev)
        self._noise = torch.zeros(streams, mc["inter_channels"], self.window, device=dev)
        self._g = torch.zeros(streams, mc["gin_channels"], device=dev)
        self._out = torch.zeros(streams, 1, self.window * self.spf, device=dev)
        self._graph = None
        self._stream = torch.cuda.Stream(dev)
        # the graph bakes the workspace pointer in: own the buffer (the engine's shared one is replaced, i.e. freed,
        # as soon as somebody asks the same engine for a larger batch)
        self._ws = self.engine.alloc_workspace(streams, self.window)
        if use_graph:
            with torch.cuda.stream(self._stream):
                self.engine.infer_batch(self._unit, self._g, self._noise, self._out, ws=self._ws)      # warm-up (code objects)
                self._stream.synchronize()
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph, stream=self._stream):
                    self.engine.infer_batch(self._unit, self._g, self._noise, self._out, ws=self._ws)

    def windows(self, total_frames: int) -> Iterator[Tuple[int, int, int]]:
        """(window_start, chunk_start, chunk_len) for every hop of an utterance of ``total_frames``."""
        if total_frames < self.window:
            raise ValueError(f"utterance of {total_frames} frames is shorter than one window ({self.window})")
        for t0 in range(0, total_frames, self.hop):
            n = min(self.hop, total_frames - t0)
            a = min(max(t0 - self.context, 0), total_frames - self.window)
            yield a, t0, n

    @torch.no_grad()
    def convert(self, unit: torch.Tensor, g: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """unit (S,256,T), g (S,gin), noise (S,inter,T) -> (S,1,T*samples_per_frame); == infer_batch on the whole thing."""
        S, _, T = unit.shape
        if S != self.streams:
            raise ValueError(f"built for {self.streams} streams, got {S}")
        dev = self.engine.device
        if noise is None:
            noise = torch.randn(S, self.model.model_config["inter_channels"], T, device=dev)
        unit, noise = unit.to(dev, torch.float32), noise.to(dev, torch.float32)
        g = g.to(dev, torch.float32)
        out = torch.empty(S, 1, T * self.spf, device=dev)
        # inputs (and the noise drawn above) may still be pending on the caller's stream: order the side stream after it
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._stream):
            for t in (unit, noise, g, out):
                t.record_stream(self._stream)
            self._g.copy_(g.reshape(S, -1))
            for a, t0, n in self.windows(T):
                self._unit.copy_(unit[:, :, a:a + self.window])
                self._noise.copy_(noise[:, :, a:a + self.window])
                if self._graph is not None:
                    self._graph.replay()
                else:
                    self.engine.infer_batch(self._unit, self._g, self._noise, self._out, ws=self._ws)
                lo = (t0 - a) * self.spf
                out[:, :, t0 * self.spf:(t0 + n) * self.spf].copy_(self._out[:, :, lo:lo + n * self.spf])
            self._stream.synchronize()
        return out


class StreamConverter:
    """Incremental conversion of ``streams`` concurrent streams, ``hop_frames`` new unit frames per step
    (qvc_stream_step; BASELINE.json configs[4]).  Unlike ChunkedConverter it needs nothing but the frames seen so
    far: the state (ring buffers at five points of the path, csrc/qvc_stream.h) lives in a tensor owned by this
    object, every step is ONE replay of a captured hipGraph (fixed shapes; the stream position is a device
    counter the graph itself advances), and a frame costs about (2*H + hop)/hop of its offline cost per segment
    instead of (2*88 + hop)/hop.  Output lags the input by ``self.lag`` frames -- the look-ahead the non-causal
    network needs; it is exact, including the first and last frames of a stream."""

    def __init__(self, model, streams: int, hop_frames: int = 320, use_graph: bool = True):
        from .engine import aligned_empty
        self.model, self.streams, self.hop = model, int(streams), int(hop_frames)
        self.engine = model.engine()
        dev = self.engine.device
        mc = model.model_config
        n_state, n_ws, self.lag, self.noise_lag = self.engine.stream_sizes(self.streams, self.hop)
        self.spf = model.samples_per_frame
        self._state = aligned_empty(n_state, dev)
        self._ws = aligned_empty(n_ws, dev)
        S, h = self.streams, self.hop
        self._unit = torch.zeros(S, mc.get("unit_channels", 256), h, device=dev)
        self._noise = torch.zeros(S, mc["inter_channels"], h, device=dev)
        self._g = torch.zeros(S, mc["gin_channels"], device=dev)
        self._out = torch.zeros(S, h * self.spf, device=dev)
        self._pos = torch.zeros(S, dtype=torch.int32, device=dev)
        self._len = torch.full((S,), 2 ** 30, dtype=torch.int32, device=dev)
        self._stream = torch.cuda.Stream(dev)
        self._graph = None
        self._pos_host = [0] * S
        self._len_host = [2 ** 30] * S
        self.reset()
        if use_graph:
            with torch.cuda.stream(self._stream):
                self._one_step()                                  # warm-up (code objects)
                self._stream.synchronize()
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph, stream=self._stream):
                    self._one_step()
            self.reset()

    def _one_step(self):
        self.engine.stream_step(self._state, self._ws, self._unit, self._g, self._noise, self._out, self._pos, self._len)
        self._pos.add_(self.hop)                                 # inside the graph: the position advances with every replay

    def _after_caller(self, *tensors) -> None:
        """Order the side stream after the caller's current stream (inputs such as a freshly computed ``g`` may still be
        pending there) and keep device inputs alive until the side stream is done with them."""
        dev = self.engine.device
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        for t in tensors:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(self._stream)

    def reset(self, g: Optional[torch.Tensor] = None, lengths: Optional[torch.Tensor] = None) -> None:
        """Start new streams in ALL slots: zero state, position 0; ``g`` (S, gin) speaker embeddings; ``lengths`` (S,) if known."""
        self._after_caller(g, lengths)
        with torch.cuda.stream(self._stream):
            self._state.zero_()
            self._pos.zero_()
            self._len.fill_(2 ** 30)
            if g is not None:
                self._g.copy_(g.reshape(self.streams, -1))
            if lengths is not None:
                self._len.copy_(torch.as_tensor(lengths, dtype=torch.int32))
        self._stream.synchronize()
        self._pos_host = [0] * self.streams
        self._len_host = [2 ** 30] * self.streams if lengths is None else [int(v) for v in torch.as_tensor(lengths).tolist()]

    def end(self, lengths) -> None:
        """Tell the converter where ALL streams end (frames); keep stepping until ``position - lag >= length`` to flush."""
        self._after_caller(lengths)
        with torch.cuda.stream(self._stream):
            self._len.copy_(torch.as_tensor(lengths, dtype=torch.int32))
        self._len_host = [int(v) for v in torch.as_tensor(lengths).tolist()]

    # ---- per-slot lifecycle: streams of a server start and end at different times
    def start(self, slot: int, g: torch.Tensor, length: Optional[int] = None) -> None:
        """Admit a new stream into ``slot``: that slot's ring rows are zeroed and its position goes back to 0 on the
        device (qvc_stream_reset_slot); the other slots keep running and the captured graph is unchanged.  ``g`` (gin,)
        or (1, gin): the stream's speaker embedding; ``length``: its frame count if already known."""
        if not 0 <= slot < self.streams:
            raise ValueError(f"slot {slot} outside [0, {self.streams})")
        n = 2 ** 30 if length is None else int(length)
        self._after_caller(g)
        with torch.cuda.stream(self._stream):
            self.engine.stream_reset_slot(self._state, self.streams, self.hop, slot, n, self._pos, self._len)
            self._g[slot].copy_(g.reshape(-1), non_blocking=True)
        self._pos_host[slot] = 0
        self._len_host[slot] = n

    def end_slot(self, slot: int, length: int) -> None:
        """The stream in ``slot`` ends after ``length`` frames; it is flushed once ``finished(slot)``."""
        with torch.cuda.stream(self._stream):
            self._len[slot:slot + 1].fill_(int(length))
        self._len_host[slot] = int(length)

    def position(self, slot: int) -> int:
        """First frame the NEXT step feeds into ``slot`` (host mirror of the device counter)."""
        return self._pos_host[slot]

    def finished(self, slot: int) -> bool:
        """Every output frame of the stream in ``slot`` has been produced: the slot can take a new stream."""
        return self._pos_host[slot] - self.lag >= self._len_host[slot]

    @torch.no_grad()
    def step(self, unit_new: torch.Tensor, noise_new: torch.Tensor) -> torch.Tensor:
        """Feed frames [pos, pos+hop) of every stream (unit_new (S, 256, hop)) and the noise for frames
        [pos - noise_lag, ... + hop) (noise_new (S, inter, hop)); returns the waveform of frames
        [pos - lag, pos - lag + hop) as (S, hop * samples_per_frame) -- zeros where that lies outside the stream."""
        dev = self.engine.device
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._stream):
            self._unit.copy_(unit_new, non_blocking=True)
            self._noise.copy_(noise_new, non_blocking=True)
            if self._graph is not None:
                self._graph.replay()
            else:
                self._one_step()
            out = self._out.clone()
        torch.cuda.current_stream(dev).wait_stream(self._stream)
        self._pos_host = [p + self.hop for p in self._pos_host]
        return out

    @torch.no_grad()
    def convert(self, unit: torch.Tensor, g: torch.Tensor, noise: Optional[torch.Tensor] = None,
                lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Convenience: stream whole utterances through the step interface.  unit (S, 256, T), g (S, gin),
        noise (S, inter, T) -> (S, 1, T * samples_per_frame); equals infer_batch on the whole thing.  ``lengths`` (S,):
        streams that end before T (their rows are zero after the end, as with infer_batch_ragged)."""
        S, UC, T = unit.shape
        if S != self.streams:
            raise ValueError(f"built for {self.streams} streams, got {S}")
        dev = self.engine.device
        inter = self.model.model_config["inter_channels"]
        if noise is None:
            noise = torch.randn(S, inter, T, device=dev)
        unit, noise = unit.to(dev, torch.float32), noise.to(dev, torch.float32)
        h, lag, nl, spf = self.hop, self.lag, self.noise_lag, self.spf
        self.reset(g.to(dev, torch.float32), torch.full((S,), T, dtype=torch.int32) if lengths is None else lengths)
        out = torch.zeros(S, 1, T * spf, device=dev)
        pad_u = torch.nn.functional.pad(unit, (0, h + lag))                      # frames past the end: ignored by the kernels
        pad_n = torch.nn.functional.pad(noise, (nl, h + lag))                    # frame f of the noise sits at column f + nl
        steps = -(-(T + lag) // h)
        for n in range(steps):
            s = n * h
            chunk = self.step(pad_u[:, :, s:s + h], pad_n[:, :, s:s + h])         # noise frames [s - nl, s - nl + h)
            f0 = s - lag
            lo, hi = max(f0, 0), min(f0 + h, T)
            if hi > lo:
                out[:, 0, lo * spf:hi * spf] = chunk[:, (lo - f0) * spf:(hi - f0) * spf]
        torch.cuda.synchronize(dev)
        return out
