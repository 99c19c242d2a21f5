"""``SynthesizerTrn`` -- the reference's model object, re-hosted on HIP kernels.

Drop-in surface (models.py:549-642 of the reference, used by convert.py:35-41,81):
same constructor signature (unknown kwargs swallowed), same ``state_dict`` keys and
shapes (so ``utils.load_checkpoint`` of the reference's ``.pth`` works unchanged),
``.cuda()``, ``.eval()``, ``.parameters()``, ``.infer(unit, mel)``.

What is different underneath: the sub-modules below are *parameter holders* only.
There is no PyTorch forward for enc_p / flow / dec: ``infer`` and ``infer_batch``
hand the raw tensors to ``libqvc_hip.so`` (engine.py), which runs the whole path as
hand-written gfx950 kernels -- the speaker encoder (3-layer LSTM)
included.  If the HIP library is missing or no GPU is present, ``infer`` raises; there is
no CPU fallback and no torch forward anywhere in this module.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch import Tensor, nn


# ----------------------------------------------------------------------------- holders
class ConvParams(nn.Module):
    """Parameters of one Conv1d / ConvTranspose1d, weight-normed or plain.

    Weight-normed convs expose ``bias, weight_g, weight_v`` (old-style
    ``torch.nn.utils.weight_norm`` naming); plain ones ``weight, bias``.
    ``shape`` is the weight shape: (Cout, Cin, K) for Conv1d, (Cin, Cout, K) for
    ConvTranspose1d -- ``weight_g`` is (shape[0], 1, 1) either way.
    """

    def __init__(self, shape: Sequence[int], weight_norm: bool, bias: bool = True, bias_size: Optional[int] = None):
        super().__init__()
        shape = tuple(int(s) for s in shape)
        nb = shape[0] if bias_size is None else int(bias_size)
        if weight_norm:
            self.bias = nn.Parameter(torch.zeros(nb)) if bias else None
            self.weight_g = nn.Parameter(torch.ones(shape[0], 1, 1))
            self.weight_v = nn.Parameter(torch.zeros(shape))
        else:
            self.weight = nn.Parameter(torch.zeros(shape))
            self.bias = nn.Parameter(torch.zeros(nb)) if bias else None


class WNParams(nn.Module):
    """modules.py:37-67: optional cond_layer, n in_layers (h->2h, k), n res_skip_layers (1x1)."""

    def __init__(self, hidden: int, kernel_size: int, n_layers: int, gin_channels: int):
        super().__init__()
        assert kernel_size % 2 == 1, f"kernel should be odd number, but {kernel_size}"
        self.hidden_channels, self.kernel_size, self.n_layers, self.gin_channels = hidden, kernel_size, n_layers, gin_channels
        if gin_channels != 0:
            self.cond_layer = ConvParams((2 * hidden * n_layers, gin_channels, 1), True)
        self.in_layers = nn.ModuleList(ConvParams((2 * hidden, hidden, kernel_size), True) for _ in range(n_layers))
        self.res_skip_layers = nn.ModuleList(
            ConvParams(((2 * hidden if i < n_layers - 1 else hidden), hidden, 1), True) for i in range(n_layers))


class CondNormalWNParams(nn.Module):
    """models.py:54-73."""

    def __init__(self, in_ch: int, out_ch: int, hidden: int, kernel_size: int, n_layers: int, gin: int):
        super().__init__()
        self.pre = ConvParams((hidden, in_ch, 1), False)
        self.enc = WNParams(hidden, kernel_size, n_layers, gin)
        self.proj = ConvParams((2 * out_ch, hidden, 1), False)


class CouplingLayerParams(nn.Module):
    """modules.py:173-197."""

    def __init__(self, channels: int, hidden: int, kernel_size: int, n_layers: int, gin: int):
        super().__init__()
        assert channels % 2 == 0, "channels should be divisible by 2"
        self.pre = ConvParams((hidden, channels // 2, 1), False)
        self.enc = WNParams(hidden, kernel_size, n_layers, gin)
        self.post = ConvParams((channels // 2, hidden, 1), False)


class _NoParams(nn.Module):
    """Placeholder for the reference's parameter-free Flip (modules.py:165-170) so that
    ``flow.flows`` keeps its even indices for the coupling layers."""


class CouplingBlockParams(nn.Module):
    """models.py:17-37."""

    def __init__(self, channels: int, hidden: int, kernel_size: int, n_layers: int, n_flows: int, gin: int):
        super().__init__()
        self.flows = nn.ModuleList()
        for _ in range(n_flows):
            self.flows.append(CouplingLayerParams(channels, hidden, kernel_size, n_layers, gin))
            self.flows.append(_NoParams())


class ResBlock1Params(nn.Module):
    """modules.py:128-145."""

    def __init__(self, ch: int, k: int):
        super().__init__()
        self.convs1 = nn.ModuleList(ConvParams((ch, ch, k), True) for _ in range(3))
        self.convs2 = nn.ModuleList(ConvParams((ch, ch, k), True) for _ in range(3))


class _Window(nn.Module):
    def __init__(self, n_fft: int):
        super().__init__()
        self.register_buffer("window", torch.hann_window(n_fft))


def pqmf_filters(subbands: int = 4, taps: int = 62, cutoff_ratio: float = 0.15, beta: float = 9.0):
    """Cosine-modulated PQMF bank (the design used by pqmf.py:16-44,65-76 of the reference):
    Kaiser-windowed sinc prototype, band k modulated by (2k+1)*pi/(2*subbands).
    Returns (analysis (S,1,taps+1), synthesis (1,S,taps+1)) fp32 tensors."""
    import numpy as np
    n = np.arange(taps + 1, dtype=np.float64)
    centred = n - 0.5 * taps
    with np.errstate(invalid="ignore", divide="ignore"):
        proto = np.sin(np.pi * cutoff_ratio * centred) / (np.pi * centred)
    proto[taps // 2] = cutoff_ratio
    proto *= np.kaiser(taps + 1, beta)
    ana = np.zeros((subbands, taps + 1))
    syn = np.zeros((subbands, taps + 1))
    for k in range(subbands):
        arg = (2 * k + 1) * (np.pi / (2 * subbands)) * (n - (taps - 1) / 2.0)
        ana[k] = 2.0 * proto * np.cos(arg + ((-1) ** k) * np.pi / 4.0)
        syn[k] = 2.0 * proto * np.cos(arg - ((-1) ** k) * np.pi / 4.0)
    return torch.from_numpy(ana).float().unsqueeze(1), torch.from_numpy(syn).float().unsqueeze(0)


class _PQMFBuffers(nn.Module):
    """Buffers the reference's PQMF module registers (pqmf.py:83-90)."""

    def __init__(self, subbands: int):
        super().__init__()
        ana, syn = pqmf_filters(subbands)
        self.register_buffer("analysis_filter", ana)
        self.register_buffer("synthesis_filter", syn)
        upd = torch.zeros(subbands, subbands, subbands)
        for k in range(subbands):
            upd[k, k, 0] = 1.0
        self.register_buffer("updown_filter", upd)


class GeneratorParams(nn.Module):
    """Parameters of the three generators, models.py:98-160 / 195-248 / 304-358."""

    def __init__(self, kind: str, initial_channel: int, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel: int, upsample_kernel_sizes,
                 n_fft: int, hop: int, subbands: int, gin_channels: int):
        super().__init__()
        self.kind = kind
        self.subbands = int(subbands) if kind != "istft" else 1
        n_bins = n_fft // 2 + 1
        self.conv_pre = ConvParams((upsample_initial_channel, initial_channel, 7), True)
        self.cond = ConvParams((upsample_initial_channel, gin_channels, 1), False)
        self.ups = nn.ModuleList()
        ch = upsample_initial_channel
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            cin, ch = upsample_initial_channel // (2 ** i), upsample_initial_channel // (2 ** (i + 1))
            self.ups.append(ConvParams((cin, ch, k), True, bias_size=ch))
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            c = upsample_initial_channel // (2 ** (i + 1))
            for k, _d in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(ResBlock1Params(c, k))
        post_name = "conv_post" if kind == "istft" else "subband_conv_post"
        setattr(self, post_name, ConvParams((self.subbands * 2 * n_bins, ch, 7), True))
        self.stft = _Window(n_fft)
        if kind == "multiband":
            self.pqmf = _PQMFBuffers(self.subbands)
        if kind == "multistream":
            upd = torch.zeros(self.subbands, self.subbands, self.subbands)
            for k in range(self.subbands):
                upd[k, k, 0] = 1.0
            self.register_buffer("updown_filter", upd)
            self.multistream_conv_post = ConvParams((1, 4, 63), True, bias=False)


# ----------------------------------------------------------------------------- speaker encoder
class _LSTMParams(nn.Module):
    """Parameters of ``nn.LSTM(input, hidden, layers)`` under PyTorch's own names (weight_ih_l{k}, weight_hh_l{k},
    bias_ih_l{k}, bias_hh_l{k}; gate order i, f, g, o) -- what the reference's checkpoint stores for ``enc_spk.lstm``."""

    def __init__(self, input_size: int, hidden: int, layers: int):
        super().__init__()
        for k in range(layers):
            cin = input_size if k == 0 else hidden
            setattr(self, f"weight_ih_l{k}", nn.Parameter(torch.zeros(4 * hidden, cin)))
            setattr(self, f"weight_hh_l{k}", nn.Parameter(torch.zeros(4 * hidden, hidden)))
            setattr(self, f"bias_ih_l{k}", nn.Parameter(torch.zeros(4 * hidden)))
            setattr(self, f"bias_hh_l{k}", nn.Parameter(torch.zeros(4 * hidden)))


class _LinearParams(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))


class SpeakerEncoder(nn.Module):
    """Parameter holder for ``enc_spk.*`` (models.py:507-513: 3-layer LSTM 80 -> hidden, Linear hidden -> embedding).

    No forward: ``SynthesizerTrn.speaker_embed`` runs ``embed_utterance`` (models.py:528-546) on the HIP kernels of
    csrc/qvc_spk.hip; the fp32 torch restatement used to check them lives in ``oracle/qvc_oracle.py``."""

    def __init__(self, mel_n_channels=80, model_num_layers=3, model_hidden_size=256, model_embedding_size=256):
        super().__init__()
        self.lstm = _LSTMParams(mel_n_channels, model_hidden_size, model_num_layers)
        self.linear = _LinearParams(model_hidden_size, model_embedding_size)


# ----------------------------------------------------------------------------- the model
class SynthesizerTrn(nn.Module):
    """QuickVC generator with the hot path on gfx950 kernels (reference: models.py:549-642)."""

    def __init__(self, spec_channels: int, segment_size: int, inter_channels: int, hidden_channels: int,
                 resblock_kernel_sizes: List[int], resblock_dilation_sizes: List[List[int]],
                 upsample_rates: List[int], upsample_initial_channel: int, upsample_kernel_sizes: List[int],
                 gen_istft_n_fft: int, gen_istft_hop_size: int,
                 istft_vits: bool = False, ms_istft_vits: bool = False, mb_istft_vits: bool = False,
                 subbands=False, gin_channels: int = 0, operand_dtype: str = "f16", verbose: bool = False,
                 **kwargs):
        super().__init__()
        if verbose:
            print(f"Loaded but not used: {kwargs}")
        if kwargs.get("resblock"):
            assert kwargs["resblock"] == "1", "ResBlock2 support is droped."   # models.py:574-575
        kind = "multiband" if mb_istft_vits else ("multistream" if ms_istft_vits else ("istft" if istft_vits else None))
        if kind is None:
            raise RuntimeError(f"Not-supported decoder flag: {mb_istft_vits}/{ms_istft_vits}/{istft_vits}")
        self.segment_size = segment_size
        unit_channels = 256                                                      # models.py:579
        self.model_config = dict(
            unit_channels=unit_channels, inter_channels=inter_channels, hidden_channels=hidden_channels,
            gin_channels=gin_channels, resblock_kernel_sizes=list(resblock_kernel_sizes),
            resblock_dilation_sizes=[list(d) for d in resblock_dilation_sizes],
            upsample_rates=list(upsample_rates), upsample_initial_channel=upsample_initial_channel,
            upsample_kernel_sizes=list(upsample_kernel_sizes), gen_istft_n_fft=gen_istft_n_fft,
            gen_istft_hop_size=gen_istft_hop_size, subbands=(int(subbands) if kind != "istft" else 1),
            decoder=kind, operand_dtype=operand_dtype, spec_channels=int(spec_channels))

        self.enc_q = CondNormalWNParams(spec_channels, inter_channels, hidden_channels, 5, 16, gin_channels)
        self.enc_p = CondNormalWNParams(unit_channels, inter_channels, hidden_channels, 5, 16, 0)
        self.flow = CouplingBlockParams(inter_channels, hidden_channels, 5, 4, 4, gin_channels)
        self.enc_spk = SpeakerEncoder(model_hidden_size=gin_channels, model_embedding_size=gin_channels)
        self.dec = GeneratorParams(kind, inter_channels, resblock_kernel_sizes, resblock_dilation_sizes,
                                   upsample_rates, upsample_initial_channel, upsample_kernel_sizes,
                                   gen_istft_n_fft, gen_istft_hop_size, subbands, gin_channels)
        self._engine = None          # built lazily, rebuilt after the weights change
        self._engine_key = None

    # ---- samples per unit frame (320 for the shipped config)
    @property
    def samples_per_frame(self) -> int:
        c = self.model_config
        n = c["gen_istft_hop_size"] * c["subbands"]
        for u in c["upsample_rates"]:
            n *= u
        return n

    # ---- engine management
    def _weights_version(self):
        return tuple(p._version for p in self.parameters()) + (str(next(self.parameters()).device),)

    def engine(self):
        """The HIP engine for the current weights (packs them on first use / after a reload)."""
        from .engine import QvcEngine   # imports the ctypes binding; raises if the .so is missing
        key = self._weights_version()
        if self._engine is None or self._engine_key != key:
            device = next(self.parameters()).device
            if device.type != "cuda":
                raise RuntimeError("SynthesizerTrn hot path runs on an MI355X only: call .cuda() first "
                                   "(there is no CPU fallback)")
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            self._engine = QvcEngine(self.model_config, sd, device)
            self._engine_key = key
        return self._engine

    def forward(self, *args, **kwargs):
        raise NotImplementedError("training forward (models.py:593-623) is outside the inference hot path")

    # ---- reference API
    @torch.no_grad()
    def infer(self, unit: Tensor, mel: Tensor, noise: Optional[Tensor] = None) -> Tensor:
        """models.py:625-642.  unit (1,256,F), mel (1,80,F') -> (1,1,320*F) fp32.

        ``noise`` (1, inter, F) optionally replaces the internal N(0,1) draw of
        models.py:94 so a caller can reproduce a run.
        """
        g = self.speaker_embed(mel)                                               # models.py:635
        return self.infer_batch(unit, g, noise)

    @torch.no_grad()
    def posterior(self, spec: Tensor, g: Tensor, noise: Optional[Tensor] = None):
        """The analysis half of ``forward`` (models.py:617-618): z ~ enc_q(spec | g), z_p = flow(z, g).

        spec (B, spec_channels, T), g (B, gin) or (B, gin, 1), noise (B, inter, T) optional -> (z, z_p), both (B, inter, T).
        """
        eng = self.engine()
        g = g.reshape(g.shape[0], -1)
        if noise is None:
            noise = torch.randn(spec.shape[0], self.model_config["inter_channels"], spec.shape[2], device=spec.device)
        z = eng.enc_q(spec, g, noise)
        z_p = eng.flow_forward(z, g)
        return z.transpose(1, 2).contiguous(), z_p.transpose(1, 2).contiguous()

    @torch.no_grad()
    def speaker_embed(self, mel: Tensor) -> Tensor:
        """SpeakerEncoder.embed_utterance (models.py:528-546) for a batch: mel (U, 80, F') -> g (U, gin).

        Runs the persistent-LSTM HIP kernels (csrc/qvc_spk.hip).  Speaker widths the kernels do not cover
        (gin > 256 or gin % 8 != 0) raise ``QvcError`` (QVC_ERR_BAD_CONFIG from the library): there is no torch path.
        """
        return self.engine().speaker_embed(mel)

    @torch.no_grad()
    def infer_batch(self, unit: Tensor, g: Tensor, noise: Optional[Tensor] = None) -> Tensor:
        """Batched path: unit (B,256,F), g (B,gin) or (B,gin,1), noise (B,inter,F) -> (B,1,320*F)."""
        eng = self.engine()
        if noise is None:
            noise = torch.randn(unit.shape[0], self.model_config["inter_channels"], unit.shape[2],
                                device=unit.device, dtype=torch.float32)
        return eng.infer_batch(unit, g.reshape(g.shape[0], -1), noise)

    @torch.no_grad()
    def infer_ragged(self, units, g: Tensor, noises=None):
        """Utterances of DIFFERENT lengths in one launch sequence (the reference converts any length per call,
        convert.py:58-86): ``units`` is a list of (256, F_b) / (1, 256, F_b) tensors, g (B, gin) or (B, gin, 1),
        ``noises`` an optional list of (inter, F_b) draws.  Returns a list of (1, 320*F_b) fp32 waveforms -- each
        equal to the utterance converted alone (every conv sees ITS sequence end, not the padded batch's)."""
        eng = self.engine()
        dev = eng.device
        us = [u.reshape(-1, u.shape[-1]) for u in units]
        lens = [int(u.shape[-1]) for u in us]
        B, tmax, inter = len(us), max(lens), self.model_config["inter_channels"]
        unit = torch.zeros(B, us[0].shape[0], tmax, device=dev, dtype=torch.float32)
        noise = torch.zeros(B, inter, tmax, device=dev, dtype=torch.float32)
        for b, u in enumerate(us):
            unit[b, :, :lens[b]] = u.to(dev, torch.float32)
            n = noises[b].reshape(inter, -1).to(dev, torch.float32) if noises is not None else \
                torch.randn(inter, lens[b], device=dev, dtype=torch.float32)
            noise[b, :, :lens[b]] = n
        out = eng.infer_batch_ragged(unit, g.reshape(B, -1), noise, torch.tensor(lens, dtype=torch.int32))
        spf = self.samples_per_frame
        return [out[b, :, :lens[b] * spf] for b in range(B)]
