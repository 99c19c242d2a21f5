"""CPU oracle for the QuickVC inference hot path (TEST INFRASTRUCTURE, not product).

This file is a from-scratch fp32 restatement of the reference algorithm for the
path ``SynthesizerTrn.infer`` = enc_p -> reverse flow -> multi-stream iSTFT
decoder (SURVEY.md section 8a, rows E1..D9').  It is written with plain
``torch.nn.functional`` calls on CPU tensors, takes the reference's own
checkpoint ``state_dict`` (``weight_g`` / ``weight_v`` pairs) and never imports
anything from the reference.

Who may use it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- only as the checker / reported baseline.  The product
package never imports this module; the product path fails loudly when the HIP
library is missing.

Parity pinning: the reference has no tests or golden vectors,
so this oracle is pinned against outputs of the reference itself, generated in
the build container by ``tests/golden/make_golden.py`` (reference imported with
two import shims) and committed under ``tests/golden/``.  ``tests/test_oracle_golden.py``
checks every tap of those fixtures.

Every function cites the reference file:line it restates.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

LRELU_SLOPE = 0.1  # modules.py:11


# --------------------------------------------------------------------------- weights
def fold_weight_norm(v: Tensor, g: Tensor) -> Tensor:
    """w = g * v / ||v||, norm over every dim but 0 (old-style weight_norm, dim=0).

    Reference: ``weight_norm(...)`` at modules.py:54,64,67,134-143 and
    models.py:327,333,346,357; recomputed on every forward.  For
    ConvTranspose1d the tensor is (Cin, Cout, K), so the norm is per *input*
    channel -- the same "all dims but 0" rule covers it.
    """
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return v * (g / norm)


def conv_weight(sd: State, prefix: str) -> Tensor:
    """Effective weight of a (possibly weight-normed) conv stored under ``prefix``."""
    if prefix + ".weight_v" in sd:
        return fold_weight_norm(sd[prefix + ".weight_v"].float(), sd[prefix + ".weight_g"].float())
    return sd[prefix + ".weight"].float()


def conv_bias(sd: State, prefix: str) -> Optional[Tensor]:
    b = sd.get(prefix + ".bias")
    return None if b is None else b.float()


# --------------------------------------------------------------------------- WN
def wn_forward(sd: State, prefix: str, x: Tensor, g: Optional[Tensor], hidden: int,
               kernel_size: int, n_layers: int, taps: Optional[dict] = None) -> Tensor:
    """WaveNet stack, modules.py:69-114.

    Per layer: k-tap 'same' conv h->2h (dilation 1, modules.py:64), add the layer's
    slice of the conditioning (modules.py:84,94-98), tanh*sigmoid gate
    (modules.py:14-34), 1x1 h->2h (h on the last layer, modules.py:66-67); first
    half is the residual, second half (everything on the last layer) is summed
    into the output (modules.py:105-112).
    """
    out = torch.zeros_like(x)
    gc = None
    if g is not None:
        gc = F.conv1d(g, conv_weight(sd, f"{prefix}.cond_layer"), conv_bias(sd, f"{prefix}.cond_layer"))
    pad = (kernel_size - 1) // 2
    for i in range(n_layers):
        a = F.conv1d(x, conv_weight(sd, f"{prefix}.in_layers.{i}"), conv_bias(sd, f"{prefix}.in_layers.{i}"),
                     padding=pad)
        if gc is not None:
            a = a + gc[:, 2 * hidden * i: 2 * hidden * (i + 1)]
        acts = torch.tanh(a[:, :hidden]) * torch.sigmoid(a[:, hidden:])
        rs = F.conv1d(acts, conv_weight(sd, f"{prefix}.res_skip_layers.{i}"),
                      conv_bias(sd, f"{prefix}.res_skip_layers.{i}"))
        if i < n_layers - 1:
            x = x + rs[:, :hidden]
            out = out + rs[:, hidden:]
        else:
            out = out + rs
        if taps is not None:
            taps[f"{prefix}.layer{i}.out"] = out.clone()
            taps[f"{prefix}.layer{i}.x"] = x.clone()
    return out


# --------------------------------------------------------------------------- enc_p / enc_q
def cond_normal_wn(sd: State, prefix: str, series: Tensor, noise: Tensor, hidden: int, out_ch: int,
                   cond: Optional[Tensor] = None, taps: Optional[dict] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """CondNormalWN.forward, models.py:75-95 (kernel 5, 16 layers: models.py:582-583).

    ``noise`` replaces ``torch.randn_like(mu)`` (models.py:94) so both sides of a
    parity test see the same sample.
    """
    h = F.conv1d(series, conv_weight(sd, f"{prefix}.pre"), conv_bias(sd, f"{prefix}.pre"))
    if taps is not None:
        taps[f"{prefix}.pre"] = h.clone()
    h = wn_forward(sd, f"{prefix}.enc", h, cond, hidden, 5, 16, taps)
    stats = F.conv1d(h, conv_weight(sd, f"{prefix}.proj"), conv_bias(sd, f"{prefix}.proj"))
    mu, logs = stats[:, :out_ch], stats[:, out_ch:]
    z = mu + noise * torch.exp(logs)
    return z, mu, logs


# --------------------------------------------------------------------------- flow
def flow_reverse(sd: State, z_p: Tensor, g: Tensor, channels: int, hidden: int, n_flows: int = 4,
                 taps: Optional[dict] = None) -> Tensor:
    """ResidualCouplingBlock.forward(reverse=True), models.py:39-51.

    The module list is [L0, Flip, L1', Flip, ...] stored as flows.{0,2,4,6} = layers
    and odd indices = Flip (models.py:33-37); reversed iteration therefore does
    Flip then layer, for layers 6,4,2,0.  Layer (modules.py:199-224):
    x0,x1 = halves; m = post(WN(pre(x0), g)); x1 <- x1 - m.  Flip = channel
    reversal (modules.py:165-170).  WN: kernel 5, 4 layers (models.py:584).
    """
    half = channels // 2
    x = z_p
    for idx in reversed(range(n_flows)):
        x = torch.flip(x, [1])
        p = f"flow.flows.{2 * idx}"
        x0, x1 = x[:, :half], x[:, half:]
        h = F.conv1d(x0, conv_weight(sd, f"{p}.pre"), conv_bias(sd, f"{p}.pre"))
        h = wn_forward(sd, f"{p}.enc", h, g, hidden, 5, 4)
        m = F.conv1d(h, conv_weight(sd, f"{p}.post"), conv_bias(sd, f"{p}.post"))
        x = torch.cat([x0, x1 - m], 1)
        if taps is not None:
            taps[f"{p}.out"] = x.clone()
    return x


def flow_forward(sd: State, z: Tensor, g: Tensor, channels: int, hidden: int, n_flows: int = 4,
                 taps: Optional[dict] = None) -> Tensor:
    """ResidualCouplingBlock.forward(reverse=False), models.py:39-51 (the posterior direction, models.py:618):
    layer then Flip, for layers 0,2,4,6; the coupling adds: x1 <- m + x1 (modules.py:217)."""
    half = channels // 2
    x = z
    for idx in range(n_flows):
        p = f"flow.flows.{2 * idx}"
        x0, x1 = x[:, :half], x[:, half:]
        h = F.conv1d(x0, conv_weight(sd, f"{p}.pre"), conv_bias(sd, f"{p}.pre"))
        h = wn_forward(sd, f"{p}.enc", h, g, hidden, 5, 4)
        m = F.conv1d(h, conv_weight(sd, f"{p}.post"), conv_bias(sd, f"{p}.post"))
        x = torch.cat([x0, m + x1], 1)
        if taps is not None:
            taps[f"{p}.fwd"] = x.clone()
        x = torch.flip(x, [1])
    return x


def posterior_encode(sd: State, cfg: dict, spec: Tensor, g: Tensor, noise: Tensor,
                     taps: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """The posterior half of SynthesizerTrn.forward, models.py:617-618: z ~ enc_q(spec | g), z_p = flow(z, g).

    spec :: (B, spec_channels, T), g :: (B, gin, 1), noise :: (B, inter, T) -> (z, z_p)."""
    sd = {k: v.float() for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}
    inter, hidden = int(cfg["inter_channels"]), int(cfg["hidden_channels"])
    with torch.no_grad():
        z, mu, logs = cond_normal_wn(sd, "enc_q", spec.float(), noise.float(), hidden, inter, g.float(), None)
        if taps is not None:
            taps["enc_q.m"], taps["enc_q.logs"], taps["enc_q.z"] = mu.clone(), logs.clone(), z.clone()
        z_p = flow_forward(sd, z, g.float(), inter, hidden, 4, taps)
        if taps is not None:
            taps["flow.z_p"] = z_p.clone()
    return z, z_p


# --------------------------------------------------------------------------- decoder pieces
def resblock1(sd: State, prefix: str, x: Tensor, k: int, dilations: Sequence[int]) -> Tensor:
    """ResBlock1.forward, modules.py:147-154: 3x [lrelu -> dilated conv -> lrelu -> conv -> +x]."""
    for j, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, conv_weight(sd, f"{prefix}.convs1.{j}"), conv_bias(sd, f"{prefix}.convs1.{j}"),
                      padding=(k - 1) * d // 2, dilation=d)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, conv_weight(sd, f"{prefix}.convs2.{j}"), conv_bias(sd, f"{prefix}.convs2.{j}"),
                      padding=(k - 1) // 2)
        x = xt + x
    return x


def hann_periodic(n: int) -> Tensor:
    """torch.hann_window(n) (periodic): the window torchaudio's InverseSpectrogram registers."""
    i = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2.0 * math.pi * i / n)).float()


def istft_closed_form(spec_logmag: Tensor, phase_raw: Tensor, n_fft: int, hop: int) -> Tensor:
    """exp / pi*sin / polar / iSTFT of models.py:394-401 as explicit overlap-add.

    ``torchaudio.transforms.InverseSpectrogram(n_fft, n_fft, hop)`` (models.py:350) is
    ``torch.istft(center=True, window=hann_periodic(n_fft), normalized=False,
    onesided=True, length=None)``: per frame t an ``irfft`` of length n_fft (the
    imaginary parts of bins 0 and n_fft/2 are ignored), multiplied by the window,
    overlap-added at t*hop, divided by the overlap-added squared window, then
    n_fft/2 samples trimmed on both ends: output length hop*(frames-1).

    spec_logmag, phase_raw :: (N, n_fft/2+1, frames)  ->  (N, hop*(frames-1))
    """
    n_bins = n_fft // 2 + 1
    assert spec_logmag.shape[1] == n_bins
    n, _, frames = spec_logmag.shape
    mag = torch.exp(spec_logmag.double())
    ph = math.pi * torch.sin(phase_raw.double())
    re, im = mag * torch.cos(ph), mag * torch.sin(ph)
    # real inverse DFT written out: x[m] = (1/N) [ Re0 + (-1)^m Re_{N/2} + 2 sum_k (Re_k cos - Im_k sin) ]
    m = torch.arange(n_fft, dtype=torch.float64)
    k = torch.arange(n_bins, dtype=torch.float64)
    ang = 2.0 * math.pi * k[:, None] * m[None, :] / n_fft        # (bins, n_fft)
    wgt = torch.full((n_bins,), 2.0, dtype=torch.float64)
    wgt[0] = 1.0
    wgt[-1] = 1.0
    cosb = (wgt[:, None] * torch.cos(ang)) / n_fft
    sinb = (wgt[:, None] * torch.sin(ang)) / n_fft
    sinb[0] = 0.0
    sinb[-1] = 0.0
    frames_td = torch.einsum("nkt,km->ntm", re, cosb) - torch.einsum("nkt,km->ntm", im, sinb)  # (N, frames, n_fft)
    win = hann_periodic(n_fft).double()
    frames_td = frames_td * win
    total = hop * (frames - 1) + n_fft
    y = torch.zeros(n, total, dtype=torch.float64)
    env = torch.zeros(total, dtype=torch.float64)
    for t in range(frames):
        y[:, t * hop: t * hop + n_fft] += frames_td[:, t]
        env[t * hop: t * hop + n_fft] += win * win
    lo, hi = n_fft // 2, total - n_fft // 2
    return (y[:, lo:hi] / env[lo:hi]).float()


def pqmf_synthesis_filter(subbands: int = 4, taps: int = 62, cutoff_ratio: float = 0.15, beta: float = 9.0) -> Tensor:
    """Fixed cosine-modulated synthesis bank of pqmf.py:16-44,65-76 -> (1, subbands, taps+1).

    Prototype: windowed sinc with cut-off 0.15*pi times a Kaiser(beta 9) window
    (scipy ``kaiser(63, 9.0)`` == ``numpy.kaiser(63, 9.0)``); band k is
    2*h[n]*cos((2k+1)*pi/(2*subbands)*(n-(taps-1)/2) - (-1)^k*pi/4).
    """
    n = np.arange(taps + 1, dtype=np.float64)
    c = n - 0.5 * taps
    with np.errstate(invalid="ignore", divide="ignore"):
        proto = np.sin(np.pi * cutoff_ratio * c) / (np.pi * c)
    proto[taps // 2] = cutoff_ratio
    proto = proto * np.kaiser(taps + 1, beta)
    bank = np.zeros((subbands, taps + 1))
    for k in range(subbands):
        bank[k] = 2.0 * proto * np.cos((2 * k + 1) * (np.pi / (2 * subbands)) * (n - (taps - 1) / 2.0)
                                      - ((-1) ** k) * np.pi / 4.0)
    return torch.from_numpy(bank).float().unsqueeze(0)


def band_synthesis(y_mb: Tensor, fir: Tensor) -> Tensor:
    """Zero-stuff x subbands with gain subbands, then a (1, subbands, K) FIR with pad (K-1)/2.

    models.py:353-357,405-406 (learned filter) and pqmf.py:106-117 (fixed filter)
    have this same structure; ``updown_filter`` is delta[k==k', j==0].
    y_mb :: (B, subbands, L) -> (B, 1, subbands*L)
    """
    b, s, l = y_mb.shape
    up = torch.zeros(b, s, l * s, dtype=y_mb.dtype)
    up[:, :, ::s] = y_mb * s
    return F.conv1d(up, fir, padding=(fir.shape[-1] - 1) // 2)


def decoder_forward(sd: State, cfg: dict, z: Tensor, g: Tensor, taps: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """Multistream_/Multiband_iSTFT_Generator.forward, models.py:360-408 / 250-293.

    conv_pre(k7)+cond(g) -> per stage [lrelu(0.1) -> ConvTranspose1d -> mean of 3
    ResBlock1] -> lrelu(0.01) -> ReflectionPad1d((1,0)) -> subband_conv_post(k7) ->
    per-band iSTFT -> band synthesis.  ConvTranspose padding (k-u+1-i)//2 and
    output_padding 1-i for stage i: models.py:333-335.
    """
    p = "dec"
    subbands = int(cfg["subbands"])
    n_fft, hop = int(cfg["gen_istft_n_fft"]), int(cfg["gen_istft_hop_size"])
    n_bins = n_fft // 2 + 1
    ks, ds = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
    x = F.conv1d(z, conv_weight(sd, f"{p}.conv_pre"), conv_bias(sd, f"{p}.conv_pre"), padding=3)
    x = x + F.conv1d(g, conv_weight(sd, f"{p}.cond"), conv_bias(sd, f"{p}.cond"))
    if taps is not None:
        taps["dec.conv_pre"] = x.clone()
    for i, (u, ku) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, conv_weight(sd, f"{p}.ups.{i}"), conv_bias(sd, f"{p}.ups.{i}"),
                               stride=u, padding=(ku - u + 1 - i) // 2, output_padding=1 - i)
        if taps is not None:
            taps[f"dec.ups.{i}"] = x.clone()
        acc = None
        for j, (k, d) in enumerate(zip(ks, ds)):
            r = resblock1(sd, f"{p}.resblocks.{i * len(ks) + j}", x, k, d)
            if taps is not None:
                taps[f"dec.resblocks.{i * len(ks) + j}"] = r.clone()
            acc = r if acc is None else acc + r
        x = acc / len(ks)
        if taps is not None:
            taps[f"dec.mrf.{i}"] = x.clone()
    x = F.leaky_relu(x)                                   # default slope 0.01, models.py:385
    x = torch.cat([x[:, :, 1:2], x], dim=2)               # ReflectionPad1d((1, 0)), models.py:345,388
    x = F.conv1d(x, conv_weight(sd, f"{p}.subband_conv_post"), conv_bias(sd, f"{p}.subband_conv_post"), padding=3)
    if taps is not None:
        taps["dec.subband_conv_post"] = x.clone()
    b, _, frames = x.shape
    x = x.reshape(b, subbands, 2 * n_bins, frames)
    spec = x[:, :, :n_bins].reshape(b * subbands, n_bins, frames)
    phase = x[:, :, n_bins:].reshape(b * subbands, n_bins, frames)
    y_mb = istft_closed_form(spec, phase, n_fft, hop).reshape(b, subbands, -1)
    if cfg.get("mb_istft_vits"):
        fir = pqmf_synthesis_filter(subbands)             # pqmf.py:65-76
    else:
        fir = conv_weight(sd, f"{p}.multistream_conv_post")  # models.py:357
    y = band_synthesis(y_mb, fir)
    return y, y_mb


# --------------------------------------------------------------------------- speaker encoder
def speaker_embed_utterance(sd: State, mel_bt: Tensor, partial_frames: int = 128, partial_hop: int = 64) -> Tensor:
    """SpeakerEncoder.embed_utterance, models.py:528-546 + forward :514-518, batch 1.

    mel_bt :: (1, frames, 80).  Partials of 128 frames at hop 64 over
    range(0, frames-128, 64) plus the last 128 frames; each through the 3-layer
    LSTM -> Linear -> ReLU -> L2 normalise; the mean of the partial embeddings is
    *not* re-normalised (models.py:539-541).
    """
    hidden = sd["enc_spk.linear.weight"].shape[1]
    n_layers = len([k for k in sd if k.startswith("enc_spk.lstm.weight_ih_l")])
    lstm = torch.nn.LSTM(mel_bt.shape[-1], hidden, n_layers, batch_first=True)
    lstm.load_state_dict({k[len("enc_spk.lstm."):]: v for k, v in sd.items() if k.startswith("enc_spk.lstm.")})

    def embed(m: Tensor) -> Tensor:
        with torch.no_grad():
            _, (h, _) = lstm(m)
        e = F.relu(F.linear(h[-1], sd["enc_spk.linear.weight"], sd["enc_spk.linear.bias"]))
        return e / e.norm(dim=1, keepdim=True)

    frames = mel_bt.shape[1]
    last = mel_bt[:, -partial_frames:]
    if frames > partial_frames:
        parts = [mel_bt[0, s:s + partial_frames] for s in range(0, frames - partial_frames, partial_hop)]
        parts.append(last[0])
        return embed(torch.stack(parts, 0)).mean(dim=0, keepdim=True)
    return embed(last)


# --------------------------------------------------------------------------- whole path
def infer_from_g(sd: State, cfg: dict, unit: Tensor, g: Tensor, noise: Tensor,
                 taps: Optional[dict] = None) -> Tensor:
    """SynthesizerTrn.infer after the speaker encoder, models.py:638-642, batched.

    unit :: (B, 256, T), g :: (B, gin, 1), noise :: (B, inter, T) -> (B, 1, 320*T).
    """
    sd = {k: v.float() for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}
    inter, hidden = int(cfg["inter_channels"]), int(cfg["hidden_channels"])
    with torch.no_grad():
        z_p, mu, logs = cond_normal_wn(sd, "enc_p", unit.float(), noise.float(), hidden, inter, None, taps)
        if taps is not None:
            taps["enc_p.mu"], taps["enc_p.logs"], taps["enc_p.z_p"] = mu.clone(), logs.clone(), z_p.clone()
        z = flow_reverse(sd, z_p, g.float(), inter, hidden, 4, taps)
        o, y_mb = decoder_forward(sd, cfg, z, g.float(), taps)
        if taps is not None:
            taps["dec.y_mb"] = y_mb.clone()
    return o


def infer(sd: State, cfg: dict, unit: Tensor, mel: Tensor, noise: Tensor) -> Tensor:
    """SynthesizerTrn.infer, models.py:625-642 (batch 1; mel :: (1, 80, frames))."""
    sdf = {k: v.float() for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}
    g = speaker_embed_utterance(sdf, mel.transpose(1, 2).float()).unsqueeze(-1)
    return infer_from_g(sd, cfg, unit, g, noise)


# --------------------------------------------------------------------------- mel front-end
def wave_to_spec(wave: Tensor, n_fft: int, hop: int, win: int) -> Tensor:
    """mel_processing.py:15-58: reflect pad (n_fft-hop)/2 per side (:46), Hann STFT center=False (:50-51),
    sqrt(re^2 + im^2 + 1e-6) (:54).  wave :: (B, T) -> (B, n_fft/2+1, frames)."""
    pad = int((n_fft - hop) / 2)
    x = F.pad(wave.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    window = torch.hann_window(win, dtype=wave.dtype)
    spec = torch.stft(x, n_fft, hop_length=hop, win_length=win, window=window, center=False, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    return torch.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-6)


def wave_to_mel(wave: Tensor, mel_basis: Tensor, n_fft: int, hop: int, win: int) -> Tensor:
    """mel_processing.py:61-98: log(clamp(mel_basis @ spec, 1e-5)) (:8, :74-75).  The filter bank is an argument:
    the reference takes it from librosa.filters.mel (:69), which is absent here -- PARITY UNPINNED for the bank."""
    return torch.log(torch.clamp(torch.matmul(mel_basis.to(wave.dtype), wave_to_spec(wave, n_fft, hop, win)), min=1e-5))


def snr_db(ref: Tensor, out: Tensor) -> float:
    """10 log10( sum ref^2 / sum (ref-out)^2 ) -- the parity metric of SURVEY 8d."""
    ref, out = ref.double().flatten(), out.double().flatten()
    err = torch.sum((ref - out) ** 2).item()
    sig = torch.sum(ref ** 2).item()
    return float("inf") if err == 0 else 10.0 * math.log10(sig / err)
