"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle and against the golden vectors recorded from the reference.

Tolerances (floating point, stated here as the task requires):
  * one conv through the MFMA kernel vs the same conv in fp64 on operands rounded to the MFMA
    operand type: max abs error <= 2e-5 * (1 + |y|max)  (fp32 accumulation order only);
  * whole path, f16 operands: waveform SNR >= 45 dB per utterance (target of BASELINE.json: 40 dB;
    measured ~52.6 dB); bf16 operands: >= 30 dB (measured ~34.5 dB -- documented as NOT meeting
    the 40 dB bar, which is why f16 is the default operand type);
  * fp32 tail (iSTFT + synthesis FIR): SNR >= 100 dB.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import qvc_oracle as oracle
from helpers import load_case, regenerate, snr_db

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lib():
    from quickvc_official_amd import lib as L
    l = L.load_library()                       # raises if the HIP library is missing: no fallback
    assert l.qvc_device_check() == 0
    return l


def _engine(entry, sd, dev, dtype):
    import quickvc_official_amd as q
    from quickvc_official_amd.engine import QvcEngine
    model = q.SynthesizerTrn(641, 32, **entry["config"])
    return QvcEngine(dict(model.model_config, operand_dtype=dtype), sd, dev)


def _fm(t):
    return t.transpose(1, 2).contiguous()


CONV_CASES = [
    # B, Cin, Cout, T, k, dil, slope
    (1, 32, 64, 40, 1, 1, 1.0),
    (2, 64, 64, 37, 3, 1, 0.1),
    (2, 128, 128, 300, 11, 5, 0.1),
    (2, 256, 256, 200, 7, 3, 0.1),
    (1, 192, 384, 250, 5, 1, 1.0),
    (2, 128, 72, 130, 7, 1, 0.01),
    (3, 40, 80, 21, 5, 1, 1.0),
    (1, 512, 128, 1, 1, 1, 1.0),      # single frame
    (1, 8, 4, 700, 3, 1, 0.1),        # smallest legal channel counts, many tiles
]


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_kernel_vs_rounded_reference(lib, dev, case, dtype):
    from quickvc_official_amd import lib as L
    B, cin, cout, T, k, dil, slope = case
    gen = torch.Generator().manual_seed(cin * 7 + cout + k + dil)
    x = torch.randn(B, cin, T, generator=gen)
    w = torch.randn(cout, cin, k, generator=gen) / (cin * k) ** 0.5
    bias = torch.randn(cout, generator=gen) * 0.1
    nb = int(lib.qvc_conv1d_scratch_bytes(cout, cin, k))
    nw = int(lib.qvc_conv1d_workspace_bytes(B, cout, cin, T))
    assert nb > 0 and nw > 0
    sh = torch.empty(nb + 256, dtype=torch.uint8)
    sd_ = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
    ws = torch.empty(nw + 256, dtype=torch.uint8, device=dev)
    al = lambda t: t.data_ptr() + ((-t.data_ptr()) % 256)
    xd = x.to(dev)
    y = torch.full((B, cout, T), float("nan"), device=dev)
    st = lib.qvc_conv1d(xd.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, cin, cout, T, k, dil, slope,
                        L.DTYPES[dtype], al(sh), al(sd_), nb, al(ws), nw, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert st == 0
    td = torch.float16 if dtype == "f16" else torch.bfloat16
    xr = F.leaky_relu(x, slope).to(td).double()
    wr = w.to(td).double()
    ref = F.conv1d(xr, wr, bias.double(), padding=(k - 1) // 2 * dil, dilation=dil).float()
    err = float((ref - y.cpu()).abs().max())
    assert err <= 2e-5 * (1.0 + float(ref.abs().max())), err


def test_conv_entry_point_rejects_bad_arguments(lib, dev):
    assert lib.qvc_conv1d_scratch_bytes(10, 16, 3) == -1          # Cout % 4
    assert lib.qvc_conv1d_scratch_bytes(16, 12, 3) == -1          # Cin % 8
    assert lib.qvc_conv1d(None, None, None, None, 1, 8, 4, 8, 3, 1, 1.0, 1, None, None, 0, None, 0, None) == -1


@pytest.mark.parametrize("name,dtype,min_db", [
    ("mini", "f16", 45.0), ("odd", "f16", 45.0), ("mini_t37", "f16", 45.0), ("mini_mb", "f16", 45.0),
    ("mini", "bf16", 30.0), ("odd", "bf16", 30.0),
])
def test_stages_and_whole_path_vs_oracle_and_golden(lib, dev, name, dtype, min_db):
    entry, gold = load_case(name)
    _m, sd, unit, g, noise = regenerate(entry)
    taps = {}
    ref = oracle.infer_from_g(sd, entry["config"], unit, g.unsqueeze(-1), noise, taps)
    eng = _engine(entry, sd, dev, dtype)
    # stage by stage, each fed with the oracle's input for that stage
    z_p = eng.enc_p(unit, noise)
    assert snr_db(_fm(taps["enc_p.z_p"]), z_p.cpu()) >= min_db + 5
    z = eng.flow_reverse(_fm(taps["enc_p.z_p"]), g)
    assert snr_db(_fm(taps["flow.flows.0.out"]), z.cpu()) >= min_db + 5
    post = eng.dec_trunk(_fm(taps["flow.flows.0.out"]), g)
    assert snr_db(_fm(taps["dec.subband_conv_post"]), post.cpu()) >= min_db + 5
    out, ymb = eng.istft_synth(_fm(taps["dec.subband_conv_post"]), want_bands=True)
    assert snr_db(ref, out.cpu()) >= 100.0                                   # fp32 tail
    assert snr_db(taps["dec.y_mb"], ymb.cpu()) >= 100.0
    # whole path vs oracle and vs the reference's own output
    full = eng.infer_batch(unit.to(dev), g.to(dev), noise.to(dev))
    torch.cuda.synchronize()
    assert full.shape == (entry["batch"], 1, 320 * entry["frames"])         # length rule
    for b in range(entry["batch"]):
        assert snr_db(ref[b], full[b].cpu()) >= min_db
    assert snr_db(gold["o"], full.cpu().reshape(-1).numpy()) >= min_db


@pytest.mark.parametrize("name", ["full_b1", "full_b2"])
def test_full_config_vs_reference_golden(lib, dev, name):
    """Shipped config, T=250: output vs the waveform the reference produced (>= 45 dB, f16)."""
    entry, gold = load_case(name)
    _m, sd, unit, g, noise = regenerate(entry)
    eng = _engine(entry, sd, dev, "f16")
    full = eng.infer_batch(unit.to(dev), g.to(dev), noise.to(dev))
    torch.cuda.synchronize()
    want = gold["o"].reshape(entry["batch"], -1)
    for b in range(entry["batch"]):
        assert snr_db(want[b], full[b].cpu().reshape(-1).numpy()) >= 45.0
    assert float(np.abs(want - full.cpu().reshape(entry["batch"], -1).numpy()).max()) < 0.05


def test_batch32_properties_at_benchmark_size(lib, dev):
    """BASELINE size (B=32, T=250): size-independent properties instead of a full oracle run.
    (1) batch independence: utterance b of the batch == the same utterance converted alone
        (different tiles are picked for B=1, the K order is the same -> bit-identical);
    (2) determinism across two runs; (3) finite output of the right length;
    (4) three utterances spot-checked against the oracle."""
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    _m, sd, _u, _g, _n = regenerate(entry)
    eng = _engine(entry, sd, dev, "f16")
    unit, g, noise = make_synthetic_inputs(32, 250, 256, 192, 256, seed0=100)
    ud, gd, nd = unit.to(dev), g.to(dev), noise.to(dev)
    out = eng.infer_batch(ud, gd, nd).clone()
    out2 = eng.infer_batch(ud, gd, nd).clone()
    torch.cuda.synchronize()
    assert out.shape == (32, 1, 80000) and bool(torch.isfinite(out).all())
    assert torch.equal(out, out2)
    for b in (0, 17, 31):
        alone = eng.infer_batch(ud[b:b + 1], gd[b:b + 1], nd[b:b + 1])
        torch.cuda.synchronize()
        assert snr_db(out[b].cpu(), alone[0].cpu()) >= 100.0
        ref = oracle.infer_from_g(sd, entry["config"], unit[b:b + 1], g[b:b + 1].unsqueeze(-1), noise[b:b + 1])
        assert snr_db(ref[0], out[b].cpu()) >= 45.0


def test_infer_api_matches_reference_semantics(lib, dev):
    """SynthesizerTrn.infer(unit, mel): speaker encoder (PyTorch) + HIP path, vs the golden infer() output."""
    import os
    import quickvc_official_amd as q
    from quickvc_official_amd.synth import make_synthetic_mel
    import helpers
    entry, _ = load_case("mini")
    spk = dict(np.load(os.path.join(helpers.GOLDEN, "mini_spk.npz")))
    model, sd, unit, g, noise = regenerate(entry)
    assert model.model_config["operand_dtype"] == "f16"        # the default must be the one that meets 40 dB
    model.load_state_dict(sd)
    model = model.cuda().eval()
    mel = make_synthetic_mel(300, 80, seed=307)
    o = model.infer(unit[:1].cuda(), mel.cuda(), noise=noise[:1].cuda())
    torch.cuda.synchronize()
    assert o.shape == (1, 1, 320 * entry["frames"]) and o.dtype == torch.float32
    assert snr_db(spk["infer_o"], o.cpu().numpy()) >= 45.0
    # a weight reload must invalidate the packed blob
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["dec.multistream_conv_post.weight_g"] = sd2["dec.multistream_conv_post.weight_g"] * 2.0
    model.load_state_dict(sd2)
    o2 = model.infer(unit[:1].cuda(), mel.cuda(), noise=noise[:1].cuda())
    assert snr_db(2.0 * spk["infer_o"], o2.cpu().numpy()) >= 45.0


def test_wn_stack_kernel_equals_per_layer_kernels(lib, dev, monkeypatch):
    """The whole-stack WaveNet kernel (4 layers per launch, overlap-tiled, coupling pre/post fused) against the
    path for configurations it does not cover -- one fused launch per layer, pre / post as separate convs --
    selected with QVC_WN_CHUNK=-1.  Same K order per frame -> identical results."""
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    _m, sd, _u, _g, _n = regenerate(entry)
    eng = _engine(entry, sd, dev, "f16")
    unit, g, noise = make_synthetic_inputs(6, 250, 256, 192, 256, seed0=300)
    z_stack = eng.enc_p(unit, noise)
    zf_stack = eng.flow_reverse(z_stack, g)
    _, recs = eng.infer_batch_timed(unit.to(dev), g.to(dev), noise.to(dev))
    assert sum(r["name"].startswith("wn_stack<f16,W12,L4") for r in recs) == 8      # enc_p: 4 launches of 4 layers; 4 flows
    monkeypatch.setenv("QVC_WN_CHUNK", "-1")
    z_layer = eng.enc_p(unit, noise)
    zf_layer = eng.flow_reverse(z_stack, g)
    _, recs = eng.infer_batch_timed(unit.to(dev), g.to(dev), noise.to(dev))
    torch.cuda.synchronize()
    names = [r["name"] for r in recs]
    assert sum(n.startswith("wn_layer<f16,W12") for n in names) == 32 and not any(n.startswith("wn_stack<") for n in names)
    assert sum(n.startswith("conv<") for n in names) == 2 + 4 * 2 + 1 + 2 + 1      # + 4 x (pre, post)
    assert snr_db(z_layer.cpu(), z_stack.cpu()) >= 100.0
    assert snr_db(zf_layer.cpu(), zf_stack.cpu()) >= 100.0


def test_hipgraph_capture_replays_identically(lib, dev):
    entry, _ = load_case("mini_t37")
    _m, sd, unit, g, noise = regenerate(entry)
    eng = _engine(entry, sd, dev, "f16")
    ud, gd, nd = unit.to(dev), g.to(dev), noise.to(dev)
    out = torch.empty(entry["batch"], 1, 320 * entry["frames"], device=dev)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        eager = eng.infer_batch(ud, gd, nd).clone()
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            eng.infer_batch(ud, gd, nd, out)
        out.zero_()
        graph.replay()
        s.synchronize()
    assert torch.equal(out, eager)


def test_timed_variant_reports_every_launch(lib, dev):
    entry, _ = load_case("mini")
    _m, sd, unit, g, noise = regenerate(entry)
    eng = _engine(entry, sd, dev, "f16")
    out, recs = eng.infer_batch_timed(unit.to(dev), g.to(dev), noise.to(dev))
    ref = eng.infer_batch(unit.to(dev), g.to(dev), noise.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    names = [r["name"] for r in recs]
    assert names.count("istft_synth") == 1 and names.count("cond_gemv") == 1
    # enc: pre + proj; dec: conv_pre + 2 ups + conv_post (the coupling layers' pre / post ride in their stack launch);
    # enc_p WaveNet: 4 launches of 4 layers; 4 coupling stacks; 2 x 9 fused ResBlock pairs
    assert sum(n.startswith("conv<") for n in names) == 2 + 1 + 2 + 1
    assert sum(n.startswith("wn_stack<") for n in names) == 8 and sum(n.startswith("wn_layer<") for n in names) == 0
    assert sum(n.startswith("rbpair") for n in names) in (6, 12, 18)   # per stage: 3 launches of three chains, or 9 of one
    assert all(r["ms"] >= 0 for r in recs) and sum(r["flops"] for r in recs) > 0


def test_chunked_streaming_is_exact(lib, dev):
    """BASELINE configs[4]: fixed-shape windows (hop + 2*88 frames of context), graph-replayed, several
    concurrent streams.  Property: the concatenated chunks equal the whole-utterance conversion."""
    import quickvc_official_amd as q
    from quickvc_official_amd.streaming import ChunkedConverter
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    model, sd, _u, _g, _n = regenerate(entry)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    S, T, hop = 3, 700, 160
    unit, g, noise = make_synthetic_inputs(S, T, 256, 192, 256, seed0=40)
    whole = model.infer_batch(unit.cuda(), g.cuda(), noise.cuda())
    conv = ChunkedConverter(model, streams=S, hop_frames=hop, context=88, use_graph=True)
    assert [w[0] for w in conv.windows(T)] == [0, 72, 232, 364, 364]          # clamped at both sequence edges
    chunked = conv.convert(unit, g, noise)
    torch.cuda.synchronize()
    assert chunked.shape == whole.shape == (S, 1, 320 * T)
    assert snr_db(whole.cpu(), chunked.cpu()) >= 90.0
    with pytest.raises(ValueError):
        ChunkedConverter(model, streams=S, hop_frames=hop, context=40)


def test_wide_config_takes_the_fallback_paths(lib, dev):
    """A config the fused pair kernel does not cover: stage-1 ResBlocks 416 channels wide (two M chunks -> the
    conv1 / conv2 launches with an operand-type residual instead of the fused pair); WaveNet width 256 = the
    widest the WaveNet kernels take (16 waves per workgroup).  Checked against the CPU oracle only (no golden:
    the reference was not run on this config)."""
    import quickvc_official_amd as q
    from quickvc_official_amd.engine import QvcEngine
    from quickvc_official_amd.synth import make_synthetic_state_dict, make_synthetic_inputs
    cfg = dict(q.DEFAULT_MODEL_CONFIG, inter_channels=64, hidden_channels=256, upsample_initial_channel=832, gin_channels=32)
    model = q.SynthesizerTrn(641, 32, **cfg)
    sd = make_synthetic_state_dict(model, 77)
    unit, g, noise = make_synthetic_inputs(2, 20, 256, 64, 32, seed0=5)
    ref = oracle.infer_from_g(sd, cfg, unit, g.unsqueeze(-1), noise)
    eng = QvcEngine(model.model_config, sd, dev)
    out, recs = eng.infer_batch_timed(unit.to(dev), g.to(dev), noise.to(dev))
    torch.cuda.synchronize()
    names = [r["name"] for r in recs]
    assert sum(n.startswith("rbpair") for n in names) in (3, 9)         # stage 2 (208 channels) still fuses
    assert sum(n.startswith("wn_stack<f16,W16") for n in names) == 8
    assert sum(n.startswith("conv<") for n in names) == 2 + 1 + 2 + 1 + 18   # + 9 x (conv1, conv2) of stage 1
    for b in range(2):
        assert snr_db(ref[b], out[b].cpu()) >= 45.0


# ------------------------------------------------------------------ speaker encoder
def _g_err(ref, got):
    ref, got = np.asarray(ref, np.float64), np.asarray(got, np.float64)
    return float(np.sqrt(((ref - got) ** 2).sum(-1) / (ref ** 2).sum(-1)).max())


@pytest.mark.parametrize("dtype,tol", [("f16", 1e-3), ("bf16", 1e-2)])
def test_speaker_embed_matches_oracle(lib, dev, dtype, tol):
    """qvc_speaker_embed (persistent-LSTM kernels) vs SpeakerEncoder.embed_utterance restated in the oracle
    (models.py:507-546; the oracle itself is pinned by golden/mini_spk.npz): full-width encoder (gin 256),
    mel lengths on both sides of the 128-frame branch and of the 64-frame hop, batches of utterances.
    Tolerance: relative L2 error of the embedding, operands rounded to the MFMA type every step."""
    from quickvc_official_amd.synth import make_synthetic_mel
    entry, _ = load_case("full_b1")
    _m, sd, _u, _g, _n = regenerate(entry)
    sdf = {k: v.float() for k, v in sd.items()}
    eng = _engine(entry, sd, dev, dtype)
    for frames, U in ((1, 1), (100, 2), (128, 1), (129, 3), (250, 32), (300, 2), (321, 1)):
        mel = torch.cat([make_synthetic_mel(frames, 80, seed=11 * frames + u) for u in range(U)], 0)
        g = eng.speaker_embed(mel.to(dev))
        torch.cuda.synchronize()
        assert g.shape == (U, 256) and g.dtype == torch.float32
        ref = torch.cat([oracle.speaker_embed_utterance(sdf, mel[u:u + 1].transpose(1, 2)) for u in range(U)], 0)
        assert _g_err(ref.numpy(), g.cpu().numpy()) <= tol, (frames, U)


def test_speaker_embed_golden_and_narrow_widths(lib, dev):
    """Reference-generated embeddings (golden/mini_spk.npz, gin 64: two waves per workgroup) and a width that
    is not a multiple of 32 (gin 24: padded hidden units must stay zero)."""
    import os
    import helpers
    from quickvc_official_amd.synth import make_synthetic_mel
    entry, _ = load_case("mini")
    spk = dict(np.load(os.path.join(helpers.GOLDEN, "mini_spk.npz")))
    _m, sd, _u, _g, _n = regenerate(entry)
    eng = _engine(entry, sd, dev, "f16")
    for frames in (100, 128, 300):
        mel = make_synthetic_mel(frames, 80, seed=7 + frames)
        g = eng.speaker_embed(mel.to(dev))
        assert _g_err(spk[f"g_{frames}"], g.cpu().numpy()) <= 1e-3, frames
    entry, _ = load_case("odd")
    _m, sd, _u, _g, _n = regenerate(entry)
    sdf = {k: v.float() for k, v in sd.items()}
    eng = _engine(entry, sd, dev, "f16")
    mel = torch.cat([make_synthetic_mel(200, 80, seed=90 + u) for u in range(5)], 0)
    g = eng.speaker_embed(mel.to(dev))
    ref = torch.cat([oracle.speaker_embed_utterance(sdf, mel[u:u + 1].transpose(1, 2)) for u in range(5)], 0)
    assert _g_err(ref.numpy(), g.cpu().numpy()) <= 4e-3


def test_speaker_embed_bad_args(lib, dev):
    import ctypes
    from quickvc_official_amd import lib as L
    entry, _ = load_case("mini")
    _m, sd, _u, _g, _n = regenerate(entry)
    eng = _engine(entry, sd, dev, "f16")
    with pytest.raises(ValueError):
        eng.speaker_embed(torch.zeros(1, 79, 50, device=dev))
    mel = torch.zeros(1, 80, 50, device=dev)
    eng.speaker_embed(mel)
    g = torch.empty(1, 64, device=dev)
    st = eng.lib.qvc_speaker_embed(ctypes.byref(eng.cfg), eng._spk_blob.data_ptr(), mel.data_ptr(), g.data_ptr(), 1, 50,
                                   eng._spk_ws.data_ptr(), 16, None)
    assert st == -5      # QVC_ERR_SMALL_BUFFER
    st = eng.lib.qvc_speaker_embed(ctypes.byref(eng.cfg), eng._spk_blob.data_ptr(), None, g.data_ptr(), 1, 50,
                                   eng._spk_ws.data_ptr(), eng._spk_ws.numel(), None)
    assert st == -1      # QVC_ERR_BAD_ARG


# ------------------------------------------------------------------ mel front-end
@pytest.mark.parametrize("samples,U", [(80000, 2), (16000 * 3 + 123, 1), (2000, 3), (641, 1)])
def test_wave_to_mel_matches_torch_restatement(lib, dev, samples, U):
    """qvc_wave_to_mel (fp32 MFMA STFT + sparse mel + log) vs the torch restatement of mel_processing.py:15-98
    in frontend.wave_to_mel run on the CPU (torch.stft, fp32), shipped front-end parameters (1280 / 320 / 80 mel).
    Lengths: a multiple of the hop, a ragged one, one shorter than a window, the shortest the reflect pad allows.
    Tolerance: |log-mel difference| <= 2e-5 (fp32 summation order of a 1280-term DFT vs the FFT's butterflies;
    the clamp at 1e-5 bounds the log's sensitivity; measured 1.4e-6).  The mel filter bank itself is parity-unpinned (no librosa)."""
    from quickvc_official_amd.frontend import MelFrontend, wave_to_mel
    gen = torch.Generator().manual_seed(samples + U)
    t = torch.arange(samples) / 16000.0
    wave = 0.3 * torch.sin(2 * np.pi * 220.0 * t)[None] * torch.rand(U, 1, generator=gen) + 0.05 * torch.randn(U, samples, generator=gen)
    wave = wave.clamp(-1, 1)
    fe = MelFrontend(1280, 80, 16000, 320, 1280, 0.0, None, device=dev)
    mel = fe(wave.to(dev))
    torch.cuda.synchronize()
    ref = wave_to_mel(wave, 1280, 80, 16000, 320, 1280, 0.0, None)
    assert mel.shape == ref.shape == (U, 80, fe.frames(samples)) and mel.dtype == torch.float32
    assert float((mel.cpu() - ref).abs().max()) <= 2e-5


def test_convert_cli_end_to_end(lib, dev, tmp_path):
    """BASELINE configs[0] plumbing on the GPU: the reference CLI surface (convert.py:19-86) from files to files --
    JSON config, checkpoint in the reference's format, `title|src|tgt` list, target wav -> trim -> HIP mel ->
    HIP speaker encoder, source units from .npy -> HIP path -> float32 wav of 320 samples per unit frame.
    The written waveform must equal what the Python API computes from the same files and seed."""
    import json
    from scipy.io import wavfile
    import quickvc_official_amd as q
    from quickvc_official_amd import convert as cli
    from quickvc_official_amd.checkpoint import save_checkpoint
    from quickvc_official_amd.frontend import MelFrontend, load_wav, trim
    from quickvc_official_amd.synth import make_synthetic_state_dict
    cfg = {"train": {"segment_size": 10240}, "data": dict(q.DEFAULT_DATA_CONFIG), "model": dict(q.MINI_MODEL_CONFIG)}
    hp = tmp_path / "config.json"
    hp.write_text(json.dumps(cfg))
    model = q.SynthesizerTrn(641, 32, **q.MINI_MODEL_CONFIG)
    sd = make_synthetic_state_dict(model, 21)
    model.load_state_dict(sd)
    pt = tmp_path / "G_1.pth"
    save_checkpoint(model, None, 2e-4, 1, str(pt))
    sr = cfg["data"]["sampling_rate"]
    t = np.arange(int(1.7 * sr)) / sr
    tgt = (0.4 * np.sin(2 * np.pi * 180 * t) * (t > 0.2) * (t < 1.5)).astype(np.float32)      # silence to trim on both sides
    wavfile.write(str(tmp_path / "tgt.wav"), sr, (tgt * 32767).astype(np.int16))
    rng = np.random.RandomState(5)
    for name, frames in (("a", 81), ("b", 81), ("c", 40)):
        np.save(str(tmp_path / f"{name}.npy"), rng.randn(frames, 256).astype(np.float32))
    (tmp_path / "convert.txt").write_text("".join(f"t_{n}|{tmp_path}/{n}.npy|{tmp_path}/tgt.wav\n" for n in "abc"))
    out = tmp_path / "out"
    cli.main(["--hpfile", str(hp), "--ptfile", str(pt), "--txtpath", str(tmp_path / "convert.txt"), "--outdir", str(out),
              "--seed", "7", "--batch", "2"])
    d = cfg["data"]
    wav = torch.from_numpy(trim(load_wav(str(tmp_path / "tgt.wav"), sr), top_db=20)).unsqueeze(0)
    assert wav.shape[1] < len(tgt)                                   # the silence was trimmed
    mel = MelFrontend(d["filter_length"], d["n_mel_channels"], sr, d["hop_length"], d["win_length"], d["mel_fmin"], d["mel_fmax"])(wav)
    net = model.cuda().eval()
    g = net.speaker_embed(mel)
    for name, frames in (("a", 81), ("b", 81), ("c", 40)):
        rate, got = wavfile.read(str(out / f"t_{name}.wav"))
        assert rate == sr and got.dtype == np.float32 and got.shape == (320 * frames,)     # convert.py:84-86
        assert np.isfinite(got).all() and np.abs(got).max() > 0
    # same seed, same batching (longest first: a and b share the first launch) -> identical noise draw
    _, got_a = wavfile.read(str(out / "t_a.wav"))
    matches = []
    for order in ("ab", "ba"):
        torch.manual_seed(7)
        unit = torch.stack([torch.from_numpy(np.load(str(tmp_path / f"{n}.npy"))).t() for n in order], 0).cuda()
        ref = net.infer_batch(unit, g.expand(2, -1))
        matches.append(np.array_equal(got_a, ref[order.index("a"), 0].cpu().numpy()))
    assert any(matches)


# ------------------------------------------------------------------ posterior direction
@pytest.mark.parametrize("name", ["mini_q", "odd_q"])
def test_posterior_direction_vs_reference_golden(lib, dev, name):
    """qvc_enc_q + qvc_flow_forward (models.py:617-618) vs what the reference produced (tests/golden/*_q.npz,
    recorded by make_golden_q.py): z ~ enc_q(spec | g) and z_p = flow(z, g), f16 operands, >= 45 dB."""
    import json
    import os
    import helpers
    import quickvc_official_amd as q
    from quickvc_official_amd.synth import make_synthetic_state_dict, make_synthetic_posterior_inputs
    entry = json.load(open(os.path.join(helpers.GOLDEN, "manifest.json")))[name]
    gold = dict(np.load(os.path.join(helpers.GOLDEN, entry["file"])))
    cfg = entry["config"]
    model = q.SynthesizerTrn(641, 32, **cfg)
    sd = make_synthetic_state_dict(model, entry["weights_seed"])
    model.load_state_dict(sd)
    model = model.cuda().eval()
    spec, g, noise = make_synthetic_posterior_inputs(entry["batch"], entry["frames"], 641, cfg["inter_channels"], cfg["gin_channels"],
                                                     seed0=entry["inputs_seed0"])
    z, z_p = model.posterior(spec.to(dev), g.to(dev), noise.to(dev))
    torch.cuda.synchronize()
    assert z.shape == z_p.shape == (entry["batch"], cfg["inter_channels"], entry["frames"])
    assert snr_db(gold["enc_q.z"], z.cpu().numpy()) >= 45.0
    assert snr_db(gold["flow.z_p"], z_p.cpu().numpy()) >= 45.0


def test_forward_flow_inverts_reverse_flow_at_benchmark_size(lib, dev):
    """Size-independent property at B=32, T=250 (shipped config): flow(reverse) o flow(forward) = identity.  Both
    directions compute m from the untouched half with the same kernels, so one coupling layer inverts to one fp32
    add + subtract per element; across the four layers that 1e-7 perturbation of the next layer's input now and
    then flips an f16 operand rounding, which bounds the round trip at ~70 dB (asserted >= 60).  The fallback path
    (QVC_WN_CHUNK=-1: unfused pre/post) must agree with the fused one."""
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    _m, sd, _u, _g, _n = regenerate(entry)
    eng = _engine(entry, sd, dev, "f16")
    _unit, g, noise = make_synthetic_inputs(32, 250, 256, 192, 256, seed0=700)
    z = noise.transpose(1, 2).contiguous().to(dev)                    # any latent will do
    z_p = eng.flow_forward(z, g)
    back = eng.flow_reverse(z_p, g)
    torch.cuda.synchronize()
    assert snr_db(z.cpu(), z_p.cpu()) < 40.0                          # the flow did something
    assert snr_db(z.cpu(), back.cpu()) >= 60.0
    import os
    os.environ["QVC_WN_CHUNK"] = "-1"
    try:
        z_p2 = eng.flow_forward(z[:3], g[:3])
    finally:
        del os.environ["QVC_WN_CHUNK"]
    torch.cuda.synchronize()
    assert snr_db(z_p[:3].cpu(), z_p2.cpu()) >= 100.0
