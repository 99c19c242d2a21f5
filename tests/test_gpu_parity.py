"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle and against the golden vectors recorded from the reference.

Tolerances (floating point, stated here as the task requires):
  * one conv through the MFMA kernel vs the same conv in fp64 on operands rounded to the MFMA
    operand type: max abs error <= 2e-5 * (1 + |y|max)  (fp32 accumulation order only);
  * whole path, waveform SNR per utterance against the fp32 oracle / the reference's goldens
    (BASELINE.json's bar: 40 dB): f16 operands >= 45 dB (measured 52-53 dB); the mixed mode bf16x
    (bf16 MFMA operands in the fused ResBlock pairs, f16 residual stream, f16 elsewhere -- the mode
    bench.py runs) >= 40 dB on every configuration tested (shipped config 45.6 dB); all-bf16 >= 30 dB
    (measured 34 dB: documented as NOT meeting the bar);
  * fp32 tail (iSTFT + synthesis FIR): SNR >= 100 dB;
  * launch-shape / kernel-selection variants (library debug switches, flipped in-process through
    qvc_debug_set): bit-identical, or >= 100 dB where the K order differs.
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


DEBUG_DEFAULTS = {"post_tail": 1, "post_tail_nf": 4, "pair_wide_launch": 1, "pair_cm4": 1, "conv_cl": 1, "wn_chunk": 0,
                  "pair_chain3": 0, "wn_kernel": 0}


@pytest.fixture(autouse=True)
def _debug_switches_at_defaults():
    """Every test starts and ends with the library's developer switches at their production values."""
    from quickvc_official_amd import lib as L
    for k, v in DEBUG_DEFAULTS.items():
        L.debug_set(k, v)
    yield
    for k, v in DEBUG_DEFAULTS.items():
        L.debug_set(k, v)


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


def test_full_width_stage_parity(lib, dev):
    """Stage-level parity on the SHIPPED config (192 / 512 channels, T = 250): every stage entry point is fed with
    the oracle's input for that stage and compared with the oracle's output (full tensors, CPU run of one
    utterance) AND with the taps the reference itself produced (tests/golden/full_b1.npz, subsampled with the
    stride recorded at generation time).  A mid-path error that cancels at the waveform would show here."""
    from helpers import subsample
    entry, gold = load_case("full_b1")
    _m, sd, unit, g, noise = regenerate(entry)
    taps = {}
    ref = oracle.infer_from_g(sd, entry["config"], unit, g.unsqueeze(-1), noise, taps)
    eng = _engine(entry, sd, dev, "f16")
    lim = entry["subsample_limit"]

    def vs_golden(name, got_cm):            # got_cm in the reference layout (B, C, T)
        want = gold[name]
        got = subsample(got_cm, lim)
        assert got.shape == want.shape, (name, got.shape, want.shape)
        return snr_db(want, got)

    z_p = eng.enc_p(unit, noise)
    assert snr_db(_fm(taps["enc_p.z_p"]), z_p.cpu()) >= 50.0
    assert vs_golden("enc_p.z_p", _fm(z_p.cpu())) >= 50.0
    wn_out = eng.wn_stack(0, _fm(taps["enc_p.pre"]))                  # WN.forward on its own (qvc_wn_stack)
    assert snr_db(_fm(taps["enc_p.enc.layer15.out"]), wn_out.cpu()) >= 50.0
    assert vs_golden("enc_p.enc.out", _fm(wn_out.cpu())) >= 50.0
    z = eng.flow_reverse(_fm(taps["enc_p.z_p"]), g)
    assert snr_db(_fm(taps["flow.flows.0.out"]), z.cpu()) >= 50.0
    assert vs_golden("flow.flows.0.out", _fm(z.cpu())) >= 50.0
    post = eng.dec_trunk(_fm(taps["flow.flows.0.out"]), g)
    assert snr_db(_fm(taps["dec.subband_conv_post"]), post.cpu()) >= 45.0
    assert vs_golden("dec.subband_conv_post", _fm(post.cpu())) >= 45.0
    out, ymb = eng.istft_synth(_fm(taps["dec.subband_conv_post"]), want_bands=True)
    assert snr_db(ref, out.cpu()) >= 100.0
    assert vs_golden("dec.y_mb", ymb.cpu()) >= 100.0
    torch.cuda.synchronize()


def test_wn_stack_entry_point(lib, dev):
    """qvc_wn_stack: WN.forward (modules.py:69-114) of the unconditioned enc_p stack and of a g-conditioned
    coupling-layer stack, against the oracle's WN on the same input."""
    entry, _ = load_case("mini")
    _m, sd, unit, g, noise = regenerate(entry)
    cfg = entry["config"]
    sdf = {k: v.float() for k, v in sd.items()}
    eng = _engine(entry, sd, dev, "f16")
    H = cfg["hidden_channels"]
    x = torch.randn(entry["batch"], H, entry["frames"], generator=torch.Generator().manual_seed(3))
    want0 = oracle.wn_forward(sdf, "enc_p.enc", x, None, H, 5, 16)
    got0 = eng.wn_stack(0, _fm(x))
    assert snr_db(_fm(want0), got0.cpu()) >= 50.0
    for i in range(4):
        want = oracle.wn_forward(sdf, f"flow.flows.{2 * i}.enc", x, g.unsqueeze(-1), H, 5, 4)
        got = eng.wn_stack(1 + i, _fm(x), g)
        assert snr_db(_fm(want), got.cpu()) >= 50.0, i
    from quickvc_official_amd import lib as L
    with pytest.raises(L.QvcError):
        eng.wn_stack(5, _fm(x), g)                                     # only n_flows coupling stacks exist
    with pytest.raises(L.QvcError):
        eng.wn_stack(1, _fm(x), None)                                  # a coupling stack needs g


def test_per_layer_wn_kernel_with_big_lds_tiles(lib, dev):
    """hidden = 256 on the per-layer WaveNet kernel (debug switch wn_chunk = -1) with >= 512 tiles: 64-frame tiles need
    (64 + 4 + 64) * 512 B = 67.6 KB of dynamic LDS, i.e. the > 64 KiB opt-in (it used to be missing there)."""
    import quickvc_official_amd as q
    from quickvc_official_amd.engine import QvcEngine
    from quickvc_official_amd.synth import make_synthetic_state_dict, make_synthetic_inputs
    cfg = dict(q.DEFAULT_MODEL_CONFIG, inter_channels=64, hidden_channels=256, upsample_initial_channel=64, gin_channels=32)
    model = q.SynthesizerTrn(641, 32, **cfg)
    sd = make_synthetic_state_dict(model, 78)
    B, T = 16, 1024                                                    # 16 * ceil(1024 / 32) = 512 tiles
    unit, _g, noise = make_synthetic_inputs(B, T, 256, 64, 32, seed0=9)
    eng = QvcEngine(model.model_config, sd, dev)
    from quickvc_official_amd import lib as L
    z_stack = eng.enc_p(unit, noise)
    L.debug_set("wn_chunk", -1)
    z_layer = eng.enc_p(unit, noise)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(z_layer).all())
    assert snr_db(z_stack.cpu(), z_layer.cpu()) >= 100.0
    taps = {}
    sdf = {k: v.float() for k, v in sd.items()}
    want, _mu, _logs = oracle.cond_normal_wn(sdf, "enc_p", unit[:1], noise[:1], 256, 64, None, taps)
    assert snr_db(_fm(want), z_layer[:1].cpu()) >= 45.0


def test_fused_post_tail_matches_two_launches(lib, dev):
    """conv_post + iSTFT/FIR tail as one launch (the default, qvc_post_tail_impl.h) against the debug switch
    post_tail = 0 (conv_post -> fp32 frames in memory -> istft_synth_kernel): same K order in the GEMM, same per-item
    math in the tail, so the waveforms must agree bit for bit -- whole batch (two utterances, 250 frames: tiles at both
    ends and in the middle) and a ragged batch (tiles past an utterance's end).  Third variant: the launch shapes of
    earlier rounds -- chains interleaved on the CUs, one chain per launch at stage 1, one workgroup per row chunk in
    up-sampler 1: where a workgroup runs must not change a bit."""
    from quickvc_official_amd import lib as L
    entry, _ = load_case("full_b2")
    _m, sd, unit, g, noise = regenerate(entry)
    variants = ({"post_tail": 1}, {"post_tail": 0},
                {"post_tail": 1, "pair_cm4": 0, "pair_wide_launch": 0, "conv_cl": 0})
    outs = []
    for extra in variants:
        for k, v in DEBUG_DEFAULTS.items():
            L.debug_set(k, extra.get(k, v))
        res = {}
        for dt in ("f16", "bf16x"):
            eng = _engine(entry, sd, dev, dt)
            out, recs = eng.infer_batch_timed(unit.to(dev), g.to(dev), noise.to(dev))
            torch.cuda.synchronize()
            names = sorted(set(r["name"] for r in recs if r["name"].startswith(("post_tail", "istft"))))
            fusedtail = extra["post_tail"] == 1
            assert any(n.startswith("post_tail<") for n in names) == fusedtail, names
            assert any(n.startswith("istft_synth") for n in names) == (not fusedtail), names
            lens = torch.tensor([unit.shape[2], 77], dtype=torch.int32)
            rag = eng.infer_batch_ragged(unit.to(dev), g.to(dev), noise.to(dev), lens.to(dev))
            torch.cuda.synchronize()
            res[dt] = (out.cpu(), rag.cpu())
        outs.append(res)
    for dt in ("f16", "bf16x"):
        for other in (1, 2):
            assert torch.equal(outs[0][dt][0], outs[other][dt][0]), (dt, other)
            assert torch.equal(outs[0][dt][1], outs[other][dt][1]), (dt, other)
        assert outs[0][dt][1][1, 0, 320 * 77:].abs().max() == 0 and outs[0][dt][1][1, 0, :320 * 77].abs().max() > 0


def test_chained_resblock_is_bit_identical(lib, dev):
    """The k = 3 ResBlock of each stage as ONE launch (qvc_chain_impl.h: three pairs chained on chip, the stream read
    once and written once; opt-in through the debug switch pair_chain3 = 1 because it measured slower)
    against the default pair-by-pair launches: same K order, the stream
    rounded to its memory type after every pair, intermediates zeroed outside the utterance -> bit-identical, in all
    three operand modes, for a whole batch and a ragged one (sequence ends inside tiles and inside halos)."""
    from quickvc_official_amd import lib as L
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    _m, sd, _u, _g, _n = regenerate(entry)
    unit, g, noise = make_synthetic_inputs(6, 250, 256, 192, 256, seed0=5200)
    lens = torch.tensor([250, 249, 97, 24, 160, 3], dtype=torch.int32)
    res = []
    for chain3 in (1, 0):
        L.debug_set("pair_chain3", chain3)
        per = {}
        for dt in ("f16", "bf16x", "bf16"):
            eng = _engine(entry, sd, dev, dt)
            out, recs = eng.infer_batch_timed(unit.to(dev), g.to(dev), noise.to(dev))
            rag = eng.infer_batch_ragged(unit.to(dev), g.to(dev), noise.to(dev), lens.to(dev))
            torch.cuda.synchronize()
            names = [r["name"] for r in recs]
            n_chain = sum(n.startswith("rbchain<") for n in names)
            n_pair = sum(n.startswith("rbpair<") for n in names)
            assert (n_chain, n_pair) == (2, 6) if chain3 else (n_chain == 0 and n_pair == 6), (chain3, dt, names)
            per[dt] = (out.cpu(), rag.cpu())
        res.append(per)
    for dt in ("f16", "bf16x", "bf16"):
        assert torch.equal(res[0][dt][0], res[1][dt][0]), dt
        assert torch.equal(res[0][dt][1], res[1][dt][1]), dt


def test_continuous_stream_wn_kernel_is_bit_identical(lib, dev):
    """The continuous-stream WaveNet stack kernel (qvc_wn2_impl.h: the default at hidden 192 / kernel 5) against the
    generic stack kernel (debug switch wn_kernel = 1): same K order per output, same roundings -> bit-identical, for
    enc_p (4 chained launches, the last with the network's final layer), the coupling layers (pre / post fused),
    a ragged batch and a single short utterance."""
    from quickvc_official_amd import lib as L
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    _m, sd, _u, _g, _n = regenerate(entry)
    unit, g, noise = make_synthetic_inputs(5, 250, 256, 192, 256, seed0=4100)
    lens = torch.tensor([250, 33, 181, 64, 2], dtype=torch.int32)
    res = []
    for variant in (0, 1):
        L.debug_set("wn_kernel", variant)
        per = {}
        for dt in ("f16", "bf16"):
            eng = _engine(entry, sd, dev, dt)
            z = eng.enc_p(unit, noise)
            zf = eng.flow_reverse(z, g)
            ws = [eng.wn_stack(i, z.new_zeros(5, 250, 192).normal_(generator=torch.Generator(device=dev).manual_seed(7 + i)), g) for i in range(5)]
            out, recs = eng.infer_batch_timed(unit.to(dev), g.to(dev), noise.to(dev))
            rag = eng.infer_batch_ragged(unit.to(dev), g.to(dev), noise.to(dev), lens.to(dev))
            one = eng.infer_batch(unit[:1, :, :40].to(dev), g[:1].to(dev), noise[:1, :, :40].to(dev))
            torch.cuda.synchronize()
            n2 = sum(r["name"].startswith("wn_stack2<") for r in recs)
            n1 = sum(r["name"].startswith("wn_stack<") for r in recs)
            assert (n2, n1) == ((8, 0) if variant == 0 else (0, 8)), (variant, n2, n1)
            per[dt] = [t.cpu() for t in (z, zf, out, rag, one, *ws)]
        res.append(per)
    for dt in ("f16", "bf16"):
        for x, y in zip(res[0][dt], res[1][dt]):
            assert torch.equal(x, y), dt


def test_mixed_bf16x_mode_meets_40_db(lib, dev):
    """operand_dtype="bf16x" (QVC_BF16X): bf16 MFMA operands in the fused ResBlock pairs (80 % of the FLOPs) with their
    residual stream kept in f16, f16 operands elsewhere.
    BASELINE.json labels its configs bf16 and asks for >= 40 dB: all-bf16 measures ~34.5 dB on the shipped config
    (the generator's ~75 chained convs and exp() amplify 8-bit-mantissa rounding), the mixed mode must clear the
    bar.  Checked on the reference's golden waveform (full_b1) and on the oracle for a second utterance."""
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, gold = load_case("full_b1")
    _m, sd, unit, g, noise = regenerate(entry)
    res = {}
    for dt in ("bf16", "bf16x", "f16"):
        eng = _engine(entry, sd, dev, dt)
        out = eng.infer_batch(unit.to(dev), g.to(dev), noise.to(dev))
        torch.cuda.synchronize()
        res[dt] = snr_db(gold["o"].reshape(-1), out.cpu().reshape(-1).numpy())
    assert res["bf16x"] >= 40.0, res
    assert res["f16"] >= 45.0 and res["bf16"] >= 30.0, res
    assert res["bf16"] < res["bf16x"] <= res["f16"] + 1.0, res
    u2, g2, n2 = make_synthetic_inputs(1, 250, 256, 192, 256, seed0=901)
    ref = oracle.infer_from_g(sd, entry["config"], u2, g2.unsqueeze(-1), n2)
    out = _engine(entry, sd, dev, "bf16x").infer_batch(u2.to(dev), g2.to(dev), n2.to(dev))
    assert snr_db(ref[0], out[0].cpu()) >= 40.0


def test_f16_dynamic_range_and_saturation_counter(lib, dev):
    """What the f16 residual streams / f16 WaveNets do when activations grow (the pretrained checkpoint is not
    available, so the range of real weights is unknown): the synthetic checkpoint with ups[1]'s weight_g scaled by
    alpha and subband_conv_post's by 1 / alpha -- the generator is positively homogeneous between the two up to its
    biases, so the waveform keeps its scale while every stage-2 tensor grows by alpha.
      * alpha such that stage-2 activations peak in [1e4, 6e4] (f16 max 65504): the path still meets its bar against
        the fp32 oracle on the same checkpoint (f16 >= 45 dB, bf16x >= 40 dB), and the debug library
        (libqvc_hip_sat.so, -DQVC_SATCOUNT) counts ZERO saturated conversions;
      * alpha 8x larger: activations exceed the f16 range, the conversions saturate (values clamp at +-65504 instead of
        becoming inf) -- and the counter says so (> 0), instead of the damage passing silently.
    The product library answers qvc_debug_saturations with QVC_ERR_BAD_CONFIG (it does not count)."""
    import ctypes, os
    import quickvc_official_amd as q
    from quickvc_official_amd import lib as L
    from quickvc_official_amd.engine import QvcEngine
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    model, sd, _u, _g, _n = regenerate(entry)
    unit, g, noise = make_synthetic_inputs(2, 120, 256, 192, 256, seed0=6400)
    cnt = ctypes.c_int64(0)
    assert lib.qvc_debug_saturations(ctypes.byref(cnt), 1) == -2 and cnt.value == -1          # QVC_ERR_BAD_CONFIG: not a debug build
    sat_path = os.path.join(os.path.dirname(L._LIB_PATH), "libqvc_hip_sat.so")
    satlib = ctypes.CDLL(sat_path)
    L.declare(satlib)

    def scaled(alpha):
        sd2 = {k: v.clone() for k, v in sd.items()}
        sd2["dec.ups.1.weight_g"] = sd["dec.ups.1.weight_g"] * alpha
        sd2["dec.subband_conv_post.weight_g"] = sd["dec.subband_conv_post.weight_g"] / alpha
        return sd2

    def peak(sd2):
        taps = {}
        ref = oracle.infer_from_g(sd2, entry["config"], unit, g.unsqueeze(-1), noise, taps)
        pk = max(float(taps[k].abs().max()) for k in taps if k.startswith(("dec.ups.1", "dec.resblocks.3", "dec.resblocks.4", "dec.resblocks.5")))
        return ref, pk

    _ref1, pk1 = peak(sd)
    alpha = 3.0e4 / pk1
    sd_a = scaled(alpha)
    ref, pk = peak(sd_a)
    assert 1.0e4 <= pk <= 6.0e4, pk
    for dt, bar in (("f16", 45.0), ("bf16x", 40.0)):
        eng = QvcEngine(dict(model.model_config, operand_dtype=dt), sd_a, dev)
        eng.lib = satlib                                     # same ABI, the counting twin
        assert satlib.qvc_debug_saturations(ctypes.byref(cnt), 1) == 0
        out = eng.infer_batch(unit.to(dev), g.to(dev), noise.to(dev))
        torch.cuda.synchronize()
        assert satlib.qvc_debug_saturations(ctypes.byref(cnt), 1) == 0
        assert cnt.value == 0, (dt, cnt.value)
        snrs = [snr_db(ref[b], out[b].cpu()) for b in range(2)]
        assert min(snrs) >= bar, (dt, pk, snrs)
    sd_b = scaled(alpha * 8.0)
    _refb, pkb = peak(sd_b)
    assert pkb > 65504.0
    eng = QvcEngine(dict(model.model_config, operand_dtype="f16"), sd_b, dev)
    eng.lib = satlib
    out = eng.infer_batch(unit.to(dev), g.to(dev), noise.to(dev))
    torch.cuda.synchronize()
    assert satlib.qvc_debug_saturations(ctypes.byref(cnt), 1) == 0
    assert cnt.value > 0                                     # visible, not silent
    assert bool(torch.isfinite(out).all())                   # saturated, not inf / nan


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


def test_wn_stack_kernel_equals_per_layer_kernels(lib, dev):
    """The whole-stack WaveNet kernels (4 layers per launch, overlap-tiled, coupling pre/post fused) against the
    path for configurations they do not cover -- one fused launch per layer, pre / post as separate convs --
    selected with the debug switch wn_chunk = -1.  Same K order per frame -> identical results."""
    from quickvc_official_amd import lib as L
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    _m, sd, _u, _g, _n = regenerate(entry)
    eng = _engine(entry, sd, dev, "f16")
    unit, g, noise = make_synthetic_inputs(6, 250, 256, 192, 256, seed0=300)
    z_stack = eng.enc_p(unit, noise)
    zf_stack = eng.flow_reverse(z_stack, g)
    _, recs = eng.infer_batch_timed(unit.to(dev), g.to(dev), noise.to(dev))
    assert sum(r["name"].startswith("wn_stack2<f16,W12,L4") for r in recs) == 8      # enc_p: 4 launches of 4 layers; 4 flows
    L.debug_set("wn_chunk", -1)
    z_layer = eng.enc_p(unit, noise)
    zf_layer = eng.flow_reverse(z_stack, g)
    _, recs = eng.infer_batch_timed(unit.to(dev), g.to(dev), noise.to(dev))
    torch.cuda.synchronize()
    names = [r["name"] for r in recs]
    assert sum(n.startswith("wn_layer<f16,W12") for n in names) == 32 and not any(n.startswith("wn_stack") for n in names)
    assert sum(n.startswith("conv<") for n in names) == 2 + 4 * 2 + 1 + 2          # + 4 x (pre, post); conv_post rides in post_tail
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


def test_two_batches_in_flight_are_bit_identical(lib, dev):
    """The library keeps no state between calls: two batches on two HIP streams, each with its own workspace, output and
    hipGraph (what bench.py --in-flight 2 and the corpus pipeline's lanes do), overlap on the device and still give
    exactly what each gives alone.  Shipped config, batch 8 x 100 frames, the two lanes replayed alternately."""
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    _m, sd, _u, _g, _n = regenerate(entry)
    eng = _engine(entry, sd, dev, "bf16x")
    B, T = 8, 100
    lanes = []
    for k in range(2):
        unit, g, noise = make_synthetic_inputs(B, T, 256, 192, 256, seed0=8800 + 100 * k)
        lane = dict(unit=unit.to(dev), g=g.to(dev), noise=noise.to(dev), ws=eng.alloc_workspace(B, T),
                    out=torch.empty(B, 1, 320 * T, device=dev), stream=torch.cuda.Stream(dev), graph=torch.cuda.CUDAGraph())
        with torch.cuda.stream(lane["stream"]):
            lane["alone"] = eng.infer_batch(lane["unit"], lane["g"], lane["noise"], ws=lane["ws"]).clone()
            lane["stream"].synchronize()
            with torch.cuda.graph(lane["graph"], stream=lane["stream"]):
                eng.infer_batch(lane["unit"], lane["g"], lane["noise"], lane["out"], ws=lane["ws"])
        lanes.append(lane)
    torch.cuda.synchronize()
    for lane in lanes:
        lane["out"].zero_()
    torch.cuda.synchronize()
    for _ in range(6):
        for lane in lanes:
            with torch.cuda.stream(lane["stream"]):
                lane["graph"].replay()
    torch.cuda.synchronize()
    for lane in lanes:
        assert torch.equal(lane["out"], lane["alone"])
    assert not torch.equal(lanes[0]["out"], lanes[1]["out"])


def test_timed_variant_reports_every_launch(lib, dev):
    entry, _ = load_case("mini")
    _m, sd, unit, g, noise = regenerate(entry)
    eng = _engine(entry, sd, dev, "f16")
    out, recs = eng.infer_batch_timed(unit.to(dev), g.to(dev), noise.to(dev))
    ref = eng.infer_batch(unit.to(dev), g.to(dev), noise.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    names = [r["name"] for r in recs]
    assert names.count("post_tail<f16>") == 1 and names.count("istft_synth") == 0 and names.count("cond_gemv") == 1
    # enc: pre + proj; dec: conv_pre + 2 ups (the coupling layers' pre / post ride in their stack launch, conv_post in
    # the tail's); enc_p WaveNet: 4 launches of 4 layers; 4 coupling stacks; 2 x 9 fused ResBlock pairs
    assert sum(n.startswith("conv<") for n in names) == 2 + 1 + 2
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


def test_chunked_streaming_at_stated_size(lib, dev):
    """BASELINE configs[4] at its stated size: 64 concurrent streams, 320-frame hop, hipGraph replay, shipped
    config.  Property at full size (no oracle run): the concatenated chunks equal the whole-utterance conversion
    of the same 64 streams; three chunks (T = 960), the first and last windows clamped to the sequence edges."""
    from quickvc_official_amd.streaming import ChunkedConverter
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    model, sd, _u, _g, _n = regenerate(entry)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    S, T, hop = 64, 960, 320
    unit, g, noise = make_synthetic_inputs(S, T, 256, 192, 256, seed0=4000)
    conv = ChunkedConverter(model, streams=S, hop_frames=hop, context=88, use_graph=True)
    assert conv._graph is not None and conv.window == 496
    # device-resident inputs that are still being produced on the caller's stream when convert() is entered
    ud, gd, nd = unit.cuda(), g.cuda(), noise.cuda()
    ud2, nd2 = ud * 1.0, nd * 1.0
    # a larger request on the shared engine between building the converter and using it must not invalidate
    # the captured graph's workspace (the converter owns its own)
    whole = model.infer_batch(ud, gd, nd)
    chunked = conv.convert(ud2, gd, nd2)
    torch.cuda.synchronize()
    assert chunked.shape == whole.shape == (S, 1, 320 * T) and bool(torch.isfinite(chunked).all())
    worst = min(snr_db(whole[s].cpu(), chunked[s].cpu()) for s in range(0, S, 7))
    assert worst >= 90.0, worst
    assert snr_db(whole.cpu(), chunked.cpu()) >= 90.0


def test_chunked_streaming_default_noise_is_seeded_like_infer_batch(lib, dev):
    """noise=None draws the noise on the caller's stream; the converter's side stream must wait for it (it used to
    race).  Same seed -> same draw -> same waveform as infer_batch with that noise."""
    from quickvc_official_amd.streaming import ChunkedConverter
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    model, sd, _u, _g, _n = regenerate(entry)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    S, T = 2, 600
    unit, g, _ = make_synthetic_inputs(S, T, 256, 192, 256, seed0=77)
    conv = ChunkedConverter(model, streams=S, hop_frames=160, context=88, use_graph=True)
    torch.manual_seed(1234)
    got = conv.convert(unit.cuda(), g.cuda())                          # noise drawn inside, on the device
    torch.manual_seed(1234)
    noise = torch.randn(S, 192, T, device="cuda")
    want = model.infer_batch(unit.cuda(), g.cuda(), noise)
    torch.cuda.synchronize()
    assert snr_db(want.cpu(), got.cpu()) >= 90.0


@pytest.mark.parametrize("hop,T", [(320, 960), (16, 250)])
def test_incremental_streaming_at_stated_size(lib, dev, hop, T):
    """qvc_stream_step (segment rings, csrc/qvc_stream.h): 64 concurrent streams on the shipped config, 320-frame hop
    (BASELINE configs[4]) and a 16-frame hop, hipGraph replay per step.  Property at full size: the concatenated
    step outputs equal the whole-utterance conversion of the same streams (same noise) -- including the first
    frames, the last frames and the flush -- and a stream that ends early gets zeros after its end."""
    from quickvc_official_amd.streaming import StreamConverter
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    model, sd, _u, _g, _n = regenerate(entry)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    S = 64
    unit, g, noise = make_synthetic_inputs(S, T, 256, 192, 256, seed0=6000 + hop)
    conv = StreamConverter(model, streams=S, hop_frames=hop, use_graph=True)
    assert conv._graph is not None and conv.lag == 32 + 4 * 8 + 20 + 6 and conv.noise_lag == 32
    streamed = conv.convert(unit.cuda(), g.cuda(), noise.cuda())
    whole = model.infer_batch(unit.cuda(), g.cuda(), noise.cuda())
    torch.cuda.synchronize()
    assert streamed.shape == whole.shape == (S, 1, 320 * T) and bool(torch.isfinite(streamed).all())
    worst = min(snr_db(whole[s].cpu(), streamed[s].cpu()) for s in range(0, S, 9))
    assert worst >= 90.0, worst
    assert snr_db(whole.cpu(), streamed.cpu()) >= 90.0


def test_incremental_streaming_odd_hop(lib, dev):
    """A hop that is not a multiple of four frames (7): the step's batched copies (copy_batch_kernel) then move rows whose
    byte widths / pitches are not multiples of 16 and fall back to 4-byte elements for those descriptors.  3 streams, no
    graph, streamed == whole utterance."""
    from quickvc_official_amd.streaming import StreamConverter
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    model, sd, _u, _g, _n = regenerate(entry)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    S, T, hop = 3, 121, 7
    unit, g, noise = make_synthetic_inputs(S, T, 256, 192, 256, seed0=6400)
    conv = StreamConverter(model, streams=S, hop_frames=hop, use_graph=False)
    streamed = conv.convert(unit.cuda(), g.cuda(), noise.cuda())
    whole = model.infer_batch(unit.cuda(), g.cuda(), noise.cuda())
    torch.cuda.synchronize()
    assert streamed.shape == whole.shape == (S, 1, 320 * T) and bool(torch.isfinite(streamed).all())
    assert min(snr_db(whole[s].cpu(), streamed[s].cpu()) for s in range(S)) >= 90.0


def test_streams_of_different_lengths_in_the_mixed_operand_mode(lib, dev):
    """Streams that end at different frames, `bf16x` operands, hop 40: every stream's concatenated step outputs
    equal that stream converted alone through the offline path in the same operand mode (the stream ends inside a
    window, later steps see only padding), zeros after its end."""
    import quickvc_official_amd as q
    from quickvc_official_amd.streaming import StreamConverter
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    _m, sd, _u, _g, _n = regenerate(entry)
    model = q.SynthesizerTrn(641, 32, **dict(entry["config"], operand_dtype="bf16x"))
    model.load_state_dict(sd)
    model = model.cuda().eval()
    lens = [200, 143, 57]
    unit, g, noise = make_synthetic_inputs(3, 200, 256, 192, 256, seed0=7100)
    conv = StreamConverter(model, streams=3, hop_frames=40, use_graph=True)
    streamed = conv.convert(unit.cuda(), g.cuda(), noise.cuda(), lengths=torch.tensor(lens, dtype=torch.int32))
    torch.cuda.synchronize()
    for b, n in enumerate(lens):
        alone = model.infer_batch(unit[b:b + 1, :, :n].cuda(), g[b:b + 1].cuda(), noise[b:b + 1, :, :n].cuda())
        assert snr_db(alone[0].cpu(), streamed[b, :, :320 * n].cpu()) >= 90.0, (b, n)
        if n < 200:
            assert float(streamed[b, :, 320 * n:].abs().max()) == 0.0


def test_stream_slots_admit_and_retire_streams_independently(lib, dev):
    """A streaming server's life cycle (BASELINE configs[4], 64 concurrent streams): 96 utterances of random lengths go
    through 64 slots -- a stream is admitted into whichever slot is free at that step (StreamConverter.start: that
    slot's ring rows zeroed, its position back to 0 on the device, the captured graph unchanged), ends at its own frame
    (end_slot) and the slot is reused as soon as it has flushed.  Every stream's concatenated step outputs equal that
    utterance converted offline (>= 90 dB, the bar of the other streaming tests); slots at different positions,
    starts and ends inside windows, reuse after a previous occupant."""
    import quickvc_official_amd as q
    from quickvc_official_amd.streaming import StreamConverter
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    _m, sd, _u, _g, _n = regenerate(entry)
    model = q.SynthesizerTrn(641, 32, **entry["config"])
    model.load_state_dict(sd)
    model = model.cuda().eval()
    S, hop, N = 64, 64, 96
    rs = np.random.RandomState(33)
    lens = [int(v) for v in rs.randint(30, 331, size=N)]
    tmax = max(lens)
    unit, g, noise = make_synthetic_inputs(N, tmax, 256, 192, 256, seed0=9100)
    unit, g, noise = unit.cuda(), g.cuda(), noise.cuda()
    eng = model.engine()
    offline = torch.cat([eng.infer_batch_ragged(unit[i:i + 32], g[i:i + 32], noise[i:i + 32], torch.tensor(lens[i:i + 32], dtype=torch.int32)).clone()
                         for i in range(0, N, 32)], 0)
    conv = StreamConverter(model, streams=S, hop_frames=hop, use_graph=True)
    lag, nl, spf = conv.lag, conv.noise_lag, conv.spf
    conv.reset(lengths=torch.zeros(S, dtype=torch.int32))              # every slot idle: length 0 -> its tiles return at once
    occupant = [None] * S
    got = {i: torch.zeros(lens[i] * spf, device=dev) for i in range(N)}
    nxt, done, steps = 0, 0, 0
    admitted_at = {}
    while done < N:
        for s in range(S):
            if occupant[s] is None and nxt < N and (steps % 3 != 1 or s % 2 == 0):      # staggered admissions
                conv.start(s, g[nxt], length=None if nxt % 2 else lens[nxt])           # half the streams announce their length later
                occupant[s] = nxt
                admitted_at[nxt] = steps
                nxt += 1
        u = torch.zeros(S, 256, hop, device=dev)
        nz = torch.zeros(S, 192, hop, device=dev)
        for s, i in enumerate(occupant):
            if i is None:
                continue
            p = conv.position(s)
            a, b = p, min(p + hop, lens[i])
            if b > a:
                u[s, :, :b - a] = unit[i, :, a:b]
            a, b = max(p - nl, 0), min(p - nl + hop, lens[i])
            if b > a:
                nz[s, :, a - (p - nl):b - (p - nl)] = noise[i, :, a:b]
            if i % 2 and p <= lens[i] < p + hop:
                conv.end_slot(s, lens[i])                                             # the stream ends inside this hop
        pos_before = [conv.position(s) for s in range(S)]
        out = conv.step(u, nz)
        steps += 1
        for s, i in enumerate(occupant):
            if i is None:
                continue
            f0 = pos_before[s] - lag
            lo, hi = max(f0, 0), min(f0 + hop, lens[i])
            if hi > lo:
                got[i][lo * spf:hi * spf] = out[s, (lo - f0) * spf:(hi - f0) * spf]
            if conv.finished(s):
                occupant[s] = None
                done += 1
        assert steps < 400
    torch.cuda.synchronize()
    assert len(set(admitted_at.values())) >= 4                                         # really admitted at different steps
    for i in range(N):
        assert snr_db(offline[i, 0, :lens[i] * spf].cpu(), got[i].cpu()) >= 90.0, (i, lens[i], admitted_at[i])


def test_ragged_batch_at_benchmark_size(lib, dev):
    """B = 32 utterances of random lengths in [100, 250] (the corpus case, BASELINE configs[3]) in ONE ragged
    call, `bf16x` operands: finite, zeros after every end, and three members spot-checked against the same
    utterance converted alone (>= 100 dB) and against the fp32 oracle (>= 40 dB, the mode's bar)."""
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("full_b1")
    _m, sd, _u, _g, _n = regenerate(entry)
    eng = _engine(entry, sd, dev, "bf16x")
    rs = np.random.RandomState(11)
    lens = [int(x) for x in rs.randint(100, 251, size=32)]
    lens[5] = 250
    unit, g, noise = make_synthetic_inputs(32, 250, 256, 192, 256, seed0=8000)
    out = eng.infer_batch_ragged(unit.to(dev), g.to(dev), noise.to(dev), torch.tensor(lens, dtype=torch.int32))
    torch.cuda.synchronize()
    assert out.shape == (32, 1, 80000) and bool(torch.isfinite(out).all())
    for b, n in enumerate(lens):
        if n < 250:
            assert float(out[b, :, 320 * n:].abs().max()) == 0.0, b
    for b in (0, 5, 31):
        n = lens[b]
        alone = eng.infer_batch(unit[b:b + 1, :, :n].to(dev), g[b:b + 1].to(dev), noise[b:b + 1, :, :n].to(dev))
        torch.cuda.synchronize()
        assert snr_db(alone[0].cpu(), out[b, :, :320 * n].cpu()) >= 100.0, (b, n)
        ref = oracle.infer_from_g(sd, entry["config"], unit[b:b + 1, :, :n], g[b:b + 1].unsqueeze(-1), noise[b:b + 1, :, :n])
        assert snr_db(ref[0], out[b, :, :320 * n].cpu()) >= 40.0, (b, n)


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
    assert sum(n.startswith("conv<") for n in names) == 2 + 1 + 2 + 18       # + 9 x (conv1, conv2) of stage 1
    for b in range(2):
        assert snr_db(ref[b], out[b].cpu()) >= 45.0


@pytest.mark.parametrize("name,over", [
    # other widths: proj rows pair up at 2 fragments per wave, up-sampler rows stay in natural order at stage 1
    ("narrow", dict(inter_channels=128, hidden_channels=96, upsample_initial_channel=256, gin_channels=128)),
    # other up-sampling geometry (x4, x4: a phase count that divides the kernel size on one stage only) and ResBlocks
    ("x4x4", dict(upsample_rates=[4, 4], upsample_kernel_sizes=[15, 16], resblock_kernel_sizes=[3, 5, 7],
                  resblock_dilation_sizes=[[1, 2, 3], [1, 2, 3], [1, 2, 3]], upsample_initial_channel=384)),
    # multi-band decoder: fixed PQMF synthesis instead of the learned FIR (models.py:250-279, pqmf.py:106-117)
    ("multiband", dict(ms_istft_vits=False, mb_istft_vits=True, upsample_initial_channel=256, inter_channels=96, hidden_channels=128)),
])
def test_other_configurations_vs_oracle(lib, dev, name, over):
    """Configurations the goldens do not cover, against the CPU oracle (no golden: the reference was not run on
    them): every fusion this path takes by default -- paired proj rows with the sampling epilogue, lane-packed
    up-samplers with skipped zero taps, conv_post + tail in one launch, three chains per launch -- has to fall back or
    re-shape itself correctly when channel counts, up-sampling rates and kernel sizes change."""
    import quickvc_official_amd as q
    from quickvc_official_amd.engine import QvcEngine
    from quickvc_official_amd.synth import make_synthetic_state_dict, make_synthetic_inputs
    cfg = dict(q.DEFAULT_MODEL_CONFIG, **over)
    model = q.SynthesizerTrn(641, 32, **cfg)
    sd = make_synthetic_state_dict(model, 311)
    B, T = 3, 41
    unit, g, noise = make_synthetic_inputs(B, T, 256, cfg["inter_channels"], cfg["gin_channels"], seed0=71)
    ref = oracle.infer_from_g(sd, cfg, unit, g.unsqueeze(-1), noise)
    # bf16x: BASELINE.json's 40 dB bar on every configuration (measured 45.1-47.1 dB: profiles/r03_snr_configs.json)
    for dt, min_db in (("f16", 45.0), ("bf16x", 40.0)):
        eng = QvcEngine(dict(model.model_config, operand_dtype=dt), sd, dev)
        out = eng.infer_batch(unit.to(dev), g.to(dev), noise.to(dev))
        lens = torch.tensor([T, 17, 30], dtype=torch.int32)
        rag = eng.infer_batch_ragged(unit.to(dev), g.to(dev), noise.to(dev), lens.to(dev))
        torch.cuda.synchronize()
        spf = eng.samples_per_frame
        assert out.shape == (B, 1, T * spf)
        for b in range(B):
            assert snr_db(ref[b], out[b].cpu()) >= min_db, (name, dt, b)
        assert torch.equal(rag[0], out[0])                      # full-length member of the ragged batch == the plain batch
        assert rag[1, 0, 17 * spf:].abs().max() == 0 and rag[1, 0, :17 * spf].abs().max() > 0


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
def test_wave_to_mel_matches_reference_fixtures(lib, dev):
    """qvc_wave_to_mel against tests/golden/mel.npz: log-mels the REFERENCE's mel_processing.wave_to_mel produced
    (imported with a stub librosa whose filters.mel returns this repo's filter bank; script
    tests/golden/make_golden_mel.py) for three waveform lengths, shipped parameters (1280 / 320 / 80 mel).
    Pins the STFT / magnitude / log arithmetic to the reference; the filter bank itself is PARITY UNPINNED
    (librosa absent).  Tolerance |log-mel difference| <= 2e-5: fp32 summation order of a 1280-term DFT vs the
    FFT's butterflies; the clamp at 1e-5 bounds the log's sensitivity."""
    import os
    import sys
    import helpers
    sys.path.insert(0, helpers.GOLDEN)
    from make_golden_mel import synth_wave
    from quickvc_official_amd.frontend import MelFrontend
    gold = dict(np.load(os.path.join(helpers.GOLDEN, "mel.npz")))
    fe = MelFrontend(1280, 80, 16000, 320, 1280, 0.0, None, device=dev)
    for n, seed in zip(gold["lengths"], gold["seeds"]):
        wave = synth_wave(int(n), int(seed))
        mel = fe(wave.to(dev))
        torch.cuda.synchronize()
        want = gold[f"mel{int(n)}"]
        assert mel.shape == want.shape == (1, 80, fe.frames(int(n)))
        assert float(np.abs(mel.cpu().numpy() - want).max()) <= 2e-5, n


@pytest.mark.parametrize("samples,U", [(80000, 2), (16000 * 3 + 123, 1), (2000, 3), (641, 1)])
def test_wave_to_mel_matches_oracle(lib, dev, samples, U):
    """qvc_wave_to_mel (fp32 MFMA STFT + sparse mel + log) vs the oracle's restatement of mel_processing.py:15-98
    (oracle/qvc_oracle.py, itself pinned bit-for-bit to the reference by golden/mel.npz), batches and lengths
    the fixtures do not cover: a multiple of the hop, a ragged one, one shorter than a window, the shortest the
    reflect pad allows.  Tolerance as above (measured 1.4e-6)."""
    from quickvc_official_amd.frontend import MelFrontend, mel_basis
    gen = torch.Generator().manual_seed(samples + U)
    t = torch.arange(samples) / 16000.0
    wave = 0.3 * torch.sin(2 * np.pi * 220.0 * t)[None] * torch.rand(U, 1, generator=gen) + 0.05 * torch.randn(U, samples, generator=gen)
    wave = wave.clamp(-1, 1)
    fe = MelFrontend(1280, 80, 16000, 320, 1280, 0.0, None, device=dev)
    mel = fe(wave.to(dev))
    torch.cuda.synchronize()
    ref = oracle.wave_to_mel(wave, torch.from_numpy(mel_basis(16000, 1280, 80, 0.0, None)), 1280, 320, 1280)
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
    # same seed, same batching (longest first: a and b share the first ragged launch, c follows) -> identical noise draws:
    # the pipeline draws one (n, inter, Tmax) tensor per batch from a generator keyed by (seed, first list line of the batch)
    inter = q.MINI_MODEL_CONFIG["inter_channels"]
    n_ab = cli.batch_noise(7, 0, 2, inter, 81, dev)
    noises = {"a": n_ab[0], "b": n_ab[1], "c": cli.batch_noise(7, 2, 1, inter, 40, dev)[0]}
    for name in "abc":
        _, got = wavfile.read(str(out / f"t_{name}.wav"))
        unit = torch.from_numpy(np.load(str(tmp_path / f"{name}.npy"))).t()[None].cuda()
        ref = net.infer_batch(unit, g, noises[name][None])           # the utterance converted ALONE
        assert snr_db(ref[0, 0].cpu().numpy(), got) >= 100.0, name


def test_convert_cli_on_the_shape_of_the_reference_demo_pair(lib, dev, tmp_path):
    """BASELINE configs[0] (`p225_001.wav -> p226_005.wav`) with stand-ins of the same shape: the reference's demo
    recordings cannot travel to the GPU box, so a 26 007-sample 16-bit source (-> 81 unit frames; units are synthetic
    anyway because HuBERT-soft cannot be fetched) and a 120 744-sample speech-like target (load -> energy trim ->
    HIP mel -> HIP speaker encoder) are synthesised here.  The written file follows the reference's output
    convention (output/quickvc/*.wav): float32, 16 kHz, 320 samples per unit frame = 25 920 samples."""
    import json
    from scipy.io import wavfile
    import quickvc_official_amd as q
    from quickvc_official_amd import convert as cli
    from quickvc_official_amd.checkpoint import save_checkpoint
    from quickvc_official_amd.frontend import load_wav, trim
    from quickvc_official_amd.synth import make_synthetic_state_dict
    cfg = {"train": {"segment_size": 10240}, "data": dict(q.DEFAULT_DATA_CONFIG), "model": dict(q.DEFAULT_MODEL_CONFIG)}
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    model = q.SynthesizerTrn(641, 32, **q.DEFAULT_MODEL_CONFIG)
    model.load_state_dict(make_synthetic_state_dict(model, 1234))
    save_checkpoint(model, None, 2e-4, 1, str(tmp_path / "G_1.pth"))

    def speechlike(n, seed):                                        # voiced bursts between silences, int16
        rs = np.random.RandomState(seed)
        t = np.arange(n) / 16000.0
        x = 0.3 * np.sin(2 * np.pi * (140.0 + 30.0 * np.sin(2 * np.pi * 0.7 * t)) * t) + 0.02 * rs.randn(n)
        x *= (np.sin(2 * np.pi * 0.9 * t) > -0.2) * (t > 0.15) * (t < t[-1] - 0.2)
        return (np.clip(x, -1, 1) * 32767).astype(np.int16)
    wavfile.write(str(tmp_path / "p225_001.wav"), 16000, speechlike(26007, 1))
    wavfile.write(str(tmp_path / "p226_005.wav"), 16000, speechlike(120744, 2))
    frames = 26007 // 320                                           # HuBERT-soft's 20 ms hop: 81 frames
    np.save(str(tmp_path / "p225_001.npy"), np.random.RandomState(1).randn(frames, 256).astype(np.float32))
    (tmp_path / "convert.txt").write_text(f"title1|{tmp_path}/p225_001.wav|{tmp_path}/p226_005.wav\n")
    cli.main(["--hpfile", str(tmp_path / "config.json"), "--ptfile", str(tmp_path / "G_1.pth"), "--txtpath", str(tmp_path / "convert.txt"),
              "--outdir", str(tmp_path / "out"), "--seed", "1"])
    rate, got = wavfile.read(str(tmp_path / "out" / "title1.wav"))
    assert rate == 16000 and got.dtype == np.float32 and got.shape == (320 * 81,) and np.isfinite(got).all()
    tgt = load_wav(str(tmp_path / "p226_005.wav"), 16000)
    assert len(tgt) == 120744 and 0 < len(trim(tgt, top_db=20)) < len(tgt)


def test_convert_cli_corpus_pipeline(lib, dev, tmp_path):
    """BASELINE configs[3] through the CLI on one GPU: 208 unit files of random lengths, 4 targets, converted by the
    three-stage pipeline (native loader pool -> ragged batches on a side stream -> native wav writer).  Every list line
    gets its wav of 320 samples per unit frame; three of them (first / middle / last of the length-sorted plan) equal
    the utterance converted ALONE with the same speaker embedding and the same noise draw (>= 100 dB: same kernels,
    same K order, other tiles); the frame-major upload equals the (B, 256, T) one bit for bit."""
    import json
    from scipy.io import wavfile
    import quickvc_official_amd as q
    from quickvc_official_amd import convert as cli
    from quickvc_official_amd.checkpoint import save_checkpoint
    from quickvc_official_amd.frontend import MelFrontend, load_wav, trim
    from quickvc_official_amd.synth import make_synthetic_state_dict
    cfg = {"train": {"segment_size": 10240}, "data": dict(q.DEFAULT_DATA_CONFIG), "model": dict(q.MINI_MODEL_CONFIG)}
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    model = q.SynthesizerTrn(641, 32, **q.MINI_MODEL_CONFIG)
    model.load_state_dict(make_synthetic_state_dict(model, 21))
    save_checkpoint(model, None, 2e-4, 1, str(tmp_path / "G_1.pth"))
    sr = cfg["data"]["sampling_rate"]
    t = np.arange(int(1.5 * sr)) / sr
    for k in range(4):
        wavfile.write(str(tmp_path / f"spk{k}.wav"), sr, (0.4 * np.sin(2 * np.pi * (140.0 + 45.0 * k) * t) * 32767).astype(np.int16))
    rng = np.random.RandomState(12)
    N = 208
    lens = [int(v) for v in rng.randint(20, 121, size=N)]
    for i, n in enumerate(lens):
        np.save(str(tmp_path / f"u{i:03d}.npy"), rng.randn(n, 256).astype(np.float32))
    items = [(f"o{i:03d}", str(tmp_path / f"u{i:03d}.npy"), str(tmp_path / f"spk{i % 4}.wav")) for i in range(N)]
    (tmp_path / "convert.txt").write_text("".join(f"{a}|{b}|{c}\n" for a, b, c in items))
    out = tmp_path / "out"
    cli.main(["--hpfile", str(tmp_path / "config.json"), "--ptfile", str(tmp_path / "G_1.pth"), "--txtpath", str(tmp_path / "convert.txt"),
              "--outdir", str(out), "--seed", "5", "--batch", "16", "--io-threads", "4"])
    waves = {}
    for i in range(N):
        rate, w = wavfile.read(str(out / f"o{i:03d}.wav"))
        assert rate == sr and w.dtype == np.float32 and w.shape == (320 * lens[i],), i
        assert np.isfinite(w).all() and np.abs(w).max() > 0, i
        waves[i] = w
    # three utterances alone: same g (HIP mel + HIP speaker encoder, as the CLI computes it), same noise draw
    net = q.SynthesizerTrn(641, 32, **q.MINI_MODEL_CONFIG).cuda().eval()
    q.load_checkpoint(str(tmp_path / "G_1.pth"), net, None)
    d = cfg["data"]
    front = MelFrontend(d["filter_length"], d["n_mel_channels"], sr, d["hop_length"], d["win_length"], d["mel_fmin"], d["mel_fmax"])
    lengths, mine, batches = cli.rank_plan(items, 0, 1, 16)
    assert lengths == lens and sorted(mine) == list(range(N)) and sum(len(b) for b in batches) == N
    inter = q.MINI_MODEL_CONFIG["inter_channels"]
    for bi, row in ((0, 0), (len(batches) // 2, 3), (len(batches) - 1, -1)):
        idxs = batches[bi]
        i = idxs[row]
        tmax = max(lens[j] for j in idxs)
        noise = cli.batch_noise(5, idxs[0], len(idxs), inter, tmax, dev)[row % len(idxs), :, :lens[i]].unsqueeze(0)
        wav = torch.from_numpy(trim(load_wav(items[i][2], sr), top_db=20)).unsqueeze(0).cuda()
        g = net.speaker_embed(front(wav))
        unit = torch.from_numpy(np.load(items[i][1])).t().unsqueeze(0).cuda()
        alone = net.infer_batch(unit, g, noise)
        torch.cuda.synchronize()
        assert snr_db(alone[0, 0].cpu().numpy(), waves[i]) >= 100.0, (bi, i)
    # frame-major units (as on disk) == the reference's (B, 256, T) layout, bit for bit
    eng = net.engine()
    u = torch.randn(3, 256, 50, device=dev)
    n3 = torch.randn(3, inter, 50, device=dev)
    g3 = torch.nn.functional.normalize(torch.rand(3, q.MINI_MODEL_CONFIG["gin_channels"], device=dev), dim=1)
    l3 = torch.tensor([50, 20, 33], dtype=torch.int32)
    a_cm = eng.infer_batch_ragged(u, g3, n3, l3)
    a_fm = eng.infer_batch_ragged(u.transpose(1, 2).contiguous(), g3, n3, l3, unit_fm=True)
    torch.cuda.synchronize()
    assert torch.equal(a_cm, a_fm)


def test_bench_starts_its_own_workers(lib, dev):
    """`python bench.py --gpus 2` with no launcher around it must start its two ranks itself (fresh processes, before any
    GPU call in the parent), have rank 0 pack and broadcast the blob, and print ONE JSON line for the whole job.
    On this one-GPU box the ranks share cuda:0 and talk over gloo (--rehearsal); the driver's 2 / 4 / 8-GPU runs take
    the same code with one GPU per rank and RCCL."""
    import json
    import os
    import subprocess
    import sys
    from helpers import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4",
                          "--rehearsal", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["batch_per_gpu"] == 4 and "roofline" in out
    # a failing worker must fail the parent: --gpus 2 with an impossible batch
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "0",
                          "--rehearsal", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0


def test_convert_cli_two_ranks_rehearsal(lib, dev, tmp_path):
    """BASELINE configs[3] control flow on a one-GPU box: the CLI started as TWO ranks (RANK / WORLD_SIZE as
    torch.distributed.run sets them; ``--device 0`` puts both on cuda:0).  Each rank plans from the .npy headers,
    converts only its own shard through the ragged path and writes its own files: together exactly one wav per list
    line, each 320 samples per unit frame.  There is no collective in this mode, so the ranks run one after the other
    (nothing in the property under test needs them to share the GPU at the same time)."""
    import json
    import os
    import subprocess
    import sys
    from scipy.io import wavfile
    import quickvc_official_amd as q
    from quickvc_official_amd.checkpoint import save_checkpoint
    from quickvc_official_amd.synth import make_synthetic_state_dict
    from helpers import ROOT
    cfg = {"train": {"segment_size": 10240}, "data": dict(q.DEFAULT_DATA_CONFIG), "model": dict(q.MINI_MODEL_CONFIG)}
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    model = q.SynthesizerTrn(641, 32, **q.MINI_MODEL_CONFIG)
    model.load_state_dict(make_synthetic_state_dict(model, 21))
    save_checkpoint(model, None, 2e-4, 1, str(tmp_path / "G_1.pth"))
    sr = cfg["data"]["sampling_rate"]
    t = np.arange(int(1.2 * sr)) / sr
    for k, f0 in enumerate((150.0, 230.0)):
        wavfile.write(str(tmp_path / f"spk{k}.wav"), sr, (0.4 * np.sin(2 * np.pi * f0 * t) * 32767).astype(np.int16))
    frames = {"a": 60, "b": 45, "c": 44, "d": 30, "e": 20, "f": 59, "g": 31}
    rng = np.random.RandomState(8)
    for name, n in frames.items():
        np.save(str(tmp_path / f"{name}.npy"), rng.randn(n, 256).astype(np.float32))
    (tmp_path / "convert.txt").write_text("".join(f"t_{n}|{tmp_path}/{n}.npy|{tmp_path}/spk{i % 2}.wav\n" for i, n in enumerate(frames)))
    outs = [tmp_path / "out0", tmp_path / "out1"]
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", PYTHONPATH=ROOT)
        res = subprocess.run([sys.executable, "-m", "quickvc_official_amd.convert", "--hpfile", str(tmp_path / "config.json"),
                              "--ptfile", str(tmp_path / "G_1.pth"), "--txtpath", str(tmp_path / "convert.txt"),
                              "--outdir", str(outs[r]), "--seed", "3", "--batch", "2", "--device", "0"], env=env, cwd=ROOT,
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
    written = [sorted(os.listdir(o)) for o in outs]
    assert sorted(written[0] + written[1]) == sorted(f"t_{n}.wav" for n in frames)       # a partition of the list
    assert written[0] and written[1]
    for r in range(2):
        for fn in written[r]:
            rate, wav = wavfile.read(str(outs[r] / fn))
            assert rate == sr and wav.dtype == np.float32 and wav.shape == (320 * frames[fn[2:-4]],)
            assert np.isfinite(wav).all() and np.abs(wav).max() > 0


def test_ragged_batch_equals_each_utterance_alone(lib, dev):
    """qvc_infer_batch_ragged on the shipped config: a batch of lengths {250, 181, 81, 40} (padding filled with
    junk) must give, for every member, the waveform of that utterance converted alone -- the reference converts
    any length per call (convert.py:58-86) -- and zeros after its end; the 250-frame member is the reference's own
    golden case (tests/golden/full_b1.npz)."""
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, gold = load_case("full_b1")
    _m, sd, unit0, g0, noise0 = regenerate(entry)
    eng = _engine(entry, sd, dev, "f16")
    lens = [250, 181, 81, 40]
    unit, g, noise = make_synthetic_inputs(4, 250, 256, 192, 256, seed0=500)
    unit[0], g[0], noise[0] = unit0[0], g0[0], noise0[0]
    pad_u, pad_n = unit.clone(), noise.clone()
    for b, n in enumerate(lens):
        pad_u[b, :, n:] = 300.0 * (b + 1)                            # junk the path must never read unmasked
        pad_n[b, :, n:] = float("nan")
    out = eng.infer_batch_ragged(pad_u.to(dev), g.to(dev), pad_n.to(dev), torch.tensor(lens, dtype=torch.int32))
    torch.cuda.synchronize()
    assert out.shape == (4, 1, 80000) and bool(torch.isfinite(out).all())
    for b, n in enumerate(lens):
        alone = eng.infer_batch(unit[b:b + 1, :, :n].to(dev), g[b:b + 1].to(dev), noise[b:b + 1, :, :n].to(dev))
        torch.cuda.synchronize()
        assert snr_db(alone[0].cpu(), out[b, :, :320 * n].cpu()) >= 100.0, (b, n)
        if n < 250:
            assert float(out[b, :, 320 * n:].abs().max()) == 0.0
    assert snr_db(gold["o"].reshape(-1), out[0].cpu().reshape(-1).numpy()) >= 45.0
    # the Python surface: a list of utterances in, a list of waveforms out
    import quickvc_official_amd as q
    model = q.SynthesizerTrn(641, 32, **entry["config"])
    model.load_state_dict(sd)
    model = model.cuda().eval()
    waves = model.infer_ragged([unit[b, :, :n] for b, n in enumerate(lens)], g.cuda(), [noise[b, :, :n] for b, n in enumerate(lens)])
    assert [tuple(w.shape) for w in waves] == [(1, 320 * n) for n in lens]
    assert snr_db(out[2, :, :320 * 81].cpu(), waves[2].cpu()) >= 100.0


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
    (debug switch wn_chunk = -1: unfused pre/post) must agree with the fused one."""
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
    from quickvc_official_amd import lib as L
    L.debug_set("wn_chunk", -1)
    z_p2 = eng.flow_forward(z[:3], g[:3])
    torch.cuda.synchronize()
    assert snr_db(z_p[:3].cpu(), z_p2.cpu()) >= 100.0
