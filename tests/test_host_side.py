"""CPU tests (no GPU): C-ABI surface, packer errors, host emulation of the launch sequence vs the
oracle, config / checkpoint drop-in behaviour, and the multi-process sharding logic (gloo)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import qvc_oracle as oracle
from helpers import ROOT, load_case, regenerate, snr_db


@pytest.fixture(scope="module")
def built():
    """Build (or reuse) the product library and the test-only emulation; host entry points only."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    from quickvc_official_amd import lib as L
    return L.load_library()


def test_library_exports_every_declared_symbol(built):
    header = open(os.path.join(ROOT, "include", "qvc.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(qvc_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 33
    for n in sorted(names):
        assert hasattr(built, n), f"{n} declared in include/qvc.h but not exported"
    assert built.qvc_abi_version() == 8
    assert built.qvc_status_string(0) == b"ok" and b"missing" in built.qvc_status_string(-3)


def test_no_hot_kernel_spills_to_scratch(built):
    """hipcc's per-kernel resource remarks (kept by build.py next to the objects): the MFMA kernels must not use
    scratch memory and must keep the occupancy their launch geometry assumes (two 4-wave workgroups or one 8-wave
    workgroup per CU = 2 waves per SIMD).  A by-reference use of a by-value kernel argument once sent the whole
    argument struct to scratch and halved the occupancy of the pair kernel -- silently, at +60 % run time."""
    import glob
    # the instantiations the shipped config launches (other widths have variants that do spill; they are correct, just slower)
    hot = ("rbpair_kernelIDF16_Li2ELi10ELi4ELi4E", "rbpair_kernelIDF16_Li2ELi10ELi8ELi8E", "conv_mfma_kernelIDF16_",
           "wn_stack_kernelIDF16_Li3ELi0ELi12E", "wn_stack_kernelIDF16_Li3ELi1ELi12E", "wn_layer_kernelIDF16_Li2ELb0ELi12E",
           "wn_layer_kernelIDF16_Li2ELb1ELi12E", "post_tail_kernelIDF16_Li4E",
           "wn_stack2_kernelIDF16_Li6ELi5ELi0E", "wn_stack2_kernelIDF16_Li6ELi5ELi1E")
    seen = set()
    paths = [os.path.join(ROOT, "quickvc-official_amd", "csrc", "_obj", n) for n in ("qvc_conv_f16.remarks.txt", "qvc_wn2.remarks.txt")]
    for path in paths:
        name = None
        for line in open(path):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            tag = next((h for h in hot if name and h in name), None)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and tag:
                assert int(m.group(1)) == 0, (name, line)
                seen.add(tag)
            m = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", line)
            if m and tag and "rbpair_kernel" in tag:
                assert int(m.group(1)) >= 2, (name, line)
            if m and tag and ("post_tail_kernel" in tag or "wn_stack2_kernel" in tag):   # post_tail: three workgroups per CU (its phases are
                assert int(m.group(1)) >= 3, (name, line)                                 # serial); wn_stack2: one 12-wave workgroup
    assert seen == set(hot), set(hot) - seen


def test_io_library_exports_and_matches_numpy_and_scipy(built, tmp_path):
    """libqvc_io.so (include/qvc_io.h): every declared symbol is exported; unit files written by np.save (the reference's
    on-disk format, dataset/encode.py:38 -- format 1.0 and 2.0 headers) come back bit for bit in the frame-major batch
    layout; wav files are byte-identical to scipy.io.wavfile.write as convert.py:84-86 calls it; bad inputs give the
    documented error codes instead of raising from native code."""
    import numpy as np
    from scipy.io import wavfile
    from quickvc_official_amd import fileio
    lib = fileio.load_library()
    header = open(os.path.join(ROOT, "include", "qvc_io.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(qvc_io_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 6
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/qvc_io.h but not exported"
    rng = np.random.RandomState(0)
    arrs = [rng.randn(n, 256).astype(np.float32) for n in (40, 7, 123, 1)]
    paths = [str(tmp_path / f"u{i}.npy") for i in range(4)]
    for p, a in zip(paths, arrs):
        np.save(p, a)
    with open(paths[3], "wb") as f:                                     # a format-2.0 header (4-byte header length)
        np.lib.format.write_array(f, arrs[3], version=(2, 0))
    assert [fileio.npy_shape(p) for p in paths] == [(40, 256), (7, 256), (123, 256), (1, 256)]
    pool = fileio.IoPool(3)
    assert pool.npy_shapes(paths) == [(40, 256), (7, 256), (123, 256), (1, 256)]
    dst = torch.full((4, 123, 256), -7.0)
    lens = torch.zeros(4, dtype=torch.int32)
    pool.load_units(paths, dst, lens)
    assert lens.tolist() == [40, 7, 123, 1]
    for i, a in enumerate(arrs):
        assert np.array_equal(dst[i, :len(a)].numpy(), a)
    assert float(dst[1, 7:].max()) == -7.0                              # rows past an utterance's end are left alone
    src = torch.from_numpy(rng.randn(3, 5000).astype(np.float32))
    outs = [str(tmp_path / f"o{i}.wav") for i in range(3)]
    pool.write_wavs(outs, src, [5000, 320, 0], 16000)
    for i, n in enumerate((5000, 320, 0)):
        ref = str(tmp_path / f"r{i}.wav")
        wavfile.write(ref, 16000, src[i, :n].numpy())
        assert open(ref, "rb").read() == open(outs[i], "rb").read()
    with pytest.raises(fileio.QvcIoError, match="more frames than the slot"):
        pool.load_units(paths, torch.zeros(4, 50, 256), lens)
    np.save(str(tmp_path / "f64.npy"), np.zeros((3, 256)))
    with pytest.raises(fileio.QvcIoError, match="float32"):
        fileio.npy_shape(str(tmp_path / "f64.npy"))
    with pytest.raises(fileio.QvcIoError, match="opened"):
        fileio.npy_shape(str(tmp_path / "missing.npy"))
    pool.close()


def test_package_fails_loudly_without_gpu_or_library(built, monkeypatch):
    import quickvc_official_amd as q
    from quickvc_official_amd import lib as L
    model = q.SynthesizerTrn(641, 32, **q.MINI_MODEL_CONFIG)
    unit = torch.zeros(1, 256, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.infer_batch(unit, torch.zeros(1, 64))
    monkeypatch.setattr(L, "_LIB_PATH", "/nonexistent/libqvc_hip.so")
    monkeypatch.setattr(L, "_lib", None)
    with pytest.raises(L.QvcError, match="missing"):
        L.load_library()


def test_pack_weights_error_codes(built):
    import quickvc_official_amd as q
    from quickvc_official_amd import lib as L
    from quickvc_official_amd.synth import make_synthetic_state_dict
    model = q.SynthesizerTrn(641, 32, **q.MINI_MODEL_CONFIG)
    sd = make_synthetic_state_dict(model, 1)
    cfg = L.make_config(model.model_config)
    blob = L.pack_weights(built, cfg, sd)
    assert blob.numel() == built.qvc_blob_bytes(ctypes.byref(cfg)) and blob.data_ptr() % 256 == 0
    # enc_q and enc_spk are not on the path: packing must not need them (tolerant like utils.load_checkpoint)
    small = {k: v for k, v in sd.items() if not k.startswith(("enc_q.", "enc_spk."))}
    assert torch.equal(L.pack_weights(built, cfg, small), blob)
    missing = {k: v for k, v in sd.items() if k != "flow.flows.2.post.weight"}
    with pytest.raises(L.QvcError, match="missing"):
        L.pack_weights(built, cfg, missing)
    bad = dict(sd); bad["dec.conv_pre.weight_v"] = sd["dec.conv_pre.weight_v"][:, :-1]
    with pytest.raises(L.QvcError, match="shape"):
        L.pack_weights(built, cfg, bad)
    cfg2 = L.make_config(dict(model.model_config)); cfg2.n_flows = 3
    assert built.qvc_blob_bytes(ctypes.byref(cfg2)) == -2
    cfg3 = L.make_config(dict(model.model_config)); cfg3.inter_channels = 50
    assert built.qvc_workspace_bytes(ctypes.byref(cfg3), 1, 10) == -2
    assert built.qvc_workspace_bytes(ctypes.byref(cfg), 0, 10) == -1
    # ConvTranspose1d paddings whose output is not rate*T (models.py:335 with k-s+1-i odd) would overrun the
    # workspace carve-up: they are refused, not mis-sized
    mc4 = dict(model.model_config); mc4["upsample_rates"] = [4, 4]; mc4["upsample_kernel_sizes"] = [16, 16]
    cfg4 = L.make_config(mc4)
    assert built.qvc_workspace_bytes(ctypes.byref(cfg4), 2, 50) == -2 and built.qvc_blob_bytes(ctypes.byref(cfg4)) == -2
    mc5 = dict(model.model_config); mc5["upsample_rates"] = [4, 4]; mc5["upsample_kernel_sizes"] = [15, 16]   # 15-4+1 even, 16-4+1-1 even
    assert built.qvc_workspace_bytes(ctypes.byref(L.make_config(mc5)), 2, 50) > 0
    # streaming queries: sizes grow with the hop, the lag is the sum of the segments' reaches, bad arguments are codes
    assert built.qvc_stream_lag_frames(ctypes.byref(cfg)) == 32 + 4 * 8 + 20 + 6 and built.qvc_stream_noise_lag_frames(ctypes.byref(cfg)) == 32
    s16, s320 = built.qvc_stream_state_bytes(ctypes.byref(cfg), 4, 16), built.qvc_stream_state_bytes(ctypes.byref(cfg), 4, 320)
    assert 0 < s16 < s320 and built.qvc_stream_workspace_bytes(ctypes.byref(cfg), 4, 16) > 0
    assert built.qvc_stream_state_bytes(ctypes.byref(cfg), 4, 0) == -1 and built.qvc_stream_state_bytes(ctypes.byref(cfg), 0, 16) == -1
    mc6 = dict(model.model_config); mc6["upsample_rates"] = [20]; mc6["upsample_kernel_sizes"] = [41]; mc6["upsample_initial_channel"] = 16
    assert built.qvc_stream_state_bytes(ctypes.byref(L.make_config(mc6)), 4, 16) == -2            # streaming needs two up-samplers
    assert built.qvc_stream_step(ctypes.byref(cfg), None, None, 0, None, None, None, None, 1, 16, None, None, None, 0, None) == -1


def test_speaker_encoder_pack_and_host_emulation(built):
    """Speaker-encoder blob (qvc_spk_pack_weights) + its launch sequence replayed on the CPU vs the oracle's
    SpeakerEncoder.embed_utterance (models.py:507-546), incl. the reference-generated golden embeddings."""
    import quickvc_official_amd as q
    from quickvc_official_amd import lib as L
    from quickvc_official_amd.synth import make_synthetic_mel
    from emu import emu_speaker_embed
    import helpers
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import qvc_oracle as oracle
    entry, _ = load_case("mini")
    model, sd, _u, _g, _n = regenerate(entry)
    spk = dict(np.load(os.path.join(helpers.GOLDEN, "mini_spk.npz")))
    err = lambda ref, got: float(np.sqrt(((np.asarray(ref, np.float64) - np.asarray(got, np.float64)) ** 2).sum(-1)
                                         / (np.asarray(ref, np.float64) ** 2).sum(-1)).max())
    for frames in (100, 128, 300):
        mel = make_synthetic_mel(frames, 80, seed=7 + frames)
        assert err(spk[f"g_{frames}"], emu_speaker_embed(model.model_config, sd, mel).numpy()) <= 4e-3
    # batch of utterances, a width that is not a multiple of 32 (padded hidden units), bf16 operands
    entry, _ = load_case("odd")
    model, sd, _u, _g, _n = regenerate(entry)
    sdf = {k: v.float() for k, v in sd.items()}
    mel = torch.cat([make_synthetic_mel(140, 80, seed=40 + u) for u in range(3)], 0)
    ref = torch.cat([oracle.speaker_embed_utterance(sdf, mel[u:u + 1].transpose(1, 2)) for u in range(3)], 0).numpy()
    assert err(ref, emu_speaker_embed(model.model_config, sd, mel).numpy()) <= 4e-3
    assert err(ref, emu_speaker_embed(model.model_config, sd, mel, dtype="bf16").numpy()) <= 3e-2
    # error codes
    cfg = L.make_config(model.model_config)
    spk_sd = {k: v for k, v in sd.items() if k.startswith("enc_spk.")}
    with pytest.raises(L.QvcError, match="missing"):
        L.pack_weights(built, cfg, {k: v for k, v in spk_sd.items() if k != "enc_spk.lstm.bias_hh_l1"}, which="spk")
    bad = dict(spk_sd); bad["enc_spk.lstm.weight_ih_l0"] = spk_sd["enc_spk.lstm.weight_ih_l0"][:, :-1]
    with pytest.raises(L.QvcError, match="shape"):
        L.pack_weights(built, cfg, bad, which="spk")
    cfg2 = L.make_config(dict(model.model_config)); cfg2.gin_channels = 320
    assert built.qvc_spk_blob_bytes(ctypes.byref(cfg2)) == -2
    assert built.qvc_spk_workspace_bytes(ctypes.byref(cfg), 0, 10) == -1


@pytest.mark.parametrize("name", ["mini_q", "odd_q"])
def test_posterior_direction_host_emulation(built, name):
    """enc_q + forward flow (models.py:617-618): packed blobs + launch sequence replayed on the CPU vs the values the
    reference produced (golden) and vs the oracle; the forward flow must invert the reverse flow."""
    from emu import emu_posterior
    from quickvc_official_amd.synth import make_synthetic_state_dict, make_synthetic_posterior_inputs
    import quickvc_official_amd as q
    entry = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))[name]
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", entry["file"])))
    cfg = entry["config"]
    model = q.SynthesizerTrn(641, 32, **cfg)
    sd = make_synthetic_state_dict(model, entry["weights_seed"])
    spec, g, noise = make_synthetic_posterior_inputs(entry["batch"], entry["frames"], 641, cfg["inter_channels"], cfg["gin_channels"],
                                                     seed0=entry["inputs_seed0"])
    otaps = {}
    oracle.posterior_encode(sd, cfg, spec, g.unsqueeze(-1), noise, otaps)
    for k in ("enc_q.m", "enc_q.logs", "enc_q.z", "flow.z_p"):                    # oracle pinned by the reference
        assert np.abs(otaps[k].numpy() - gold[k]).max() <= 2e-5 * max(1.0, float(np.abs(gold[k]).max())), k
    z, z_p = emu_posterior(model.model_config, sd, spec, g, noise)
    assert snr_db(gold["enc_q.z"], z.numpy()) >= 50.0
    assert snr_db(gold["flow.z_p"], z_p.numpy()) >= 50.0


def test_encq_pack_error_codes(built):
    """qvc_encq_pack_weights: tolerant of unrelated keys, loud on a missing / mis-shaped enc_q tensor."""
    import quickvc_official_amd as q
    from quickvc_official_amd import lib as L
    from quickvc_official_amd.synth import make_synthetic_state_dict
    model = q.SynthesizerTrn(641, 32, **q.MINI_MODEL_CONFIG)
    sd = make_synthetic_state_dict(model, 3)
    cfg = L.make_config(model.model_config)
    assert int(cfg.spec_channels) == 641
    only_q = {k: v for k, v in sd.items() if k.startswith("enc_q.")}
    blob = L.pack_weights(built, cfg, only_q, which="encq")
    assert blob.numel() == built.qvc_encq_blob_bytes(ctypes.byref(cfg))
    assert torch.equal(L.pack_weights(built, cfg, sd, which="encq"), blob)          # extra keys are ignored
    with pytest.raises(L.QvcError, match="missing"):
        L.pack_weights(built, cfg, {k: v for k, v in only_q.items() if k != "enc_q.enc.cond_layer.weight_v"}, which="encq")
    bad = dict(only_q); bad["enc_q.pre.weight"] = only_q["enc_q.pre.weight"][:, :-1]
    with pytest.raises(L.QvcError, match="shape"):
        L.pack_weights(built, cfg, bad, which="encq")


def test_mel_table_packer(built):
    """qvc_mel_pack_tables (host code): the windowed DFT table, unpacked with the kernel's index math, equals
    hann[k] * cos / -sin(2 pi bin k / n_fft); filter ranges cover exactly the non-zero weights; error codes."""
    from quickvc_official_amd.frontend import mel_basis
    n_fft, hop, n_mels = 64, 16, 10
    basis = np.ascontiguousarray(mel_basis(16000, n_fft, n_mels, 0.0, None), dtype=np.float32)
    n = int(built.qvc_mel_table_bytes(n_fft, n_mels))
    assert n > 0
    tab = np.zeros(n, dtype=np.uint8)
    vp = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)
    assert built.qvc_mel_pack_tables(n_fft, hop, n_mels, vp(basis), vp(tab), n) == 0
    bins, ksteps, nchunk = n_fft // 2 + 1, n_fft // 16, 1
    dft = tab[: nchunk * 4 * ksteps * 4 * 64 * 4 * 4].view(np.float32).reshape(nchunk, 4, ksteps, 4, 64, 4)
    k = np.arange(n_fft)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * k / n_fft)
    for w in range(4):
        for m in range(4):
            for lane in (0, 5, 17, 63):
                b = w * 32 + (m >> 1) * 16 + (lane & 15)
                for s_ in range(ksteps):
                    kk = s_ * 16 + (lane >> 4) * 4 + np.arange(4)
                    want = np.zeros(4) if b >= bins else (win[kk] * np.cos(2 * np.pi * b * kk / n_fft) if m % 2 == 0
                                                          else -win[kk] * np.sin(2 * np.pi * b * kk / n_fft))
                    assert np.abs(dft[0, w, s_, m, lane] - want).max() < 1e-6
    off_basis = (dft.size * 4 + 255) // 256 * 256
    got_basis = tab[off_basis: off_basis + n_mels * bins * 4].view(np.float32).reshape(n_mels, bins)
    assert np.array_equal(got_basis, basis)
    off_range = (off_basis + n_mels * bins * 4 + 255) // 256 * 256
    rng = tab[off_range: off_range + n_mels * 8].view(np.int32).reshape(n_mels, 2)
    for m in range(n_mels):
        nz = np.nonzero(basis[m])[0]
        assert rng[m, 0] == nz[0] and rng[m, 1] == nz[-1] + 1
    assert built.qvc_mel_pack_tables(60, hop, n_mels, vp(basis), vp(tab), n) == -2        # n_fft % 16
    assert built.qvc_mel_pack_tables(n_fft, hop, n_mels, vp(basis), vp(tab), 16) == -5    # small buffer
    assert built.qvc_mel_workspace_bytes(n_fft, hop, 1, 10) == -1                         # shorter than the reflect pad
    assert built.qvc_mel_workspace_bytes(n_fft, hop, 2, 1000) > 0


def test_weight_norm_and_flip_folding_change_nothing(built):
    """Packing the same effective weights from weight_g/weight_v pairs or from pre-folded plain weights
    gives the identical blob (the fold is exact in fp32 before the operand conversion)."""
    import quickvc_official_amd as q
    from quickvc_official_amd import lib as L
    from quickvc_official_amd.synth import make_synthetic_state_dict
    model = q.SynthesizerTrn(641, 32, **q.MINI_MODEL_CONFIG)
    sd = make_synthetic_state_dict(model, 5)
    cfg = L.make_config(model.model_config)
    a = L.pack_weights(built, cfg, sd)
    scaled = dict(sd)
    for k in sd:                       # w = g*v/|v| is invariant to a rescale of v
        if k.endswith(".weight_v") and k.startswith(("dec.resblocks.0", "enc_p.enc.in_layers.3")):
            scaled[k] = sd[k] * 4.0    # power of two: exact
    assert torch.equal(L.pack_weights(built, cfg, scaled), a)


@pytest.mark.parametrize("name,dtype,min_db", [("mini", "f16", 50.0), ("odd", "f16", 50.0), ("mini_mb", "f16", 50.0),
                                               ("mini", "bf16", 32.0), ("mini", "bf16x", 40.0), ("odd", "bf16x", 40.0)])
def test_host_emulation_of_launch_sequence_matches_oracle(built, name, dtype, min_db):
    """The product's launch sequence + packed blob, replayed on the CPU by oracle/qvc_emu.cpp."""
    from emu import emu_infer
    entry, gold = load_case(name)
    model, sd, unit, g, noise = regenerate(entry)
    ref = oracle.infer_from_g(sd, entry["config"], unit, g.unsqueeze(-1), noise)
    out = emu_infer(model.model_config, sd, unit, g, noise, dtype)
    assert out.shape == ref.shape
    assert snr_db(ref, out) >= min_db
    assert snr_db(gold["o"], out.reshape(-1).numpy()) >= min_db


def test_ragged_batch_host_emulation_matches_per_utterance_oracle(built):
    """qvc_infer_batch_ragged's launch sequence replayed on the CPU: utterances of lengths 21 / 13 / 5 padded to 21
    in ONE batch (padding filled with junk) against the oracle run on each utterance alone -- every conv, WaveNet
    layer and the iSTFT envelope must see its own sequence end (convert.py:58-86 converts any length per line)."""
    from emu import emu_infer_ragged
    entry, _ = load_case("odd")
    _m, sd, _u, _g, _n = regenerate(entry)
    cfg = entry["config"]
    from quickvc_official_amd.synth import make_synthetic_inputs
    lens = [21, 13, 5]
    unit, g, noise = make_synthetic_inputs(3, 21, 256, cfg["inter_channels"], cfg["gin_channels"], seed0=70)
    pad_u, pad_n = unit.clone(), noise.clone()
    for b, n in enumerate(lens):
        pad_u[b, :, n:] = 1e3 * (b + 1)                           # junk that must never reach a result
        pad_n[b, :, n:] = -7.0
    out = emu_infer_ragged(cfg, sd, pad_u, g, pad_n, lens, "f16")
    for b, n in enumerate(lens):
        ref = oracle.infer_from_g(sd, cfg, unit[b:b + 1, :, :n], g[b:b + 1].unsqueeze(-1), noise[b:b + 1, :, :n])
        assert snr_db(ref[0], out[b, :, :320 * n]) >= 45.0, (b, n)
        assert float(out[b, :, 320 * n:].abs().max()) == 0.0 if n < 21 else True      # zeros after the utterance's end


def test_streaming_steps_host_emulation_match_whole_utterance(built):
    """qvc_stream_step replayed on the CPU (odd config: 24 / 40 channels): two streams of lengths 37 and 29 fed 7
    frames per step through the segment rings, junk in the padding -- the concatenated step outputs must equal
    the oracle's whole-utterance conversion of each stream (same noise, aligned by the documented lags), including
    the first frames (sequence start inside a window) and the last ones (end inside a window, then flushing)."""
    from emu import emu_stream_convert
    from quickvc_official_amd.synth import make_synthetic_inputs
    entry, _ = load_case("odd")
    _m, sd, _u, _g, _n = regenerate(entry)
    cfg = entry["config"]
    lens = [37, 29]
    unit, g, noise = make_synthetic_inputs(2, 37, 256, cfg["inter_channels"], cfg["gin_channels"], seed0=90)
    out, lag = emu_stream_convert(cfg, sd, unit, g, noise, hop=7, dtype="f16", lens=lens)
    assert lag == 32 + 4 * 8 + 20 + 6
    for b, n in enumerate(lens):
        ref = oracle.infer_from_g(sd, cfg, unit[b:b + 1, :, :n], g[b:b + 1].unsqueeze(-1), noise[b:b + 1, :, :n])
        assert snr_db(ref[0], out[b, :, :320 * n]) >= 45.0, (b, n)
        assert float(out[b, :, 320 * n:].abs().max()) == 0.0 if n < 37 else True


def test_config_and_checkpoint_drop_in(tmp_path, built):
    import quickvc_official_amd as q
    from quickvc_official_amd.synth import make_synthetic_state_dict
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps({"train": {"segment_size": 10240, "seed": 1234},
                                    "data": dict(q.DEFAULT_DATA_CONFIG), "model": dict(q.MINI_MODEL_CONFIG)}))
    hps = q.get_hparams_from_file(str(cfg_path))
    assert hps.data.filter_length == 1280 and hps.train["segment_size"] == 10240 and "model" in hps
    net = q.SynthesizerTrn(hps.data.filter_length // 2 + 1, hps.train.segment_size // hps.data.hop_length, **hps.model)
    assert net.samples_per_frame == 320
    assert sum(p.numel() for p in net.parameters()) > 0
    sd = make_synthetic_state_dict(net, 9)
    ck = tmp_path / "G_100.pth"
    torch.save({"model": {k: v for k, v in sd.items() if k != "enc_p.pre.bias"}, "iteration": 100,
                "optimizer": None, "learning_rate": 2e-4, "extra_key_ignored": 1}, ck)
    torch.save({"model": sd, "iteration": 20, "optimizer": None, "learning_rate": 1e-4}, tmp_path / "G_20.pth")
    before = net.state_dict()["enc_p.pre.bias"].clone()
    _, _, lr, it = q.load_checkpoint(str(ck), net, None)
    assert (lr, it) == (2e-4, 100)
    assert torch.equal(net.state_dict()["enc_p.pre.bias"], before)                 # missing key keeps own value
    assert torch.equal(net.state_dict()["dec.conv_pre.weight_v"], sd["dec.conv_pre.weight_v"])
    assert q.latest_checkpoint_path(str(tmp_path)).endswith("G_100.pth")
    with pytest.raises(RuntimeError, match="Not-supported decoder flag"):
        q.SynthesizerTrn(641, 32, **dict(q.MINI_MODEL_CONFIG, ms_istft_vits=False))
    with pytest.raises(AssertionError):
        q.SynthesizerTrn(641, 32, **dict(q.MINI_MODEL_CONFIG, resblock="2"))


def test_sharding_is_a_partition():
    from quickvc_official_amd.dist import shard_indices
    lengths = [int(x) for x in np.random.RandomState(0).randint(50, 400, size=37)]
    for world in (1, 2, 3, 8):
        shards = [shard_indices(37, r, world, lengths) for r in range(world)]
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(37))
        loads = [sum(lengths[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(lengths)


_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from quickvc_official_amd import dist as qd
rank, _, world = qd.env_world()
dist.init_process_group("gloo", rank=rank, world_size=world)
blob = torch.arange(1000, dtype=torch.int64).to(torch.uint8) if rank == 0 else torch.zeros(1000, dtype=torch.uint8)
qd.broadcast_blob(blob, src=0)
ok = bool((blob == torch.arange(1000, dtype=torch.int64).to(torch.uint8)).all())
mine = qd.shard_indices(10, rank, world)
wall = qd.max_over_ranks(1.0 + rank, torch.device("cpu"))
total = qd.sum_over_ranks(float(len(mine)), torch.device("cpu"))
# the CLI's shard -> batches plan for this rank, from the .npy headers alone (convert.rank_plan)
from quickvc_official_amd.convert import rank_plan
items = [ln.strip().split("|") for ln in open({listfile!r}) if ln.strip()]
lengths, cli_mine, batches = rank_plan(items, rank, world, 4)
print(json.dumps({{"rank": rank, "ok": ok, "mine": mine, "wall": wall, "total": total, "cli_mine": cli_mine,
                  "batches": batches, "lengths": lengths}}))
dist.destroy_process_group()
"""


def test_world_size_2_broadcast_and_sharding_gloo(tmp_path):
    # a 13-line list whose sources are .npy unit files of different lengths (only their headers are read by the plan)
    rs = np.random.RandomState(3)
    lens = [int(x) for x in rs.randint(30, 300, size=13)]
    lines = []
    for i, n in enumerate(lens):
        np.save(str(tmp_path / f"u{i}.npy"), np.zeros((n, 256), dtype=np.float32))
        lines.append(f"title{i}|{tmp_path}/u{i}.npy|{tmp_path}/spk{i % 3}.wav")
    listfile = tmp_path / "convert.txt"
    listfile.write_text("\n".join(lines) + "\n")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, listfile=str(listfile)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=180)
        assert p.returncode == 0, se
        outs.append(json.loads(so.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert all(d["ok"] for d in outs)
    assert outs[0]["mine"] == [0, 2, 4, 6, 8] and outs[1]["mine"] == [1, 3, 5, 7, 9]
    assert all(d["wall"] == 2.0 and d["total"] == 10.0 for d in outs)
    # the two ranks' CLI plans partition the list, balance the frames, and batch similar lengths together
    assert outs[0]["lengths"] == outs[1]["lengths"] == lens
    assert sorted(outs[0]["cli_mine"] + outs[1]["cli_mine"]) == list(range(13))
    loads = [sum(lens[i] for i in d["cli_mine"]) for d in outs]
    assert abs(loads[0] - loads[1]) <= max(lens)
    for d in outs:
        assert sorted(i for b in d["batches"] for i in b) == sorted(d["cli_mine"])
        assert all(len(b) <= 4 and min(lens[i] for i in b) >= 0.75 * max(lens[i] for i in b) for b in d["batches"])


def test_frontend_and_cli_planning():
    """Host-side pieces around the path (parity-unpinned: librosa is absent, so these
    check the published definitions' properties, not the reference's numbers)."""
    from quickvc_official_amd import frontend as F
    from quickvc_official_amd.convert import plan_batches
    basis = F.mel_basis(16000, 1280, 80, 0.0, None)
    assert basis.shape == (80, 641) and (basis >= 0).all()
    peaks = basis.argmax(axis=1)
    assert (np.diff(peaks) > 0).all()                              # centre frequencies increase
    # Slaney normalisation: every triangle has (almost) the same area in Hz
    areas = basis.sum(axis=1) * (8000.0 / 640)
    assert np.allclose(areas, 1.0, atol=0.08)
    wav = np.zeros(16000, dtype=np.float32)
    wav[4000:12000] = 0.5 * np.sin(2 * np.pi * 220 * np.arange(8000) / 16000)
    t = F.trim(wav, top_db=20)
    assert 7000 <= len(t) <= 10500 and len(t) < len(wav)            # silence dropped, tone kept (512-sample hops)
    mel = oracle.wave_to_mel(torch.from_numpy(wav).unsqueeze(0), torch.from_numpy(basis), 1280, 320, 1280)
    assert mel.shape == (1, 80, 50) and torch.isfinite(mel).all()  # 16000 samples / hop 320 (mel_processing.py:79-98)
    assert float(mel[:, :, 20:30].max()) > float(mel[:, :, :5].max()) + 3.0
    # ragged batches: length-sorted runs of at most `batch`, cut where the padding would exceed 25 %
    plan = plan_batches([81, 250, 81, 250, 250, 120], batch=2)
    assert plan == [[1, 3], [4], [5], [0, 2]]
    plan = plan_batches([250, 240, 181, 81, 40, 239, 200], batch=4)
    assert plan == [[0, 1, 5, 6], [2], [3], [4]]
    lens = list(np.random.RandomState(0).randint(40, 400, size=1000))
    plan = plan_batches(lens, batch=32)
    assert sorted(i for b in plan for i in b) == list(range(1000)) and max(len(b) for b in plan) <= 32
    assert all(min(lens[i] for i in b) >= 0.75 * max(lens[i] for i in b) for b in plan)
    assert len(plan) <= 60                                          # a corpus does not degenerate to batch-1 launches
    ref = os.path.join(ROOT, "..", "reference", "test_data", "p225_001.wav")
    if os.path.exists(ref):                                         # container only: the reference's own demo input
        w = F.load_wav(ref, 16000)
        assert w.dtype == np.float32 and abs(len(w) - 26007) <= 1 and np.abs(w).max() <= 1.0


def test_shipped_configuration_takes_the_fast_paths(built):
    """The plan for the shipped model configuration must select every fused / repacked layout the measured numbers
    rest on -- a silent fall-back (natural rows, separate launches) would still be correct, only slower: proj with
    paired [mu | log sigma] rows (sampling in its epilogue), lane-packed up-samplers, conv_post + tail as one launch,
    fused ResBlock pairs with 4 waves per workgroup at 128 channels and 8 at 256."""
    import quickvc_official_amd as q
    from quickvc_official_amd import lib as L
    import emu
    emulib = emu.load_emu()
    for dt in ("f16", "bf16", "bf16x"):
        cfg = L.make_config(dict(q.SynthesizerTrn(641, 32, **q.DEFAULT_MODEL_CONFIG).model_config, operand_dtype=dt))
        out = (ctypes.c_int32 * 8)()
        assert emulib.qvc_emu_plan_flags(ctypes.byref(cfg), out) == 0
        assert list(out) == [1, 6, 1, 1, 1, 8, 4, 1], (dt, list(out))
    # a narrow configuration keeps natural up-sampler rows where a lane's channels would straddle two phases
    cfg = L.make_config(dict(q.SynthesizerTrn(641, 32, **q.MINI_MODEL_CONFIG).model_config))
    out = (ctypes.c_int32 * 8)()
    assert emulib.qvc_emu_plan_flags(ctypes.byref(cfg), out) == 0
    assert out[0] == 1 and out[4] == 1 and out[7] == 1

