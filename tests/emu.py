"""Test-only loader for oracle/_build/libqvc_emu.so (host replay of the launch sequence)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_emu():
    import importlib
    build = importlib.import_module("quickvc_official_amd.build")
    path = build.build_emu()
    lib = ctypes.CDLL(path)
    from quickvc_official_amd import lib as L
    P, I, Lg, V = ctypes.POINTER, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
    lib.qvc_emu_infer_batch.restype = ctypes.c_int
    lib.qvc_emu_infer_batch.argtypes = [P(L.QvcConfig), V, V, V, V, V, I, I, V, Lg]
    lib.qvc_emu_infer_batch_ragged.restype = ctypes.c_int
    lib.qvc_emu_infer_batch_ragged.argtypes = [P(L.QvcConfig), V, V, V, V, V, I, I, V, V, Lg]
    lib.qvc_emu_stream_step.restype = ctypes.c_int
    lib.qvc_emu_stream_step.argtypes = [P(L.QvcConfig), V, V, Lg, V, V, V, V, I, I, V, V, V, Lg]
    lib.qvc_emu_speaker_embed.restype = ctypes.c_int
    lib.qvc_emu_speaker_embed.argtypes = [P(L.QvcConfig), V, V, V, I, I, V, Lg]
    lib.qvc_emu_enc_q.restype = ctypes.c_int
    lib.qvc_emu_enc_q.argtypes = [P(L.QvcConfig), V, V, V, V, V, I, I, V, Lg]
    lib.qvc_emu_flow_forward.restype = ctypes.c_int
    lib.qvc_emu_flow_forward.argtypes = [P(L.QvcConfig), V, V, V, I, I, V, Lg]
    lib.qvc_emu_tap_offset.restype = Lg
    lib.qvc_emu_tap_offset.argtypes = [P(L.QvcConfig), I, I, I]
    lib.qvc_emu_plan_flags.restype = ctypes.c_int
    lib.qvc_emu_plan_flags.argtypes = [P(L.QvcConfig), P(I)]
    return lib


def emu_infer(model_config, sd, unit, g, noise, dtype="f16", taps=None):
    """Runs the emulated path on CPU. Returns (B,1,N) waveform; fills taps dict with workspace views."""
    from quickvc_official_amd import lib as L
    hip = L.load_library()          # host-only entry points (packer, size queries) work without a GPU
    emu = load_emu()
    mc = dict(model_config, operand_dtype=dtype)
    cfg = L.make_config(mc)
    blob = L.pack_weights(hip, cfg, sd)
    B, _, T = unit.shape
    n_ws = int(hip.qvc_workspace_bytes(ctypes.byref(cfg), B, T))
    raw = torch.zeros(n_ws + 256, dtype=torch.uint8)
    shift = (-raw.data_ptr()) % 256
    ws = raw[shift:shift + n_ws]
    spf = mc["gen_istft_hop_size"] * mc["subbands"]
    for u in mc["upsample_rates"]:
        spf *= u
    out = torch.empty(B, 1, T * spf)
    unit, g, noise = unit.float().contiguous(), g.float().contiguous(), noise.float().contiguous()
    st = emu.qvc_emu_infer_batch(ctypes.byref(cfg), blob.data_ptr(), unit.data_ptr(), g.data_ptr(), noise.data_ptr(),
                                 out.data_ptr(), B, T, ws.data_ptr(), n_ws)
    assert st == 0, hip.qvc_status_string(st)
    if taps is not None:
        C = mc["inter_channels"]
        off = emu.qvc_emu_tap_offset(ctypes.byref(cfg), B, T, 0)
        taps["z"] = ws[off:off + B * T * C * 4].view(torch.float32).reshape(B, T, C).clone()
        tu = 1
        for u in mc["upsample_rates"]:
            tu *= u
        F = T * tu + 1
        off = emu.qvc_emu_tap_offset(ctypes.byref(cfg), B, T, 1)
        taps["post"] = ws[off:off + B * F * 72 * 4].view(torch.float32).reshape(B, F, 72).clone()
        off = emu.qvc_emu_tap_offset(ctypes.byref(cfg), B, T, 5)
        taps["stats"] = ws[off:off + B * T * 2 * C * 4].view(torch.float32).reshape(B, T, 2 * C).clone()
        ch0 = mc["upsample_initial_channel"] // 2
        t1 = T * mc["upsample_rates"][0]
        off = emu.qvc_emu_tap_offset(ctypes.byref(cfg), B, T, 4)
        td = torch.float16 if dtype == "f16" else torch.bfloat16   # stage streams are carried in the operand type
        taps["ups0"] = ws[off:off + B * t1 * ch0 * 2].view(td).reshape(B, t1, ch0).float()
        off = emu.qvc_emu_tap_offset(ctypes.byref(cfg), B, T, 2)
        taps["rb0"] = ws[off:off + B * t1 * ch0 * 2].view(td).reshape(B, t1, ch0).float()
    return out


def emu_infer_ragged(model_config, sd, unit, g, noise, frames, dtype="f16"):
    """Host replay of qvc_infer_batch_ragged: padded unit / noise (B, C, Tmax), frames = per-utterance lengths."""
    from quickvc_official_amd import lib as L
    hip = L.load_library()
    emu = load_emu()
    mc = dict(model_config, operand_dtype=dtype)
    cfg = L.make_config(mc)
    blob = L.pack_weights(hip, cfg, sd)
    B, _, T = unit.shape
    n_ws = int(hip.qvc_workspace_bytes(ctypes.byref(cfg), B, T))
    raw = torch.zeros(n_ws + 256, dtype=torch.uint8)
    shift = (-raw.data_ptr()) % 256
    ws = raw[shift:shift + n_ws]
    spf = mc["gen_istft_hop_size"] * mc["subbands"]
    for u in mc["upsample_rates"]:
        spf *= u
    out = torch.full((B, 1, T * spf), float("nan"))
    unit, g, noise = unit.float().contiguous(), g.float().contiguous(), noise.float().contiguous()
    lens = torch.as_tensor(frames, dtype=torch.int32).contiguous()
    st = emu.qvc_emu_infer_batch_ragged(ctypes.byref(cfg), blob.data_ptr(), unit.data_ptr(), g.data_ptr(), noise.data_ptr(),
                                        out.data_ptr(), B, T, lens.data_ptr(), ws.data_ptr(), n_ws)
    assert st == 0, hip.qvc_status_string(st)
    return out


def emu_stream_convert(model_config, sd, unit, g, noise, hop, dtype="f16", lens=None):
    """Host replay of a whole streamed conversion: feeds `hop` frames per qvc_stream_step until everything (plus the
    lag) is out.  unit (B, 256, T), noise (B, inter, T); lens: per-stream lengths <= T (default T).  -> (B, 1, T*spf)."""
    from quickvc_official_amd import lib as L
    hip = L.load_library()
    emu = load_emu()
    mc = dict(model_config, operand_dtype=dtype)
    cfg = L.make_config(mc)
    blob = L.pack_weights(hip, cfg, sd)
    B, UC, T = unit.shape
    C = mc["inter_channels"]
    n_state = int(hip.qvc_stream_state_bytes(ctypes.byref(cfg), B, hop))
    n_ws = int(hip.qvc_stream_workspace_bytes(ctypes.byref(cfg), B, hop))
    lag, nlag = int(hip.qvc_stream_lag_frames(ctypes.byref(cfg))), int(hip.qvc_stream_noise_lag_frames(ctypes.byref(cfg)))
    assert n_state > 0 and n_ws > 0 and lag > 0

    def aligned(n):
        raw = torch.zeros(n + 256, dtype=torch.uint8)
        shift = (-raw.data_ptr()) % 256
        return raw, raw[shift:shift + n]
    _keep1, state = aligned(n_state)
    _keep2, ws = aligned(n_ws)
    spf = mc["gen_istft_hop_size"] * mc["subbands"]
    for u in mc["upsample_rates"]:
        spf *= u
    lens_t = torch.as_tensor(lens if lens is not None else [T] * B, dtype=torch.int32).contiguous()
    out = torch.zeros(B, 1, T * spf)
    unit, g, noise = unit.float().contiguous(), g.float().contiguous(), noise.float().contiguous()
    steps = -(-(T + lag) // hop)
    for n in range(steps):
        s = n * hop
        u_new = torch.full((B, UC, hop), 123.0)                       # junk past the end must be ignored
        n_new = torch.full((B, C, hop), -55.0)
        lo, hi = s, min(s + hop, T)
        if hi > lo:
            u_new[:, :, :hi - lo] = unit[:, :, lo:hi]
        a, b_ = s - nlag, s - nlag + hop                                # frames enc_p completes in this step
        lo2, hi2 = max(a, 0), min(b_, T)
        if hi2 > lo2:
            n_new[:, :, lo2 - a:hi2 - a] = noise[:, :, lo2:hi2]
        pos = torch.full((B,), s, dtype=torch.int32)
        chunk = torch.full((B, hop * spf), float("nan"))
        st = emu.qvc_emu_stream_step(ctypes.byref(cfg), blob.data_ptr(), state.data_ptr(), n_state, u_new.data_ptr(), g.data_ptr(),
                                     n_new.data_ptr(), chunk.data_ptr(), B, hop, pos.data_ptr(), lens_t.data_ptr(), ws.data_ptr(), n_ws)
        assert st == 0, hip.qvc_status_string(st)
        f0 = s - lag                                                    # the chunk holds frames [f0, f0 + hop)
        lo3, hi3 = max(f0, 0), min(f0 + hop, T)
        if hi3 > lo3:
            out[:, 0, lo3 * spf:hi3 * spf] = chunk[:, (lo3 - f0) * spf:(hi3 - f0) * spf]
    return out, lag


def emu_speaker_embed(model_config, sd, mel, dtype="f16"):
    """Host replay of qvc_speaker_embed: mel (U, n_mel, F) -> g (U, gin)."""
    from quickvc_official_amd import lib as L
    hip = L.load_library()
    emu = load_emu()
    cfg = L.make_config(dict(model_config, operand_dtype=dtype))
    blob = L.pack_weights(hip, cfg, {k: v for k, v in sd.items() if k.startswith("enc_spk.")}, which="spk")
    U, _, F = mel.shape
    n_ws = int(hip.qvc_spk_workspace_bytes(ctypes.byref(cfg), U, F))
    assert n_ws > 0
    ws = torch.zeros(n_ws + 256, dtype=torch.uint8)
    shift = (-ws.data_ptr()) % 256
    ws = ws[shift:shift + n_ws]
    mel = mel.float().contiguous()
    g = torch.empty(U, int(cfg.gin_channels))
    st = emu.qvc_emu_speaker_embed(ctypes.byref(cfg), blob.data_ptr(), mel.data_ptr(), g.data_ptr(), U, F,
                                   ws.data_ptr(), n_ws)
    assert st == 0, hip.qvc_status_string(st)
    return g


def emu_posterior(model_config, sd, spec, g, noise, dtype="f16"):
    """Host replay of qvc_enc_q + qvc_flow_forward: returns (z, z_p) as (B, inter, T)."""
    from quickvc_official_amd import lib as L
    hip = L.load_library()
    emu = load_emu()
    cfg = L.make_config(dict(model_config, operand_dtype=dtype))
    qblob = L.pack_weights(hip, cfg, {k: v for k, v in sd.items() if k.startswith("enc_q.")}, which="encq")
    blob = L.pack_weights(hip, cfg, sd)
    B, _, T = spec.shape
    n_ws = int(hip.qvc_workspace_bytes(ctypes.byref(cfg), B, T))
    raw = torch.zeros(n_ws + 256, dtype=torch.uint8)
    shift = (-raw.data_ptr()) % 256
    ws = raw[shift:shift + n_ws]
    C = int(cfg.inter_channels)
    spec, g, noise = spec.float().contiguous(), g.float().contiguous(), noise.float().contiguous()
    z = torch.empty(B, T, C)
    st = emu.qvc_emu_enc_q(ctypes.byref(cfg), qblob.data_ptr(), spec.data_ptr(), g.data_ptr(), noise.data_ptr(), z.data_ptr(),
                           B, T, ws.data_ptr(), n_ws)
    assert st == 0, hip.qvc_status_string(st)
    zp = z.clone()
    st = emu.qvc_emu_flow_forward(ctypes.byref(cfg), blob.data_ptr(), zp.data_ptr(), g.data_ptr(), B, T, ws.data_ptr(), n_ws)
    assert st == 0, hip.qvc_status_string(st)
    return z.transpose(1, 2).contiguous(), zp.transpose(1, 2).contiguous()
