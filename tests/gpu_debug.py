"""Developer script (GPU box): per-stage parity report.  Not a test; prints SNRs."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes
import torch
import torch.nn.functional as F
import qvc_oracle as o
import quickvc_official_amd as q
from quickvc_official_amd import lib as L
from quickvc_official_amd.engine import QvcEngine
from helpers import load_case, regenerate, snr_db

dev = torch.device("cuda:0")
lib = L.load_library()
print("device check:", lib.qvc_device_check(), torch.cuda.get_device_name(0))


def conv_case(B, cin, cout, T, k, dil, slope, dtype):
    torch.manual_seed(cin * 7 + cout + k + dil)
    x = torch.randn(B, cin, T)
    w = torch.randn(cout, cin, k) / (cin * k) ** 0.5
    bias = torch.randn(cout) * 0.1
    nb = int(lib.qvc_conv1d_scratch_bytes(cout, cin, k)); nw = int(lib.qvc_conv1d_workspace_bytes(B, cout, cin, T))
    sh = torch.empty(nb + 256, dtype=torch.uint8).pin_memory()
    sd_ = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
    ws = torch.empty(nw + 256, dtype=torch.uint8, device=dev)
    al = lambda t: t.data_ptr() + ((-t.data_ptr()) % 256)
    xd = x.to(dev); y = torch.empty(B, cout, T, device=dev)
    st = lib.qvc_conv1d(xd.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, cin, cout, T, k, dil, slope,
                        L.DTYPES[dtype], al(sh), al(sd_), nb, al(ws), nw, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    td = torch.float16 if dtype == "f16" else torch.bfloat16
    xr = F.leaky_relu(x, slope).to(td).float(); wr = w.to(td).float()
    ref_r = F.conv1d(xr.double(), wr.double(), bias.double(), padding=(k - 1) // 2 * dil, dilation=dil).float()
    ref = F.conv1d(F.leaky_relu(x, slope), w, bias, padding=(k - 1) // 2 * dil, dilation=dil)
    print(f"conv B{B} {cin}->{cout} T{T} k{k} d{dil} {dtype}: st={st} SNR(rounded ref)={snr_db(ref_r, y.cpu()):.1f} "
          f"SNR(fp32 ref)={snr_db(ref, y.cpu()):.1f} maxerr={float((ref_r - y.cpu()).abs().max()):.2e}")


for dt in ("f16", "bf16"):
    conv_case(1, 32, 64, 40, 1, 1, 1.0, dt)
    conv_case(2, 64, 64, 37, 3, 1, 0.1, dt)
    conv_case(2, 128, 128, 300, 11, 5, 0.1, dt)
    conv_case(2, 256, 256, 200, 7, 3, 0.1, dt)
    conv_case(1, 192, 384, 250, 5, 1, 1.0, dt)
    conv_case(2, 128, 72, 130, 7, 1, 0.01, dt)
    conv_case(3, 40, 80, 21, 5, 1, 1.0, dt)


def stage_report(name, dtype):
    entry, gold = load_case(name)
    model, sd, unit, g, noise = regenerate(entry)
    cfg = entry["config"]
    taps = {}
    ref = o.infer_from_g(sd, cfg, unit, g.unsqueeze(-1), noise, taps)
    mc = dict(model.model_config, operand_dtype=dtype)
    eng = QvcEngine(mc, sd, dev)
    fm = lambda t: t.transpose(1, 2).contiguous()
    z_p = eng.enc_p(unit, noise); torch.cuda.synchronize()
    print(f"[{name} {dtype}] enc_p   SNR {snr_db(fm(taps['enc_p.z_p']), z_p.cpu()):.1f}")
    z = eng.flow_reverse(fm(taps["enc_p.z_p"]), g); torch.cuda.synchronize()
    print(f"[{name} {dtype}] flow    SNR {snr_db(fm(taps['flow.flows.0.out']), z.cpu()):.1f}")
    post = eng.dec_trunk(fm(taps["flow.flows.0.out"]), g); torch.cuda.synchronize()
    print(f"[{name} {dtype}] trunk   SNR {snr_db(fm(taps['dec.subband_conv_post']), post.cpu()):.1f}")
    out, ymb = eng.istft_synth(fm(taps["dec.subband_conv_post"]), want_bands=True); torch.cuda.synchronize()
    print(f"[{name} {dtype}] tail    SNR {snr_db(ref, out.cpu()):.1f}  y_mb SNR {snr_db(taps['dec.y_mb'], ymb.cpu()):.1f}")
    full = eng.infer_batch(unit.to(dev), g.to(dev), noise.to(dev)); torch.cuda.synchronize()
    print(f"[{name} {dtype}] full    SNR {snr_db(ref, full.cpu()):.1f}  vs golden o: {snr_db(gold['o'], full.cpu().reshape(-1).numpy()):.1f}")


for name in ("mini", "odd", "mini_t37", "full_b2"):
    for dt in ("f16", "bf16"):
        try:
            stage_report(name, dt)
        except Exception as e:  # keep going: this is a report
            print(f"[{name} {dt}] FAILED: {type(e).__name__}: {e}")

# quick timing at the benchmark shape
entry, gold = load_case("full_b1")
model, sd, unit, g, noise = regenerate(entry)
from quickvc_official_amd.synth import make_synthetic_inputs
for dt in ("f16", "bf16"):
    eng = QvcEngine(dict(model.model_config, operand_dtype=dt), sd, dev)
    for B in (1, 32):
        u, gg, nn = make_synthetic_inputs(B, 250, 256, 192, 256)
        u, gg, nn = u.to(dev), gg.to(dev), nn.to(dev)
        out = eng.infer_batch(u, gg, nn); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            out = eng.infer_batch(u, gg, nn, out)
        torch.cuda.synchronize()
        dt_s = (time.time() - t0) / 5
        print(f"timing {dt} B={B}: {dt_s * 1e3:.3f} ms/batch  {B * 80000 / dt_s:.3e} samples/s")
