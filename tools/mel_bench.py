"""Times qvc_wave_to_mel against the torch restatement (torch.stft on the same GPU).  usage: python tools/mel_bench.py [utterances ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickvc_official_amd  # noqa: E402,F401
from quickvc_official_amd.frontend import MelFrontend, mel_basis  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import qvc_oracle as oracle  # noqa: E402  (developer tool: the comparator lives with the oracle)

BASIS = torch.from_numpy(mel_basis(16000, 1280, 80, 0.0, None))


def wave_to_mel(wave, *_):
    return oracle.wave_to_mel(wave.cpu(), BASIS, 1280, 320, 1280).to(wave.device)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    fe = MelFrontend(1280, 80, 16000, 320, 1280, 0.0, None)
    for U in [int(x) for x in sys.argv[1:]] or [1, 32]:
        wave = (torch.randn(U, 80000, device="cuda") * 0.1).clamp(-1, 1)
        mel = fe(wave)
        ref = wave_to_mel(wave, 1280, 80, 16000, 320, 1280, 0.0, None)
        err = (mel - ref).abs().max().item()
        print(f"U={U:3d} x 5 s: HIP {timed(lambda: fe(wave)):.3f} ms, torch (stft + matmul + log) "
              f"{timed(lambda: wave_to_mel(wave, 1280, 80, 16000, 320, 1280, 0.0, None)):.3f} ms, max |dlogmel| {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
