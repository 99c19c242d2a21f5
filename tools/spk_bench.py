"""Times the speaker-encoder launches (qvc_speaker_embed) against torch.nn.LSTM on the same GPU.
usage: python tools/spk_bench.py [utterances ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickvc_official_amd as q  # noqa: E402
from quickvc_official_amd.synth import make_synthetic_mel, make_synthetic_state_dict  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
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
    model = q.SynthesizerTrn(641, 32, **q.DEFAULT_MODEL_CONFIG)
    model.load_state_dict(make_synthetic_state_dict(model, 1234))
    model = model.cuda().eval()
    for U in [int(x) for x in sys.argv[1:]] or [1, 32]:
        mel = torch.cat([make_synthetic_mel(250, 80, seed=u) for u in range(U)], 0).cuda()
        g = model.speaker_embed(mel)
        ref = torch.cat([model.enc_spk.embed_utterance(m[None].transpose(1, 2)) for m in mel], 0)
        err = ((g - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
        t_hip = timed(lambda: model.speaker_embed(mel))
        t_torch = timed(lambda: model.enc_spk.embed_utterance(mel[:1].transpose(1, 2)), n=5)
        print(f"U={U:3d} F=250: HIP {t_hip:.3f} ms for all utterances, torch.nn.LSTM {t_torch:.3f} ms per utterance, "
              f"rel err vs torch fp32 {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
