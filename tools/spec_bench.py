"""Time the analysis side (taco_spectrogram): python tools/spec_bench.py [B] [seconds of audio per utterance]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
hp = HParams(); eng = TacoEngine(init_weights(hp, seed=1234), hp)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 12.5
n = int(secs * hp.sample_rate)
wav = torch.randn(B, n, device='cuda') * 0.1
for what, kw in (('linear+mel', {}), ('mel only', {'want_linear': False}), ('linear only', {'want_mel': False})):
    eng.spectrogram(wav, **kw); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): eng.spectrogram(wav, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 5 * 1e3
    print('spectrogram %-11s B=%d x %.1f s: %.2f ms -> %.0f x real time, %.2f M frames/s' % (
        what, B, secs, ms, B * secs / (ms * 1e-3), B * (1 + n // 300) / ms * 1e-3), flush=True)
