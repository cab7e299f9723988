"""Long-form configuration (BASELINE configs[3]): T_in=400, max_iters=400. python tools/time_cfg4.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
hp = HParams(); eng = TacoEngine(init_weights(hp, seed=1234), hp)
for B in (1, 32):
    enc = torch.randn(B, 400, 256, device='cuda') * 0.5
    for mode in (2, 1):
        eng.set_decode_graph(mode)
        eng.decode(enc, max_iters=400); torch.cuda.synchronize(); eng.decode_status()
        t = time.perf_counter()
        for _ in range(3): eng.decode(enc, max_iters=400)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 3 * 1e3
        print('B=%d T_in=400 S=400 mode %d: %.2f ms (%.1f us/step)' % (B, mode, ms, ms * 1e3 / 400), flush=True)
