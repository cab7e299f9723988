"""Time decode only with an alternative library build: TACO_LIB=path python tools/time_decode.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron_amd import _lib
if os.environ.get('TACO_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['TACO_LIB'])
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
hp = HParams(); eng = TacoEngine(init_weights(hp, seed=1234), hp)
for B in [int(a) for a in sys.argv[1:]] or [32]:
    enc = torch.randn(B, 150, 256, device='cuda') * 0.5
    res = {}
    for mode in (1, 2):
        eng.set_decode_graph(mode)
        m, a, _ = eng.decode(enc); torch.cuda.synchronize(); eng.decode_status()
        t = time.perf_counter()
        for _ in range(5): eng.decode(enc)
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t) / 5 * 1e3, m.clone())
    print('B=%d decode: graph %.2f ms (%.1f us/step)  persistent %.2f ms (%.1f us/step)  rel diff %.1e' % (
        B, res[1][0], res[1][0] * 5, res[2][0], res[2][0] * 5,
        float((res[1][1] - res[2][1]).abs().sum() / res[1][1].abs().sum())), flush=True)
