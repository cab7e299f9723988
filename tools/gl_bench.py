"""Time Griffin-Lim alone: python tools/gl_bench.py [B] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron_amd import _lib
if os.environ.get('TACO_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['TACO_LIB'])
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
hp = HParams(); eng = TacoEngine(init_weights(hp, seed=1234), hp)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
lin = torch.rand(B, 1000, hp.num_freq, device='cuda')
w = eng.griffin_lim(lin, iters=iters); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3): w = eng.griffin_lim(lin, iters=iters)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / 3 * 1e3
print('GL B=%d T=1000 iters=%d: %.2f ms (%.3f ms per iteration), checksum %.6e' % (B, iters, ms, ms / (iters + 1), float(w.double().abs().sum())), flush=True)
