"""Fixed cost of one persistent decode (weights into registers, key / value staging, region memset, finalisation):
decode time at a few step counts; the intercept is the overhead. python tools/decode_overhead.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
hp = HParams(); eng = TacoEngine(init_weights(hp, seed=1234), hp)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
enc = torch.randn(B, 150, 256, device='cuda') * 0.5
res = []
for S in (1, 10, 50, 200):
    eng.decode(enc, max_iters=S); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): eng.decode(enc, max_iters=S)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 10 * 1e3
    res.append((S, ms))
    print('B=%d S=%3d: %.3f ms' % (B, S, ms), flush=True)
slope = (res[-1][1] - res[1][1]) / (res[-1][0] - res[1][0])
print('slope %.1f us per step, intercept %.0f us' % (slope * 1e3, (res[1][1] - slope * res[1][0]) * 1e3))
