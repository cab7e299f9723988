"""What slows the persistent decoder under load: N decodes side by side (no other work), and one decode beside post-net
GEMM streams. Prints wall ms per decode (200 steps) for each case.   python tools/decode_contention.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import _lib
if os.environ.get('TACO_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['TACO_LIB'])
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights

hp = HParams(); w = init_weights(hp, seed=1234)
engs = [TacoEngine(w, hp) for _ in range(4)]
for e in engs: e.detached = True
enc = torch.randn(32, 150, 256, device='cuda') * 0.5
mel = torch.randn(32, 1000, hp.num_mels, device='cuda')
for e in engs:
    e.decode(enc); e.postnet(mel)
torch.cuda.synchronize()


def run(n_dec, n_post, reps=3):
    torch.cuda.synchronize()
    if n_dec: engs[0].timing_begin('dec_persistent')      # HIP events around engine 0's persistent launches
    t = time.perf_counter()
    for _ in range(reps):
        for i in range(n_dec): engs[i].decode(enc)
        for i in range(n_post):
            for _ in range(4): engs[3 - i].postnet(mel)          # ~4 x 2.2 ms of GEMM + scan per decode
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / reps * 1e3
    if n_dec:
        n, ms = engs[0].timing_end()
        return '%.2f ms per round, decode kernel %.2f ms (%.1f us per step)' % (wall, ms / max(n, 1), ms / max(n, 1) * 5)
    return '%.2f ms per round' % wall


cases = [(1, 0), (3, 0), (1, 1)] if os.environ.get('QUICK') else [(1, 0), (2, 0), (3, 0), (4, 0), (1, 1), (1, 2), (2, 2), (0, 1)]
for n_dec, n_post in cases:
    print('decodes %d  post-net streams %d: %s' % (n_dec, n_post, run(n_dec, n_post)), flush=True)
for e in engs: e.decode_status()
