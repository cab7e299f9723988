#!/usr/bin/env python
"""Decode loop alone per mode (2 = persistent, 32 workgroups per tile; 3 = tile kernel, one workgroup per tile):
ms per decode and us per decoder step at B=32 / T_in=150 / S=200, one decode alone and N side by side.
    python tools/time_tile.py [B] [N]"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
hp = HParams()
w = init_weights(hp, seed=1234)
engs = [TacoEngine(w, hp) for _ in range(N)]
for e in engs:
    e.detached = True
enc = torch.from_numpy((0.5 * np.random.default_rng(0).standard_normal((B, 150, 256))).astype(np.float32)).cuda()
S = 200
for mode in (2, 3):
    for e in engs:
        e.set_decode_graph(mode)
    for e in engs:
        e.decode(enc, max_iters=S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        engs[0].decode(enc, max_iters=S)
        torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for e in engs:
        e.decode(enc, max_iters=S)
    torch.cuda.synchronize()
    side = time.perf_counter() - t0
    print('mode %d B=%d: alone %.2f ms (%.1f us/step); %d side by side %.2f ms (%.2f ms per decode)' % (
        mode, B, 1e3 * alone, 1e6 * alone / S, N, 1e3 * side, 1e3 * side / N), flush=True)
for e in engs:
    e.decode_status()
    e.close()
