"""What the round-3 counter passes run (rocprofv3 --pmc ... -- python3 tools/pmc_targets.py): a few repetitions of
  * taco_debug_conv_gemm on post-net proj_1, the linear dense, the encoder's proj_1 (M = 4800) and an encoder bank shape,
  * one full synthesis at B = 32 (cbhg_tail_kernel of encoder and post-net, bigru_scan, ONE dec_persistent launch),
so that tools/pmc_any.py can print SQ counters per kernel. Nothing is timed here."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tacotron_amd import _lib
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
from tacotron_amd.data.synthetic import synth_ids

lib = _lib.load_library()
B = 32
p = lambda t: t.data_ptr()
shapes = [('post proj1', 1000, 1024, 3, 256, 1), ('linear dense', 1000, 256, 1, 1025, 0),
          ('enc proj1', 150, 2048, 3, 128, 1), ('enc bank k=8', 150, 128, 8, 128, 0)]
for name, T, Cin, taps, N, pool in shapes:
    M = B * T
    ldw = (N + 127) // 128 * 128
    X = torch.randn(M, Cin, device='cuda')
    W = torch.randn(taps * Cin, ldw, device='cuda') * 0.05
    bias = torch.randn(ldw, device='cuda')
    C = torch.empty(M, N, device='cuda')
    for _ in range(3):
        rc = lib.taco_debug_conv_gemm(p(X), Cin, M, T, Cin, taps, (taps - 1) // 2, p(W), ldw, N, p(bias), None, None,
                                      None, N, p(C), N, 1, pool, None)
        assert rc == 0, lib.taco_last_error()
    torch.cuda.synchronize()
hp = HParams()
eng = TacoEngine(init_weights(hp, seed=1234), hp)
ids, lens = synth_ids(np.random.default_rng(0), B, 150, ragged=False)
o = {}
for _ in range(2):
    eng.synthesize(ids, lens, max_iters=200, to_wav=False, out=o)
    torch.cuda.synchronize()
eng.decode_status()
eng.close()
print('done')
