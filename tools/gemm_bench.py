"""Time taco_debug_conv_gemm on the GEMM shapes of one B=32 / T_in=150 / S=200 synthesis step.
   TACO_LIB=path python tools/gemm_bench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron_amd import _lib
if os.environ.get('TACO_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['TACO_LIB'])
import torch
lib = _lib.load_library()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 32
shapes = []   # name, T, Cin, taps, N, pool, count
for k in (1, 4, 8, 16):
    shapes.append(('enc bank k=%d' % k, 150, 128, k, 128, 0))
shapes += [('enc proj1', 150, 2048, 3, 128, 1), ('enc proj2', 150, 128, 3, 128, 0), ('enc highway', 150, 128, 1, 256, 0),
           ('enc rnn_in', 150, 128, 1, 768, 0), ('keys', 150, 256, 1, 256, 0)]
for k in (1, 4, 8):
    shapes.append(('post bank k=%d' % k, 1000, 80, k, 128, 0))
shapes += [('post proj1', 1000, 1024, 3, 256, 1), ('post proj2', 1000, 256, 3, 80, 0), ('post widthfix', 1000, 80, 1, 128, 0),
           ('post highway', 1000, 128, 1, 256, 0), ('post rnn_in', 1000, 128, 1, 768, 0), ('linear dense', 1000, 256, 1, 1025, 0)]
p = lambda t: t.data_ptr()
tot = 0.0
for name, T, Cin, taps, N, pool in shapes:
    M = B * T
    ldw = (N + 127) // 128 * 128
    X = torch.randn(M, Cin, device='cuda')
    W = torch.randn(taps * Cin, ldw, device='cuda') * 0.05
    bias = torch.randn(ldw, device='cuda')
    C = torch.empty(M, N, device='cuda')
    def run():
        rc = lib.taco_debug_conv_gemm(p(X), Cin, M, T, Cin, taps, (taps - 1) // 2, p(W), ldw, N, p(bias), None, None,
                                      None, N, p(C), N, 1, pool, None)
        assert rc == 0, lib.taco_last_error()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * M * taps * Cin * N
    print('%-16s M=%6d K=%5d N=%5d  %8.1f us  %6.1f TF' % (name, M, taps * Cin, N, us, fl / us / 1e6), flush=True)
