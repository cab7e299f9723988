"""post-net proj1-shaped GEMM (K = 3072, N = 256) at row counts that put 2, 3 and 6 workgroups on every CU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron_amd import _lib
import torch
lib = _lib.load_library()
Cin, taps, N = 1024, 3, 256
p = lambda t: t.data_ptr()
for M in (32000, 32768, 49152, 98304):
    X = torch.randn(M, Cin, device='cuda'); W = torch.randn(taps * Cin, N, device='cuda') * 0.05
    bias = torch.randn(N, device='cuda'); C = torch.empty(M, N, device='cuda')
    def run(n):
        for _ in range(n):
            assert lib.taco_debug_conv_gemm(p(X), Cin, M, M // 32, Cin, taps, 1, p(W), N, N, p(bias), None, None, None, N, p(C), N, 1, 0, None) == 0
    run(200); torch.cuda.synchronize()
    t = time.perf_counter(); n = 1000
    run(n); torch.cuda.synchronize()
    el = time.perf_counter() - t
    print('M=%6d (%4d workgroups): %.1f us, %.1f TFLOP/s' % (M, (M + 127) // 128 * 2, el / n * 1e6, 2.0 * M * Cin * taps * N * n / el / 1e12), flush=True)
