"""Sample the GPU clock (rocm-smi) while the largest post-net GEMM runs back to back: python tools/clock_under_load.py"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron_amd import _lib
import torch
lib = _lib.load_library()
M, Cin, taps, N = 32000, 1024, 3, 256
X = torch.randn(M, Cin, device='cuda'); W = torch.randn(taps * Cin, N, device='cuda') * 0.05
bias = torch.randn(N, device='cuda'); C = torch.empty(M, N, device='cuda')
p = lambda t: t.data_ptr()
stop = False
samples = []
def sampler():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=10).stdout
            samples.append([l.strip() for l in out.splitlines() if 'sclk' in l or 'mclk' in l][:2])
        except Exception as e:
            samples.append([repr(e)])
        time.sleep(0.3)
def run(n):
    for _ in range(n):
        assert lib.taco_debug_conv_gemm(p(X), Cin, M, 1000, Cin, taps, 1, p(W), N, N, p(bias), None, None, None, N, p(C), N, 1, 0, None) == 0
run(20); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t = time.perf_counter(); n = 6000
run(n); torch.cuda.synchronize()
el = time.perf_counter() - t
stop = True; th.join()
print('%.1f us per GEMM, %.1f TFLOP/s over %.1f s' % (el / n * 1e6, 2.0 * M * Cin * taps * N * n / el / 1e12, el))
for s in samples[:3] + samples[-3:]: print(s)
