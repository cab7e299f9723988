"""What a plain streaming kernel reaches on this GPU for Griffin-Lim's read : write mix (2 : 1) and volume:
torch.add(a, b, out=c) on three 154 MB buffers (462 MB per pass, against 448 MB per gl_iter launch)."""
import torch
n = 32000 * 1200
a = torch.rand(n, device='cuda'); b = torch.rand(n, device='cuda'); c = torch.empty(n, device='cuda')
bufs = [(torch.rand(n, device='cuda'), torch.rand(n, device='cuda'), torch.empty(n, device='cuda')) for _ in range(2)]
for _ in range(3):
    torch.add(a, b, out=c)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 60
for i in range(reps):
    x, y, z = bufs[i % 2]            # alternate working sets (924 MB in all: nothing survives in the 256 MB cache)
    torch.add(x, y, out=z)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print('add 2R:1W %.0f MB per pass: %.1f us = %.2f TB/s' % (3 * n * 4 / 1e6, us, 3 * n * 4 / us / 1e6))
e0.record()
for i in range(reps):
    x, y, z = bufs[i % 2]
    z.copy_(x)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print('copy 1R:1W %.0f MB per pass: %.1f us = %.2f TB/s' % (2 * n * 4 / 1e6, us, 2 * n * 4 / us / 1e6))
