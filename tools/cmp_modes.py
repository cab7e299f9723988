import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
hp = HParams()
eng = TacoEngine(init_weights(hp, seed=1234, randomize_bias=True), hp)
B, T_in, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
enc = torch.randn(B, T_in, 256, device='cuda') * 0.5
eng.set_decode_graph(1); m1, a1, _ = eng.decode(enc, max_iters=S)
eng.set_decode_graph(2); m2, a2, _ = eng.decode(enc, max_iters=S)
torch.cuda.synchronize(); eng.decode_status()
m1 = m1.cpu().numpy().reshape(B, S, 400); m2 = m2.cpu().numpy().reshape(B, S, 400)
a1 = a1.cpu().numpy(); a2 = a2.cpu().numpy()
for s in range(min(S, 4)):
    print('step', s, 'align err %.2e' % (np.abs(a1[:, :, s] - a2[:, :, s]).sum() / np.abs(a1[:, :, s]).sum()),
          'out err per 80-block', ['%.1e' % (np.abs(m1[:, s, k:k+80] - m2[:, s, k:k+80]).sum() / np.abs(m1[:, s, k:k+80]).sum()) for k in range(0, 400, 80)])
print('b0 step0 out[:8] ref', m1[0, 0, :8], '\n got', m2[0, 0, :8])
print('per-b err step0', ['%.1e' % (np.abs(m1[b, 0] - m2[b, 0]).sum() / np.abs(m1[b, 0]).sum()) for b in range(B)])
