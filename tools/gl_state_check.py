"""Griffin-Lim with run records against the frame state (TACO_GL_FRAME_STATE=1), bit for bit, over batch sizes, lengths
(whole runs, short last runs, the one- and two-frame last runs that keep the frame state) and run lengths; then times.
python tools/gl_state_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights

hp = HParams()
eng = TacoEngine(init_weights(hp, seed=1234), hp)


def run(lin, iters, frame_state, runlen=None):
    if frame_state is None: os.environ.pop('TACO_GL_FRAME_STATE', None)
    else: os.environ['TACO_GL_FRAME_STATE'] = '1' if frame_state else '0'
    if runlen: os.environ['TACO_GL_RUN'] = str(runlen)
    else: os.environ.pop('TACO_GL_RUN', None)
    w = eng.griffin_lim(lin, iters=iters)
    torch.cuda.synchronize()
    return w.cpu().numpy()


bad = 0
g = torch.Generator(device='cuda').manual_seed(7)
for B, T, iters in [(1, 16, 2), (1, 3, 3), (2, 19, 3), (3, 40, 4), (1, 17, 3), (2, 18, 3), (5, 200, 5), (2, 1000, 6),
                    (33, 100, 3), (1, 8, 1), (1, 11, 2), (4, 27, 3), (1, 1000, 60)]:
    lin = torch.rand(B, T, hp.num_freq, device='cuda', generator=g)
    ref = run(lin, iters, True)
    for rl in (None, 8, 16):
        got = run(lin, iters, False, rl)
        same = np.array_equal(ref, got)
        d = float(np.abs(ref.astype(np.float64) - got).max())
        print('B=%d T=%d iters=%d run=%s: %s (max diff %.3e, finite %s)' % (B, T, iters, rl, 'identical' if same else 'DIFFERENT', d,
                                                                         bool(np.isfinite(got).all())), flush=True)
        bad += not same
print('mismatches:', bad, flush=True)
if bad: sys.exit(1)

for B in (32, 16, 8, 4, 2, 1):
    lin = torch.rand(B, 1000, hp.num_freq, device='cuda', generator=g)
    for fs, rl in ((True, None), (False, None), (False, 8), (False, 16), (None, None)):
        run(lin, 60, fs, rl)
        t = time.perf_counter()
        for _ in range(3): eng.griffin_lim(lin, iters=60)
        torch.cuda.synchronize()
        print('B=%d frame_state=%s run=%s: %.2f ms' % (B, fs, rl, (time.perf_counter() - t) / 3 * 1e3), flush=True)
