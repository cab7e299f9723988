"""Run records against the frame state over ~80 random and edge shapes (B 1..300, T 3..2000), both run lengths and the
default choice, bit for bit.   python tools/gl_fuzz.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
hp = HParams(); eng = TacoEngine(init_weights(hp, seed=1234), hp)
rng = np.random.default_rng(5)
bad = 0
shapes = [(int(rng.integers(1, 41)), int(rng.integers(3, 120)), int(rng.integers(1, 4))) for _ in range(70)]
shapes += [(300, 4, 2), (270, 7, 2), (64, 33, 2), (17, 257, 2), (3, 999, 2), (2, 1001, 2), (1, 1002, 2), (1, 2000, 2)]
for B, T, it in shapes:
    lin = torch.rand(B, T, hp.num_freq, device='cuda')
    os.environ['TACO_GL_FRAME_STATE'] = '1'; os.environ.pop('TACO_GL_RUN', None)
    ref = eng.griffin_lim(lin, iters=it).cpu().numpy()
    for fs, rl in (('0', '8'), ('0', '16'), (None, None)):
        if fs is None: os.environ.pop('TACO_GL_FRAME_STATE', None)
        else: os.environ['TACO_GL_FRAME_STATE'] = fs
        if rl: os.environ['TACO_GL_RUN'] = rl
        else: os.environ.pop('TACO_GL_RUN', None)
        got = eng.griffin_lim(lin, iters=it).cpu().numpy()
        if not np.array_equal(ref, got) or not np.isfinite(got).all():
            bad += 1; print('MISMATCH', B, T, it, fs, rl, flush=True)
print('shapes', len(shapes), 'mismatches', bad, flush=True)
