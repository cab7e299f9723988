"""Diagnostic: where one workgroup of gl_iter2 spends its cycles (needs the -DGL_STAMPS build of the library):
   TACO_LIB=tacotron_amd/libtaco_v_stamps.so (tools/build_variant.sh stamps -DGL_STAMPS -DGL_STAMP_WG=1016) python tools/gl_stamps.py [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ['TACO_LIB'])
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
hp = HParams(); eng = TacoEngine(init_weights(hp, seed=1234), hp)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lin = torch.rand(B, 1000, hp.num_freq, device='cuda')
for rep in range(3):
    eng.griffin_lim(lin, iters=8); torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 64)()
    assert eng.lib.taco_debug_gl_stamps(st) == 0
    t = [int(x) for x in st]
    names = ['start', 'tables', 'gather issued', 'barrier'] + ['p1.' + n for n in ('begin', 's1', 's2', 's3', 's4', 's5', 's6', 'end')]
    idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]
    names += ['', ''] + ['p2.' + n for n in ('begin', 's1', 's2', 's3', 's4', 's5', 's6', 'end')]
    idx += [14, 15, 16, 17, 18, 19, 20, 21]
    names += ['r1.B2', 'r1.pass', 'r1.B3', 'r2.B2', 'r2.pass', 'r2.end']
    idx += [24, 25, 26, 27, 28, 29]
    if os.environ.get('GL_STAMPS_PASS'):
        names += ['r1.issued', 'r1.landed', 'r2.issued', 'r2.landed']; idx += [30, 31, 32, 33]
    names = [n for n in names if n]
    print(' '.join('%s %d' % (n, t[i] - t[0]) for n, i in zip(names, idx)), flush=True)
