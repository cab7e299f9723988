"""Run one persistent decode with the -DPD_STAMPS build of the library and print workgroup 0's stage stamps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libtaco_hip_stamps%s.so' % os.environ.get('PD_WG', '0'))
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
hp = HParams()
eng = TacoEngine(init_weights(hp, seed=1234), hp)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
enc = torch.randn(B, 150, 256, device='cuda')
for _ in range(3):
    eng.decode(enc, max_iters=20)
    torch.cuda.synchronize()
    eng.decode_status()
