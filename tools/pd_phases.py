"""With the -DPD_PHASE_STAMPS build (tacotron_amd/libtaco_hip_stamps_phase.so): where a persistent decode's time goes
with several batches in flight - from workgroup 0's start to the end of step 0 (all workgroups resident, weights and
keys staged), and the remaining steps. python tools/pd_phases.py [streams] [rounds]"""
import os, sys
if int(os.environ.get('GPU_MAX_HW_QUEUES', '0') or 0) < 8:
    os.environ['GPU_MAX_HW_QUEUES'] = '8'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libtaco_hip_stamps_phase.so')
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
from tacotron_amd.data.synthetic import synth_ids
n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
hp = HParams(); w = init_weights(hp, seed=1234)
engines = [TacoEngine(w, hp) for _ in range(n_eng)]
for e in engines: e.detached = True
ids, lens = synth_ids(np.random.default_rng(0), 32, 150, ragged=False)
ids = torch.from_numpy(ids).cuda(); lens = torch.from_numpy(lens).cuda()
outs = [{} for _ in engines]
for r in range(rounds):
    for k in range(3 * n_eng):
        engines[k % n_eng].synthesize(ids, lens, out=outs[k % n_eng])
    torch.cuda.synchronize()
    for e in engines: e.decode_status()       # prints the phases of each engine's LAST decode
