"""One B=32 synthesis step (optionally to the waveform), repeated: the command to put under rocprofv3 --kernel-trace.
   python tools/one_step.py [reps] [to_wav]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
from tacotron_amd.data.synthetic import synth_ids
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
to_wav = len(sys.argv) > 2 and sys.argv[2] == '1'
hp = HParams(); eng = TacoEngine(init_weights(hp, seed=1234), hp)
ids, lens = synth_ids(np.random.default_rng(0), 32, 150, ragged=False)
out = {}
for _ in range(reps):
    eng.synthesize(ids, lens, to_wav=to_wav, out=out)
    torch.cuda.synchronize()
eng.decode_status()
