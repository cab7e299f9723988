"""Per-kernel-family launches and ms of one batch run alone, plus stage times. python tools/time_families.py [B]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron_amd import _lib
if os.environ.get('TACO_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['TACO_LIB'])
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
from tacotron_amd.data.synthetic import synth_ids
hp = HParams()
eng = TacoEngine(init_weights(hp, seed=1234), hp)
eng.set_bigru_mode(int(os.environ.get('TACO_BIGRU_MODE', '0')))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ids, lens = synth_ids(np.random.default_rng(0), B, 150, ragged=False)
ids_d, lens_d = torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
enc = eng.encode(ids_d, lens_d); mel, al, st = eng.decode(enc); lin = eng.postnet(mel)
print('encode %.3f ms  decode %.3f ms  postnet %.3f ms' % (timeit(lambda: eng.encode(ids_d, lens_d)),
      timeit(lambda: eng.decode(enc)), timeit(lambda: eng.postnet(mel))), flush=True)
for stage, f in (('encode', lambda: eng.encode(ids_d, lens_d)), ('postnet', lambda: eng.postnet(mel))):
    for k in ('conv_gemm', 'highway_combine', 'bigru_scan'):
        eng.timing_begin(k); f(); n, ms = eng.timing_end()
        print('  %-8s %-16s launches %3d  %.3f ms' % (stage, k, n, ms), flush=True)
eng.close()
