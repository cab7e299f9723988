"""Time the stages of the engine separately (GPU). python tools/time_stages.py [B ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd.engine import TacoEngine
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights
from tacotron_amd.data.synthetic import synth_ids

hp = HParams()
eng = TacoEngine(init_weights(hp, seed=1234), hp)
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for B in [int(a) for a in sys.argv[1:]] or [32]:
    ids, lens = synth_ids(np.random.default_rng(0), B, 150, ragged=False)
    ids_d, lens_d = torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()
    enc = eng.encode(ids_d, lens_d)
    mel, al, st = eng.decode(enc)
    lin = eng.postnet(mel)
    t_enc = timeit(lambda: eng.encode(ids_d, lens_d))
    t_dec = timeit(lambda: eng.decode(enc))
    eng.set_decode_graph(1); t_dec_g = timeit(lambda: eng.decode(enc)); eng.set_decode_graph(0); t_dec_e = timeit(lambda: eng.decode(enc)); eng.set_decode_graph(2)
    t_post = timeit(lambda: eng.postnet(mel))
    t_gl = timeit(lambda: eng.griffin_lim(lin[:, :, :]), n=2)
    out = {}
    t_all = timeit(lambda: eng.synthesize(ids_d, lens_d, out=out))
    print('B=%d: encode %.2f ms | decode persistent %.2f ms (%.1f us/step) graph %.2f eager %.2f ms | postnet %.2f ms | GL60 %.1f ms | synthesize %.2f ms -> %.0f frames/s'
          % (B, t_enc, t_dec, t_dec * 1e3 / 200, t_dec_g, t_dec_e, t_post, t_gl, t_all, B * 1000 / t_all * 1e3), flush=True)
eng.close()
