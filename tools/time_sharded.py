#!/usr/bin/env python
"""Times the PRODUCT sharded call (SynthesisPipeline.synthesize_sharded: pinned result tensors, downloads on the
engines' streams) against the rate bench.py reports with outputs left on the device. One GPU:
    python tools/time_sharded.py [--n 256] [--batch 32] [--depth 6] [--mel-linear]
Prints one JSON line: frames/s of the call (wall clock, host arrays delivered), per key set."""
import argparse
import json
import os
import sys
import time

if int(os.environ.get('GPU_MAX_HW_QUEUES') or 0) < 8:
    os.environ['GPU_MAX_HW_QUEUES'] = '8'
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=256)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--depth', type=int, default=6)
    ap.add_argument('--t-in', type=int, default=150)
    ap.add_argument('--max-iters', type=int, default=200)
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args()
    import torch
    from tacotron_amd.data.synthetic import synth_ids
    from tacotron_amd.hparams import HParams
    from tacotron_amd.pipeline import SynthesisPipeline
    from tacotron_amd.weights import init_weights
    hp = HParams()
    hp.max_iters = args.max_iters
    w = init_weights(hp, seed=1234)
    ids, lens = synth_ids(np.random.default_rng(0), args.n, args.t_in, ragged=False)
    pipe = SynthesisPipeline(w, hp, depth=args.depth, devices=[0])
    out = {'n': args.n, 'batch': args.batch, 'depth': args.depth}
    frames = args.n * args.max_iters * hp.outputs_per_step
    for name, to_wav, keys in (('to_wav_keys_wav', True, None), ('mel_linear_keys_default', False, None),
                               ('mel_linear_keys_mel', False, ('mel',))):
        best = None
        for _ in range(args.reps + 1):                       # (first call: workspaces, pinned pools)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = pipe.synthesize_sharded(ids, lens, batch_size=args.batch, to_wav=to_wav, keys=keys)
            el = time.perf_counter() - t0
            assert len(r) == args.n
            best = el if best is None else min(best, el)
            del r
        out[name] = {'frames_per_s': frames / best, 'ms_per_batch': 1e3 * best / (args.n / args.batch)}
    pipe.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
