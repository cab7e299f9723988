"""Writes tests/golden/oracle_small.npz: outputs of the fp64 numpy oracle on a tiny seeded case
(B=2, T_in=7/5, 3 decoder steps, 2 Griffin-Lim iterations). These vectors come from the oracle
itself (the reference cannot run here: TF 1.15 / librosa absent), so they pin the oracle against
regressions, not against the reference - "parity unpinned", see oracle/taco_oracle.py.
    python tests/golden/make_oracle_goldens.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import taco_oracle as O  # noqa: E402
from tacotron_amd.hparams import HParams  # noqa: E402
from tacotron_amd.weights import init_weights  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'oracle_small.npz')


def case():
    hp = HParams()
    w = init_weights(hp, seed=1234, randomize_bn=True, randomize_bias=True)
    ids = np.array([[35, 32, 39, 39, 42, 58, 1], [50, 42, 45, 39, 1, 0, 0]], np.int32)
    lens = np.array([7, 5], np.int32)
    return hp, w, ids, lens


def main():
    hp, w, ids, lens = case()
    r = O.tacotron_forward(ids, lens, w, hp, max_iters=3)
    lin = np.random.default_rng(0).uniform(0.3, 1.0, size=(6, 1025))
    wav = O.spectrogram_tensorflow_inv(lin, hp, iters=2)
    np.savez_compressed(OUT, ids=ids, lens=lens, enc_out=r['enc_out'].astype(np.float32),
                        mel=r['mel'].astype(np.float32), linear=r['linear'][:, :, ::16].astype(np.float32),
                        alignments=r['alignments'].astype(np.float32), gl_linear=lin.astype(np.float32),
                        gl_wav=wav.astype(np.float32))
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
