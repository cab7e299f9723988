"""Generate tests/golden/hparams_goldens.json and tests/golden/host_dsp_goldens.npz from the REFERENCE's own files,
in the build container (the reference does not exist on the GPU box; only the data written here travels).

    python tests/golden/make_refpin_goldens.py

What is executed from /root/reference:
  * hparams.py:7-68   - the `HParams(...)` keyword literals. `tensorflow.contrib.training.HParams` is not installed; an
                        inert class that stores its keyword arguments and answers `.get(name)` is registered under that
                        module path for the import (as `unidecode` was for the text goldens): every VALUE written below
                        is the reference file's own literal.
  * tools/audio.py    - the numpy/scipy-only helpers: save_wav's scaling (:15-17), find_endpoint (:81-88), pre_emphasis
                        (:91-109), pre_emphasis_inv (:111-116), _stft_params (:150-164), _amp_to_db (:212-217),
                        _db_to_amp (:219-224), _normalize (:254-264), _normalize_inv (:266-274). `tensorflow` and
                        `librosa` are imported at the top of that file but not used by these functions; empty modules
                        stand under those names, and `librosa.output.write_wav` is a recorder that keeps the array
                        save_wav hands it (so the fixture holds the int16 samples the reference would have written).
Nothing that needs TF or librosa arithmetic is called (the Griffin-Lim / STFT / mel functions stay unpinned).
"""
import json
import os
import sys
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


class _RecordedHParams:
    """Inert stand-in for tf.contrib.training.HParams: keeps the keyword literals, answers get()."""

    def __init__(self, **kw):
        self._kw = dict(kw)

    def get(self, name, default=None):
        return self._kw.get(name, default)

    def values(self):
        return dict(self._kw)


def _register_stubs(written):
    tf = types.ModuleType('tensorflow')
    contrib = types.ModuleType('tensorflow.contrib')
    training = types.ModuleType('tensorflow.contrib.training')
    training.HParams = _RecordedHParams
    contrib.training = training
    tf.contrib = contrib
    sys.modules.setdefault('tensorflow', tf)
    sys.modules.setdefault('tensorflow.contrib', contrib)
    sys.modules.setdefault('tensorflow.contrib.training', training)
    librosa = types.ModuleType('librosa')
    output = types.ModuleType('librosa.output')
    output.write_wav = lambda path, y, sr, **k: written.append((path, np.array(y), sr))
    librosa.output = output
    sys.modules.setdefault('librosa', librosa)
    sys.modules.setdefault('librosa.output', output)
    sys.modules.setdefault('unidecode', types.SimpleNamespace(unidecode=lambda s: s))


def main():
    written = []
    _register_stubs(written)
    sys.path.insert(0, REF)
    import hparams as ref_hparams                      # /root/reference/hparams.py
    from tools import audio as ref_audio               # /root/reference/tools/audio.py

    vals = ref_hparams.hparams.values()
    with open(os.path.join(HERE, 'hparams_goldens.json'), 'w') as f:
        json.dump({'source': 'keyword literals of /root/reference/hparams.py:7-68 (imported with an inert HParams class)',
                   'values': vals,
                   'stft_params': list(ref_audio._stft_params())}, f, indent=1, sort_keys=True)

    rng = np.random.default_rng(20240)
    g = {}
    # pre-emphasis and its inverse (scipy lfilter; float64)
    x = rng.standard_normal(4000)
    g['pre_x'] = x
    g['pre_emphasis'] = ref_audio.pre_emphasis(x)
    g['pre_emphasis_inv'] = ref_audio.pre_emphasis_inv(x)
    x32 = (0.1 * rng.standard_normal(3000)).astype(np.float32)   # synthesizer.py:44 passes the float32 session output
    g['pre_x32'] = x32
    g['pre_emphasis_inv_x32'] = ref_audio.pre_emphasis_inv(x32)
    # find_endpoint: signed max over 0.8 s windows every 0.2 s (audio.py:81-88)
    t = 2 * np.pi * (np.arange(30000) % 48) / 48.0        # periodic in samples: the fixture compresses
    ep_cases = [
        np.concatenate([0.5 * np.sin(t), np.zeros(40000)]),                         # speech then silence
        np.ones(50000),                                                             # never silent
        np.concatenate([0.5 * np.sin(t), -0.3 * np.ones(30000)]),                   # loud but negative: signed max < thr
        np.concatenate([np.zeros(30000), 0.5 * np.sin(t)]),                         # leading silence
        0.009 * np.ones(24000 + 4800 + 1),                                          # just long enough for one window
        0.009 * np.ones(24000),                                                     # too short for any window
        np.concatenate([0.5 * np.sin(t)[:20000], 0.0099 * np.ones(19200), 0.5 * np.sin(t)[:20000]]),
        (rng.standard_normal(45000) * np.concatenate([np.ones(13333), 1e-3 * np.ones(31667)])).astype(np.float32),
    ]
    g['n_endpoint_cases'] = np.int64(len(ep_cases))
    for i, w in enumerate(ep_cases):
        g['ep_wav_%d' % i] = w
        g['ep_out_%d' % i] = np.int64(ref_audio.find_endpoint(w))
    w = ep_cases[7]
    g['ep_out_7_m30_05'] = np.int64(ref_audio.find_endpoint(w, threshold_db=-30, min_silence_sec=0.5))
    # dB <-> amplitude, normalisation
    d = np.concatenate([np.linspace(-140.0, 40.0, 181), rng.uniform(-120, 20, 64)])
    g['db_in'] = d
    g['db_to_amp'] = ref_audio._db_to_amp(d)
    a = np.concatenate([[0.0, 1e-7, 1e-5, 1.0], np.abs(rng.standard_normal(64)) * 3])
    g['amp_in'] = a
    g['amp_to_db'] = ref_audio._amp_to_db(a)
    s = np.concatenate([np.linspace(-130.0, 30.0, 161), rng.uniform(-110, 10, 32)])
    g['norm_in'] = s
    g['normalize'] = ref_audio._normalize(s)
    n = np.concatenate([np.linspace(-0.5, 1.5, 41), rng.uniform(0, 1, 32)])
    g['norm_inv_in'] = n
    g['normalize_inv'] = ref_audio._normalize_inv(n)
    n32 = n.astype(np.float32)
    g['normalize_inv_f32'] = ref_audio._normalize_inv(n32)
    # the amplitude the vocoder starts from (audio.py:53, numpy flavour): S = db_to_amp(normalize_inv(x) + ref) ** power
    g['spec_amp_pow'] = ref_audio._db_to_amp(ref_audio._normalize_inv(n) + ref_hparams.hparams.get('ref_level_db')) \
        ** ref_hparams.hparams.get('power')
    # save_wav: in-place scaling by 32767 / max(0.01, max|wav|), int16 cast (audio.py:15-17)
    sw_cases = [0.3 * np.sin(np.arange(2000) / 7.0) + 0.01 * rng.standard_normal(2000),
                1e-4 * rng.standard_normal(500),                      # below the 0.01 floor: not blown up to full scale
                np.array([0.0, 0.5, -1.0, 0.999, -0.25]),
                (0.2 * rng.standard_normal(1000)).astype(np.float32)]
    g['n_save_wav_cases'] = np.int64(len(sw_cases))
    for i, w in enumerate(sw_cases):
        g['sw_in_%d' % i] = w.copy()
        buf = w.copy()
        ref_audio.save_wav(buf, 'unused-%d.wav' % i)
        path, y, sr = written[-1]
        assert y.dtype == np.int16
        g['sw_pcm_%d' % i] = y
        g['sw_scaled_%d' % i] = buf                                  # the in-place scaled float array
        g['sw_rate_%d' % i] = np.int64(sr)
    np.savez_compressed(os.path.join(HERE, 'host_dsp_goldens.npz'), **g)
    print('wrote hparams_goldens.json (%d values) and host_dsp_goldens.npz (%d arrays)' % (len(vals), len(g)))


if __name__ == '__main__':
    main()
