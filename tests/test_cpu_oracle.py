"""The oracle against (a) its committed golden vectors and (b) independent formulations of the
same TF op semantics (torch CPU ops, direct DFTs). No GPU."""
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import taco_oracle as O
from tests.util import rel_l1

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_matches_golden_vectors():
    from tests.golden.make_oracle_goldens import case
    g = np.load(os.path.join(HERE, 'golden', 'oracle_small.npz'))
    hp, w, ids, lens = case()
    assert np.array_equal(ids, g['ids']) and np.array_equal(lens, g['lens'])
    r = O.tacotron_forward(ids, lens, w, hp, max_iters=3)
    assert rel_l1(r['enc_out'], g['enc_out']) < 1e-6
    assert rel_l1(r['mel'], g['mel']) < 1e-6
    assert rel_l1(r['linear'][:, :, ::16], g['linear']) < 1e-6
    assert rel_l1(r['alignments'], g['alignments']) < 1e-6
    wav = O.spectrogram_tensorflow_inv(g['gl_linear'].astype(np.float64), hp, iters=2)
    assert rel_l1(wav, g['gl_wav']) < 1e-5


def test_conv1d_same_vs_torch():
    rng = np.random.default_rng(0)
    for K in (1, 2, 3, 8, 16):
        x = rng.standard_normal((2, 11, 6))
        w = rng.standard_normal((K, 6, 5))
        b = rng.standard_normal(5)
        pl = (K - 1) // 2
        xt = torch.nn.functional.pad(torch.from_numpy(x).permute(0, 2, 1), (pl, K - 1 - pl))
        ref = torch.nn.functional.conv1d(xt, torch.from_numpy(w).permute(2, 1, 0), torch.from_numpy(b))
        assert np.allclose(O.conv1d_same(x, w, b), ref.permute(0, 2, 1).numpy(), atol=1e-12)


def test_maxpool_and_batchnorm_vs_torch():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 9, 4))
    xt = torch.nn.functional.pad(torch.from_numpy(x).permute(0, 2, 1), (0, 1), value=-np.inf)
    ref = torch.nn.functional.max_pool1d(xt, 2, 1).permute(0, 2, 1).numpy()
    assert np.array_equal(O.maxpool2_same(x), ref)
    # any window at stride 1 (hparams *_pooling_width): TF 'same' pads (w-1)//2 on the left, the rest on the right
    for width in (1, 2, 3, 4, 5, 9, 12):
        pl = (width - 1) // 2
        xt = torch.nn.functional.pad(torch.from_numpy(x).permute(0, 2, 1), (pl, width - 1 - pl), value=-np.inf)
        ref = torch.nn.functional.max_pool1d(xt, width, 1).permute(0, 2, 1).numpy()
        assert np.array_equal(O.maxpool_same(x, width), ref), width
    with pytest.raises(ValueError):
        O.maxpool_same(x, 2, stride=2)
    g, b, m, v = (rng.uniform(0.5, 1.5, 4) for _ in range(4))
    ref = torch.nn.functional.batch_norm(torch.from_numpy(x).reshape(-1, 4), torch.from_numpy(m), torch.from_numpy(v),
                                         torch.from_numpy(g), torch.from_numpy(b), False, 0.0, 1e-3)
    assert np.allclose(O.batch_norm_inference(x, g, b, m, v).reshape(-1, 4), ref.numpy(), atol=1e-12)


def test_gru_cell_is_tf_flavoured():
    """Reset gate first, applied before the candidate matmul; update gate multiplies the OLD state."""
    rng = np.random.default_rng(2)
    H, D = 4, 3
    w = {'s/gates/kernel': rng.standard_normal((D + H, 2 * H)), 's/gates/bias': np.ones(2 * H),
         's/candidate/kernel': rng.standard_normal((D + H, H)), 's/candidate/bias': np.zeros(H)}
    x, h = rng.standard_normal((1, D)), rng.standard_normal((1, H))
    got = O.gru_cell(x, h, w, 's/')
    g = 1 / (1 + np.exp(-(np.concatenate([x, h], 1) @ w['s/gates/kernel'] + 1.0)))
    r, u = g[:, :H], g[:, H:]
    c = np.tanh(np.concatenate([x, r * h], 1) @ w['s/candidate/kernel'])
    assert np.allclose(got, u * h + (1 - u) * c)
    # saturated update gate keeps the state
    w2 = dict(w)
    w2['s/gates/bias'] = np.concatenate([np.zeros(H), 50 * np.ones(H)])
    assert np.allclose(O.gru_cell(x, h, w2, 's/'), h)


def test_bidirectional_rnn_masking():
    """A.6: outputs past the length are 0, a padded row equals the same row run unpadded."""
    from tacotron_amd.hparams import HParams
    from tacotron_amd.weights import init_weights
    hp = HParams()
    w = O.cast_weights(init_weights(hp, seed=5), np.float64)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 9, 128))
    out = O.bidirectional_gru(x, np.array([9, 4]), w, 'encoder_cbhg', 128)
    assert np.all(out[1, 4:] == 0)
    solo = O.bidirectional_gru(x[1:2, :4], np.array([4]), w, 'encoder_cbhg', 128)
    assert np.allclose(out[1, :4], solo[0])
    full = O.bidirectional_gru(x, None, w, 'encoder_cbhg', 128)
    assert np.allclose(full[0], out[0])


def test_decode_properties():
    from tacotron_amd.hparams import HParams
    from tacotron_amd.weights import init_weights
    hp = HParams()
    w = O.cast_weights(init_weights(hp, seed=6, randomize_bias=True), np.float64)
    enc = np.random.default_rng(4).standard_normal((2, 6, 256))
    mel, al, steps = O.decode(enc, w, hp, max_iters=4)
    assert steps == 4 and mel.shape == (2, 20, 80) and al.shape == (2, 6, 4)
    assert np.allclose(al.sum(1), 1.0)
    # utterances are independent
    mel0, al0, _ = O.decode(enc[:1], w, hp, max_iters=4)
    assert np.allclose(mel0, mel[:1]) and np.allclose(al0, al[:1])
    # stop test (helpers.py:75): all-zero outputs end the loop after one step
    w['decoder/output_projection_wrapper/kernel'] = np.zeros((256, 400))
    w['decoder/output_projection_wrapper/bias'] = np.zeros(400)
    assert O.decode(enc, w, hp, max_iters=4)[2] == 1


def test_stft_istft_tf_semantics():
    """Uncentred frames, periodic Hann, end-zero-padded 2048-point rfft; istft without window-sum division."""
    rng = np.random.default_rng(5)
    y = rng.standard_normal(300 * 3 + 1200)
    X = O.stft_tf(y, 2048, 300, 1200)
    assert X.shape == (4, 1025)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(1200) / 1200)
    n = np.arange(1200)
    for m, k in [(0, 0), (1, 7), (3, 1024), (2, 513)]:
        direct = np.sum(y[300 * m:300 * m + 1200] * win * np.exp(-2j * np.pi * k * n / 2048))
        assert abs(X[m, k] - direct) < 1e-9
    z = O.istft_tf(X, 2048, 300, 1200)
    assert z.shape == y.shape
    # steady-state samples are scaled by sum of squared windows at 75 % overlap = 1.5
    assert np.allclose(z[900:1200], 1.5 * y[900:1200], atol=1e-9)


def test_griffin_lim_fixed_point_and_librosa_variant():
    from tacotron_amd.hparams import HParams
    hp = HParams()
    rng = np.random.default_rng(6)
    lin = rng.uniform(0.4, 1.0, size=(8, 1025))
    y0 = O.spectrogram_tensorflow_inv(lin, hp, iters=0)
    y3 = O.spectrogram_tensorflow_inv(lin, hp, iters=3)
    assert y0.shape == y3.shape == (7 * 300 + 1200,) and np.isfinite(y3).all()
    y = O.spectrogram_inv_librosa(lin.T, hp, np.random.default_rng(0))
    assert y.shape == (7 * 300,) and np.isfinite(y).all()


def test_fft_core_host_emulation():
    """The HIP Griffin-Lim kernel's FFT butterflies / split step, emulated thread by thread on the host."""
    exe = os.path.join(HERE, 'host', 'fft_host_test')
    subprocess.run(['g++', '-O2', '-o', exe, os.path.join(HERE, 'host', 'fft_host_test.cpp')], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_mel_basis_restatements_agree():
    """tools.audio.build_mel_basis (product, vectorised) against the oracle's filter-by-filter formulation of
    librosa.filters.mel (Slaney scale, area normalisation). Both come from library knowledge: unpinned."""
    from tacotron_amd.hparams import HParams
    from tacotron_amd.tools.audio import build_mel_basis
    hp = HParams()
    a = build_mel_basis(hp)
    b = O.mel_basis_librosa(hp.sample_rate, hp.n_fft, hp.num_mels)
    assert a.shape == (80, 1025) and a.dtype == np.float32
    assert np.abs(a - b).max() < 1e-8
    assert (a >= 0).all() and (a.sum(1) > 0).all() and a[:, 0].sum() == 0
    # Slaney normalisation: every triangle has (approximately, on the FFT grid) unit area in Hz
    hz_per_bin = hp.sample_rate / hp.n_fft
    assert np.allclose(a.sum(1) * hz_per_bin, 1.0, atol=0.15)


def test_pre_emphasis_roundtrip():
    x = np.random.default_rng(0).standard_normal(1000)
    assert np.allclose(O.pre_emphasis_inv(O.pre_emphasis(x, 0.97), 0.97), x, atol=1e-9)
