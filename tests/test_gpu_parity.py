"""GPU parity tests: every stage of the HIP path, called through the C ABI (TacoEngine is a
ctypes shim), against the fp64 numpy oracle on the same seeded inputs.

Tolerances: north_star asks for mel / linear within 1e-3 relative L1 of the reference and an
identical per-frame attention argmax. Stage tests feed the ORACLE's output of the previous
stage so each stage is judged alone, at 1e-4; the end-to-end test uses the stated 1e-3.
With random-init weights the attention is almost flat, so the argmax is compared wherever the
oracle's top-1/top-2 margin exceeds fp32 resolution (1e-6 absolute on probabilities ~1/T_in)
and on a peaked-attention weight set everywhere.
"""
import numpy as np
import pytest

from oracle import taco_oracle as O
from tests.util import rel_l1, max_abs, synth_ids

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize('M,T,Cin,taps,N,act,pool,extras', [
    (95, 95, 256, 1, 256, 1, 0, False),
    (150, 50, 128, 16, 128, 1, 0, True),
    (64, 64, 80, 8, 128, 1, 0, True),
    (300, 100, 2048, 3, 128, 1, 1, True),
    (77, 11, 256, 1, 1025, 0, 0, False),
    (1000, 250, 1024, 3, 256, 1, 1, True),
    (4096, 512, 128, 5, 384, 0, 0, True),
])
def test_conv_gemm_kernel(engine, M, T, Cin, taps, N, act, pool, extras):
    """conv_gemm vs fp64 numpy: dense, even/odd-width SAME convs, pooled loader, BN/residual epilogue,
    N not a multiple of the tile (1025), both tile configurations."""
    import ctypes
    import torch
    rng = np.random.default_rng(M + N + taps)
    B = M // T
    x = rng.standard_normal((B, T, Cin)).astype(np.float32)
    w = (rng.standard_normal((taps, Cin, N)) / np.sqrt(taps * Cin)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, N).astype(np.float32) if extras else None
    shift = rng.standard_normal(N).astype(np.float32) if extras else None
    res = rng.standard_normal((B, T, N)).astype(np.float32) if extras else None
    xin = O.maxpool2_same(x.astype(np.float64)) if pool else x.astype(np.float64)
    ref = O.conv1d_same(xin, w.astype(np.float64), bias.astype(np.float64))
    if act == 1:
        ref = np.maximum(ref, 0)
    if extras:
        ref = ref * scale + shift + res
    ldw = (N + 127) // 128 * 128
    wp = np.zeros((taps * Cin, ldw), np.float32)
    wp[:, :N] = w.reshape(taps * Cin, N)
    dev = engine.device
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if a is not None else None
    X, W, Bi, Sc, Sh, R = d(x), d(wp), d(bias), d(scale), d(shift), d(res)
    C = torch.zeros(M, N, dtype=torch.float32, device=dev)
    p = lambda t: t.data_ptr() if t is not None else None
    rc = engine.lib.taco_debug_conv_gemm(p(X), Cin, M, T, Cin, taps, (taps - 1) // 2, p(W), ldw, N, p(Bi), p(Sc),
                                         p(Sh), p(R), N, p(C), N, act, pool, None)
    assert rc == 0, engine.lib.taco_last_error()
    torch.cuda.synchronize()
    got = _np(C).reshape(B, T, N)
    assert rel_l1(got, ref) < 2e-6, rel_l1(got, ref)
    assert max_abs(got, ref) < 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('N,pool', [(256, 1), (128, 0), (1025, 0)])
def test_conv_gemm_big_tiles(engine, monkeypatch, N, pool):
    """The large-M tile shapes (128 x 128, 128 x 64; the 64 x 64 kernel takes the shapes of test_conv_gemm_kernel) against
    fp64 numpy, and bit for bit against each other: 64 x 64 tiles forced through a small sub-problem of the same rows,
    and the eight-wave layout of the 128 x 128 tile (TACO_GEMM_W8). An element's sum has the same k order in all."""
    import torch
    M, T, Cin, taps = 32768, 512, 64, 3          # 256 row blocks of 128: the big-tile rule holds for N = 128 too
    rng = np.random.default_rng(N + pool)
    B = M // T
    x = rng.standard_normal((B, T, Cin)).astype(np.float32)
    w = (rng.standard_normal((taps, Cin, N)) / np.sqrt(taps * Cin)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    xin = O.maxpool2_same(x.astype(np.float64)) if pool else x.astype(np.float64)
    ref = np.maximum(O.conv1d_same(xin, w.astype(np.float64), bias.astype(np.float64)), 0)
    ldw = (N + 127) // 128 * 128
    wp = np.zeros((taps * Cin, ldw), np.float32)
    wp[:, :N] = w.reshape(taps * Cin, N)
    dev = engine.device
    X, W, Bi = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (x, wp, bias))

    def run(rows):
        C = torch.zeros(rows, N, dtype=torch.float32, device=dev)
        rc = engine.lib.taco_debug_conv_gemm(X.data_ptr(), Cin, rows, T, Cin, taps, (taps - 1) // 2, W.data_ptr(), ldw, N,
                                             Bi.data_ptr(), None, None, None, N, C.data_ptr(), N, 1, pool, None)
        assert rc == 0, engine.lib.taco_last_error()
        torch.cuda.synchronize()
        return _np(C)

    monkeypatch.delenv('TACO_GEMM_W8', raising=False)
    big = run(M)
    assert rel_l1(big.reshape(B, T, N), ref) < 2e-6
    small = run(2 * T)                                   # two utterances: 64 x 64 tiles
    assert np.array_equal(small, big[:2 * T])
    monkeypatch.setenv('TACO_GEMM_W8', '1')
    assert np.array_equal(run(M), big)


@pytest.mark.parametrize('B,T_in', [(1, 5), (2, 7), (3, 37), (5, 150)])
def test_encode(engine, hp, weights, B, T_in):
    rng = np.random.default_rng(10 * B + T_in)
    ids, lens = synth_ids(rng, B, T_in)
    ref = O.encode(ids, lens, O.cast_weights(weights, np.float64), hp)
    got = _np(engine.encode(ids, lens))
    assert got.shape == ref.shape
    # padded steps of the BiGRU are exactly zero (A.6)
    for b in range(B):
        assert np.all(got[b, lens[b]:] == 0)
    assert rel_l1(got, ref) < 1e-4, rel_l1(got, ref)


def _check_decode(engine, hp, w, enc, S, tag):
    w64 = O.cast_weights(w, np.float64)
    mel_r, al_r, steps_r = O.decode(enc, w64, hp, max_iters=S)
    mel, al, steps = engine.decode(enc.astype(np.float32), max_iters=S)
    mel, al, steps = _np(mel), _np(al), int(_np(steps)[0])
    engine.decode_status()
    assert steps == steps_r == S
    assert mel.shape == mel_r.shape and al.shape == al_r.shape
    e_mel, e_al = rel_l1(mel, mel_r), rel_l1(al, al_r)
    # argmax over T_in per (utterance, step)
    top = np.sort(al_r, axis=1)
    margin = top[:, -1, :] - top[:, -2, :]
    same = al.argmax(1) == al_r.argmax(1)
    decided = margin > 1e-6
    print('%s: mel relL1 %.2e align relL1 %.2e argmax same %d/%d (decided %d, min margin %.2e)' % (
        tag, e_mel, e_al, same.sum(), same.size, decided.sum(), margin.min()))
    assert e_mel < 1e-4 and e_al < 1e-4
    assert np.all(same[decided])
    assert np.allclose(al.sum(1), 1.0, atol=1e-5)
    return same, decided


@pytest.mark.parametrize('B,T_in,S', [(1, 6, 3), (2, 9, 6), (5, 37, 12), (17, 150, 8), (32, 150, 4)])
def test_decode(engine, hp, weights, B, T_in, S):
    rng = np.random.default_rng(B * 1000 + T_in)
    enc = (0.5 * rng.standard_normal((B, T_in, 256))).astype(np.float32).astype(np.float64)
    _check_decode(engine, hp, weights, enc, S, 'decode B=%d T_in=%d S=%d' % (B, T_in, S))


def test_decode_peaked_attention(hp):
    """A weight set whose attention_v is scaled so the alignment is sharply peaked: the argmax
    must then match the oracle on every frame."""
    from tacotron_amd.engine import TacoEngine
    from tacotron_amd.weights import init_weights
    w = init_weights(hp, seed=7, randomize_bn=True, randomize_bias=True, attention_sharpness=40.0)
    eng = TacoEngine(w, hp)
    try:
        rng = np.random.default_rng(5)
        enc = rng.standard_normal((4, 50, 256)).astype(np.float32).astype(np.float64)
        same, decided = _check_decode(eng, hp, w, enc, 20, 'peaked')
        assert decided.mean() > 0.9
        assert np.all(same)
    finally:
        eng.close()


def test_decode_modes_agree(engine, hp):
    """Mode 1 (captured hipGraph replay) and mode 0 (eager) launch the same per-stage kernels: bit-equal.
    Mode 2 (persistent kernel, the default) orders its reductions differently: equal to fp32 rounding,
    and bit-reproducible run to run."""
    import torch
    rng = np.random.default_rng(3)
    enc = rng.standard_normal((19, 20, 256)).astype(np.float32)
    try:
        engine.set_decode_graph(1)
        m1, a1, _ = engine.decode(enc, max_iters=7)
        m1b, a1b, _ = engine.decode(enc, max_iters=7)
        engine.set_decode_graph(0)
        m2, a2, _ = engine.decode(enc, max_iters=7)
        engine.set_decode_graph(2)
        m3, a3, _ = engine.decode(enc, max_iters=7)
        m3b, a3b, _ = engine.decode(enc, max_iters=7)
        engine.set_decode_graph(3)                       # one workgroup per 16-utterance tile (dec_tile.hip)
        m4, a4, _ = engine.decode(enc, max_iters=7)
        m4b, a4b, _ = engine.decode(enc, max_iters=7)
        torch.cuda.synchronize()
        engine.decode_status()
    finally:
        engine.set_decode_graph(2)
    assert torch.equal(m1, m2) and torch.equal(a1, a2) and torch.equal(m1, m1b)
    assert torch.equal(m3, m3b) and torch.equal(a3, a3b)
    assert rel_l1(_np(m3), _np(m1)) < 1e-5 and rel_l1(_np(a3), _np(a1)) < 1e-5
    assert torch.equal(m4, m4b) and torch.equal(a4, a4b)
    assert rel_l1(_np(m4), _np(m1)) < 1e-5 and rel_l1(_np(a4), _np(a1)) < 1e-5


@pytest.mark.parametrize('mode', [1, 2, 3])
def test_decode_parity_per_mode(engine, hp, weights, mode):
    """Every decode implementation against the oracle (the other decode tests run the default, mode 2)."""
    rng = np.random.default_rng(77)
    enc = (0.5 * rng.standard_normal((18, 61, 256))).astype(np.float32).astype(np.float64)
    try:
        engine.set_decode_graph(mode)
        _check_decode(engine, hp, weights, enc, 9, 'mode %d' % mode)
    finally:
        engine.set_decode_graph(2)


def test_decode_batch_invariance(engine, hp):
    """an utterance's result must not depend on what else is in the batch / shard."""
    import torch
    rng = np.random.default_rng(4)
    enc = rng.standard_normal((19, 33, 256)).astype(np.float32)
    for mode in (2, 1, 3):
        try:
            engine.set_decode_graph(mode)
            m_all, a_all, _ = engine.decode(enc, max_iters=5)
            for lo, hi in [(0, 1), (3, 10), (16, 19)]:
                m, a, _ = engine.decode(enc[lo:hi], max_iters=5)
                assert torch.equal(m, m_all[lo:hi]) and torch.equal(a, a_all[lo:hi]), mode
        finally:
            engine.set_decode_graph(2)


def test_decode_dropout_replay(engine, hp, weights):
    """Optional training-mode prenet dropout replay (A.3): keep flags in, oracle applies the same."""
    rng = np.random.default_rng(8)
    B, T_in, S = 3, 11, 4
    enc = rng.standard_normal((B, T_in, 256))
    keep = (rng.random((S, B, 384)) >= 0.5).astype(np.float32)
    w = O.cast_weights(weights, np.float64)
    # oracle with masks: restate the loop with the mask applied after each prenet ReLU
    from oracle.taco_oracle import _ATTW, _CELL0, _DEC, gru_cell, dense
    keys = enc @ w['memory_layer/kernel']
    h_att = np.zeros((B, 256)); ctx = np.zeros((B, 256)); h1 = np.zeros((B, 256)); h2 = np.zeros((B, 256))
    prev = np.zeros((B, 80)); outs = []
    for s in range(S):
        x = np.concatenate([prev, ctx], -1)
        sc = _ATTW + 'decoder_prenet_wrapper/decoder_prenet/prenet_dense_%d/'
        x = dense(x, w[sc % 1 + 'kernel'], w[sc % 1 + 'bias'], 'relu') * keep[s, :, :256] * 2.0
        x = dense(x, w[sc % 2 + 'kernel'], w[sc % 2 + 'bias'], 'relu') * keep[s, :, 256:] * 2.0
        h_att = gru_cell(x, h_att, w, _ATTW + 'decoder_prenet_wrapper/gru_cell/')
        q = h_att @ w[_ATTW + 'bahdanau_attention/query_layer/kernel']
        e = np.sum(w[_ATTW + 'bahdanau_attention/attention_v'] * np.tanh(keys + q[:, None]), -1)
        a = np.exp(e - e.max(-1, keepdims=True)); a /= a.sum(-1, keepdims=True)
        ctx = np.einsum('bt,btd->bd', a, enc)
        d0 = np.concatenate([h_att, ctx], -1) @ w[_CELL0 + 'kernel'] + w[_CELL0 + 'bias']
        h1 = gru_cell(d0, h1, w, _DEC + 'multi_rnn_cell/cell_1/gru_cell/'); o1 = h1 + d0
        h2 = gru_cell(o1, h2, w, _DEC + 'multi_rnn_cell/cell_2/gru_cell/'); o2 = h2 + o1
        out = o2 @ w[_DEC + 'kernel'] + w[_DEC + 'bias']
        outs.append(out); prev = out[:, -80:]
    ref = np.stack(outs, 1).reshape(B, S * 5, 80)
    mel, _, _ = engine.decode(enc.astype(np.float32), max_iters=S, dropout_keep=keep)
    assert rel_l1(_np(mel), ref) < 1e-4


def test_decode_stop_test(hp):
    """helpers.py:75: with all-zero output projection every step emits exactly 0.0, so
    dynamic_decode stops after the first step; outputs past it are zero."""
    from tacotron_amd.engine import TacoEngine
    from tacotron_amd.weights import init_weights
    w = init_weights(hp, seed=3)
    w['decoder/output_projection_wrapper/kernel'][:] = 0
    w['decoder/output_projection_wrapper/bias'][:] = 0
    eng = TacoEngine(w, hp)
    try:
        enc = np.random.default_rng(0).standard_normal((2, 9, 256)).astype(np.float32)
        mel, al, steps = eng.decode(enc, max_iters=6)
        assert int(_np(steps)[0]) == 1
        _, al_r, steps_r = O.decode(enc.astype(np.float64), O.cast_weights(w, np.float64), hp, max_iters=6)
        assert steps_r == 1
        assert np.all(_np(mel) == 0)
        assert rel_l1(_np(al)[:, :, :1], al_r) < 1e-4 and np.all(_np(al)[:, :, 1:] == 0)
    finally:
        eng.close()


@pytest.mark.parametrize('B,T', [(1, 5), (2, 15), (3, 100), (2, 1000)])
def test_postnet(engine, hp, weights, B, T):
    rng = np.random.default_rng(B + T)
    mel = (0.3 * rng.standard_normal((B, T, 80))).astype(np.float32)
    ref = O.postnet(mel.astype(np.float64), O.cast_weights(weights, np.float64), hp)
    got = _np(engine.postnet(mel))
    assert got.shape == ref.shape
    assert rel_l1(got, ref) < 1e-4, rel_l1(got, ref)


@pytest.mark.parametrize('B,T,iters', [(1, 4, 0), (1, 7, 1), (2, 12, 3), (1, 40, 60)])
def test_griffin_lim(engine, hp, B, T, iters):
    """Griffin-Lim vs the fp64 oracle. The phase projection E/|E| is ill-conditioned where |E| is
    tiny, so fp32 round-off grows with the iteration count; the fp32 oracle's own distance from the
    fp64 one is printed as the noise floor, and the bound is on relative L2."""
    rng = np.random.default_rng(T + iters)
    lin = rng.uniform(0.3, 1.0, size=(B, T, 1025)).astype(np.float32)
    ref = O.spectrogram_tensorflow_inv(lin.astype(np.float64), hp, iters=iters)
    ref32 = O.spectrogram_tensorflow_inv(lin, hp, iters=iters, dtype=np.float32)
    got = _np(engine.griffin_lim(lin, iters=iters))
    assert got.shape == ref.shape == (B, (T - 1) * 300 + 1200)
    l2 = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print('GL T=%d iters=%d: gpu-vs-f64 relL2 %.2e ; f32-numpy-vs-f64 %.2e' % (T, iters, l2(got, ref), l2(ref32, ref)))
    assert l2(got, ref) < (1e-5 if iters <= 1 else 1e-3)


@pytest.mark.parametrize('B,T,iters,zero_phase', [(1, 6, 0, False), (2, 9, 1, False), (2, 37, 3, False), (1, 20, 2, True),
                                                  (1, 50, 60, False)])
def test_griffin_lim_librosa(engine, hp, B, T, iters, zero_phase):
    """The numpy/librosa Griffin-Lim (audio.py:44-55,167-181; taco_griffin_lim_librosa) vs the fp64 oracle on the
    same uniform phase draws, incl. runs that straddle both reflected ends (T = 6: every frame reads reflected
    samples) and a ragged last run (T = 37). Same conditioning caveat as test_griffin_lim."""
    rng = np.random.default_rng(100 + T + iters)
    lin = rng.uniform(0.3, 1.0, size=(B, T, 1025)).astype(np.float32)
    u = np.zeros((B, T, 1025)) if zero_phase else rng.random((B, T, 1025))
    ref = np.stack([O.griffin_lim_librosa(lin[b].T.astype(np.float64), hp, u[b].T, iters) for b in range(B)])
    got = _np(engine.griffin_lim_librosa(lin, None if zero_phase else u.astype(np.float32), iters=iters))
    assert got.shape == ref.shape == (B, (T - 1) * 300)
    l2 = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    print('GL librosa T=%d iters=%d: gpu-vs-f64 relL2 %.2e' % (T, iters, l2))
    assert l2 < (2e-5 if iters <= 1 else 1e-3)


def test_spectrogram_inv_tool(engine, hp):
    """tools.audio.spectrogram_inv (audio.py:44-55): global numpy seed -> same draws as the oracle, de-emphasis on
    the host; and too-short inputs are refused (reflect padding needs hop (T-1) > n_fft/2)."""
    from tacotron_amd.tools import audio
    rng = np.random.default_rng(5)
    spect = rng.uniform(0.3, 1.0, size=(1025, 12))
    np.random.seed(1234)
    got = audio.spectrogram_inv(spect, engine, iters=2)
    np.random.seed(1234)
    u = np.random.rand(1025, 12)
    ref = O.pre_emphasis_inv(O.griffin_lim_librosa(spect, hp, u, 2), hp.preemphasis)
    assert got.shape == ref.shape == (11 * 300,)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-3
    with pytest.raises(RuntimeError):
        engine.griffin_lim_librosa(spect.T[:4].astype(np.float32), None, iters=1)


def test_synthesize_end_to_end(engine, hp, weights):
    """ids -> mel, linear, alignments through taco_synthesize vs the oracle end to end (1e-3, north_star)."""
    rng = np.random.default_rng(42)
    B, T_in, S = 3, 23, 10
    ids, lens = synth_ids(rng, B, T_in)
    ref = O.tacotron_forward(ids, lens, weights, hp, max_iters=S)
    o = engine.finish(engine.synthesize(ids, lens, max_iters=S, to_wav=True, gl_iters=2), to_wav=True, gl_iters=2)
    assert o['steps_run'] == S
    assert rel_l1(_np(o['enc_out']), ref['enc_out']) < 1e-3
    assert rel_l1(_np(o['mel']), ref['mel']) < 1e-3
    assert rel_l1(_np(o['linear']), ref['linear']) < 1e-3
    assert rel_l1(_np(o['alignments']), ref['alignments']) < 1e-3
    wav_ref = O.spectrogram_tensorflow_inv(ref['linear'], hp, iters=2)
    assert _np(o['wav']).shape == wav_ref.shape
    assert np.linalg.norm(_np(o['wav']) - wav_ref) / np.linalg.norm(wav_ref) < 1e-3


def test_persistent_decode_under_uneven_load(engine, hp, weights):
    """The persistent decoder's inter-workgroup hand-offs must not depend on an idle chip: the same decode, run
    while two other engines keep the other CUs busy with GEMM-heavy post-nets and a second persistent decode on their
    own streams, is bit-identical to the run alone (and its bounded waits do not time out)."""
    import torch
    from tacotron_amd.engine import TacoEngine
    rng = np.random.default_rng(11)
    enc = torch.from_numpy(rng.standard_normal((24, 150, 256)).astype(np.float32) * 0.5).cuda()
    mel_load = torch.from_numpy(rng.standard_normal((16, 400, hp.num_mels)).astype(np.float32)).cuda()
    engine.set_decode_graph(2)
    m0, a0, _ = engine.decode(enc, max_iters=60)
    torch.cuda.synchronize()
    engine.decode_status()
    m0, a0 = m0.clone(), a0.clone()
    others = [TacoEngine(weights, hp, device=0) for _ in range(2)]
    try:
        for e in others + [engine]:
            e.detached = True
        torch.cuda.synchronize()
        for rep in range(3):
            for _ in range(4):
                others[0].postnet(mel_load)
            others[1].decode(enc[:7], max_iters=40)
            m1, a1, _ = engine.decode(enc, max_iters=60)
            for _ in range(4):
                others[0].postnet(mel_load)
            torch.cuda.synchronize()
            engine.decode_status()
            others[1].decode_status()
            assert torch.equal(m1, m0) and torch.equal(a1, a0), 'rep %d' % rep
    finally:
        engine.detached = False
        torch.cuda.synchronize()
        for e in others:
            e.close()


@pytest.mark.parametrize('B,T_in,T_out', [(3, 11, 20), (17, 40, 33)])
def test_decode_teacher_forced(engine, hp, weights, B, T_in, T_out):
    """Teacher-forced decode (TrainingHelper, model/helpers.py:6-43) against the oracle; T_out = 33 is not a multiple
    of r = 5: targets[:, r-1::r] then has 6 frames, the 7th group is dropped as the reference's slice drops it."""
    rng = np.random.default_rng(5 * B + T_out)
    enc = rng.standard_normal((B, T_in, 256)) * 0.5
    targets = rng.standard_normal((B, T_out, hp.num_mels)) * 0.7
    w64 = O.cast_weights(weights, np.float64)
    mel_r, al_r, steps_r = O.decode(enc, w64, hp, targets=targets)
    mel, al, steps = engine.decode(enc.astype(np.float32), targets=targets.astype(np.float32))
    mel, al = _np(mel), _np(al)
    assert int(_np(steps)[0]) == steps_r == (T_out - 4 + 4) // 5
    assert mel.shape == mel_r.shape and al.shape == al_r.shape
    assert rel_l1(mel, mel_r) < 1e-4 and rel_l1(al, al_r) < 1e-4
    # and it differs from the free-running decode of the same memory (the targets are really fed)
    mel_f, _, _ = engine.decode(enc.astype(np.float32), max_iters=steps_r)
    assert rel_l1(_np(mel_f), mel_r) > 1e-3


@pytest.mark.parametrize('n', [3001, 24000])
def test_spectrogram_analysis(engine, hp, n):
    """Analysis side (tools/audio.py:23-42,66-79): centred STFT of the pre-emphasised signal, dB, normalisation, mel
    projection, against the oracle's float64 restatement of librosa's stft / filters.mel. Values are in [0, 1]
    (1 unit = 100 dB): absolute tolerance; bins the clip at 0 silences compare as equal."""
    rng = np.random.default_rng(n)
    t = np.arange(n) / hp.sample_rate
    y = 0.3 * np.sin(2 * np.pi * 440.0 * t) * np.hanning(n) + 0.02 * rng.standard_normal(n)
    lin_r = O.spectrogram_librosa(y, hp).T
    mel_r = O.mel_spectrogram_librosa(y, hp).T
    lin, mel = engine.spectrogram(y.astype(np.float32))
    lin, mel = _np(lin)[0], _np(mel)[0]
    assert lin.shape == lin_r.shape == (1 + n // 300, 1025) and mel.shape == mel_r.shape == (1 + n // 300, 80)
    assert lin_r.max() > 0.5 and mel_r.max() > 0.5            # a real signal, not all clipped
    assert max_abs(lin, lin_r) < 2e-3 and max_abs(mel, mel_r) < 2e-3
    assert rel_l1(lin, lin_r) < 1e-4 and rel_l1(mel, mel_r) < 1e-4
    # batch of two equal-length signals, one output skipped
    y2 = np.stack([y, y[::-1]]).astype(np.float32)
    lin2, none = engine.spectrogram(y2, want_mel=False)
    assert none is None and np.array_equal(_np(lin2)[0], lin)
    # tools.audio mirrors the reference's [freq, T] orientation
    from tacotron_amd.tools import audio
    assert audio.spectrogram(y, engine).shape == (1025, 1 + n // 300)
    assert audio.mel_spectrogram(y, engine).shape == (80, 1 + n // 300)


def test_griffin_lim_batch_and_run_invariance(engine, hp):
    """An utterance's waveform is bit-identical alone (short runs of frames per workgroup) and inside a large batch
    (16-frame runs): the overlap-add order per sample does not depend on the launch geometry."""
    rng = np.random.default_rng(9)
    lin = rng.uniform(0.3, 1.0, size=(40, 200, 1025)).astype(np.float32)
    full = _np(engine.griffin_lim(lin, iters=3))
    for b in (0, 17, 39):
        assert np.array_equal(_np(engine.griffin_lim(lin[b:b + 1], iters=3))[0], full[b])
    u = rng.random((40, 200, 1025)).astype(np.float32)
    full = _np(engine.griffin_lim_librosa(lin, u, iters=2))
    assert np.array_equal(_np(engine.griffin_lim_librosa(lin[5:6], u[5:6], iters=2))[0], full[5])


def test_griffin_lim_group_split_is_bit_identical(engine, hp):
    """taco_set_gl_split: the utterances of a batch as 2..4 chains on streams of the handle's own (a scheduling choice
    for a batch that has the GPU to itself): the same bits as the single chain, for uneven group sizes too."""
    import torch
    rng = np.random.default_rng(41)
    lin = rng.uniform(0.2, 1.0, size=(27, 70, hp.num_freq)).astype(np.float32)
    ref = engine.griffin_lim(lin, iters=4)
    try:
        for groups in (2, 3, 4):
            engine.set_gl_split(groups)
            got = engine.griffin_lim(lin, iters=4)
            torch.cuda.synchronize()
            assert torch.equal(got, ref), groups
    finally:
        engine.set_gl_split(1)
    with pytest.raises(Exception):
        engine.set_gl_split(5)


def test_hip_path_matches_committed_goldens():
    """The HIP path against tests/golden/oracle_small.npz read from disk (no oracle call at run time): the same vectors
    the CPU suite pins the oracle with (tests/test_cpu_oracle.py::test_oracle_matches_golden_vectors)."""
    import os
    from tests.golden.make_oracle_goldens import case
    from tacotron_amd.engine import TacoEngine
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'oracle_small.npz'))
    hp, w, ids, lens = case()
    eng = TacoEngine(w, hp)
    try:
        o = eng.finish(eng.synthesize(g['ids'], g['lens'], max_iters=3))
        assert rel_l1(_np(o['enc_out']), g['enc_out']) < 1e-3
        assert rel_l1(_np(o['mel']), g['mel']) < 1e-3
        assert rel_l1(_np(o['linear'])[:, :, ::16], g['linear']) < 1e-3
        assert rel_l1(_np(o['alignments']), g['alignments']) < 1e-3
        wav = _np(eng.griffin_lim(g['gl_linear'], iters=2))
        assert np.linalg.norm(wav - g['gl_wav']) / np.linalg.norm(g['gl_wav']) < 1e-4
    finally:
        eng.close()


def test_decode_long_input_split_invariance(engine, hp, weights):
    """Long inputs (T_in = 400: a full tile's key parts no longer fit in LDS, so part of the rows is scored from L2)
    and the attention split that follows the batch size (32 workgroups per utterance alone, 8 in a batch of 3, 2 in a
    full tile): parity with the oracle, and the same utterance bit-identical in every batch."""
    import torch
    rng = np.random.default_rng(21)
    enc = (0.5 * rng.standard_normal((18, 400, 256))).astype(np.float32)
    _check_decode(engine, hp, weights, enc[:2].astype(np.float64), 3, 'decode B=2 T_in=400')
    m_all, a_all, _ = engine.decode(enc, max_iters=4)
    for lo, hi in [(0, 1), (0, 3), (5, 10), (16, 18)]:
        m, a, _ = engine.decode(enc[lo:hi], max_iters=4)
        assert torch.equal(m, m_all[lo:hi]) and torch.equal(a, a_all[lo:hi]), (lo, hi)
    ref = O.decode(enc[16:18].astype(np.float64), O.cast_weights(weights, np.float64), hp, max_iters=4)
    assert rel_l1(_np(m_all[16:18]), ref[0]) < 1e-3


@pytest.mark.parametrize('B,T_in,S', [(1, 1, 1), (2, 2, 2), (3, 3, 1)])
def test_minimum_sizes(engine, hp, weights, B, T_in, S):
    """One-character inputs, one decoder step: every stage at its smallest size (conv banks wider than the sequence,
    attention parts without rows, a single Griffin-Lim frame run)."""
    rng = np.random.default_rng(B + 10 * T_in)
    ids, lens = synth_ids(rng, B, T_in, min_len=1)
    ref = O.tacotron_forward(ids, lens, weights, hp, max_iters=S)
    o = engine.finish(engine.synthesize(ids, lens, max_iters=S, to_wav=True, gl_iters=2), to_wav=True, gl_iters=2)
    assert o['steps_run'] == S
    for name in ('enc_out', 'mel', 'linear', 'alignments'):
        assert rel_l1(_np(o[name]), ref[name]) < 1e-3, name
    wav_ref = O.spectrogram_tensorflow_inv(ref['linear'], hp, iters=2)
    assert np.linalg.norm(_np(o['wav']) - wav_ref) / np.linalg.norm(wav_ref) < 1e-3


@pytest.mark.parametrize('B', [128, 130])
def test_decode_large_batch(engine, hp, weights, B):
    """B = 128 is the persistent decoder's largest launch (8 batch tiles, all 256 CUs); beyond it the decode takes
    the per-stage kernels. Both against the oracle on a few utterances, and consistent across the boundary."""
    rng = np.random.default_rng(B)
    enc = (0.5 * rng.standard_normal((B, 12, 256))).astype(np.float32)
    mel, al, steps = engine.decode(enc, max_iters=3)
    engine.decode_status()
    assert int(_np(steps)[0]) == 3
    pick = [0, 77, B - 1]
    ref = O.decode(enc[pick].astype(np.float64), O.cast_weights(weights, np.float64), hp, max_iters=3)
    assert rel_l1(_np(mel)[pick], ref[0]) < 1e-3 and rel_l1(_np(al)[pick], ref[1]) < 1e-3


def test_very_long_input(engine, hp, weights):
    """T_in = 1000 characters (far beyond the reference's own utterances): encoder, decoder with most key rows and
    all values read from L2, alignments [B, 1000, S]."""
    rng = np.random.default_rng(31)
    ids, lens = synth_ids(rng, 2, 1000)
    ref = O.tacotron_forward(ids, lens, weights, hp, max_iters=2)
    o = engine.finish(engine.synthesize(ids, lens, max_iters=2))
    for name in ('enc_out', 'mel', 'linear', 'alignments'):
        assert rel_l1(_np(o[name]), ref[name]) < 1e-3, name


@pytest.mark.parametrize('mode', [1, 2])
@pytest.mark.parametrize('B,T_in', [(1, 5), (3, 37), (17, 40), (33, 21)])
def test_encode_bigru_mfma_mode(engine, hp, weights, B, T_in, mode):
    """The throughput variants of the BiGRU scan (taco_set_bigru_mode 1 / 2: sixteen / four utterances per workgroup on
    fp32 MFMA) against the oracle through the encoder, ragged lengths, batches that fill one, two and three column groups partly;
    padded steps exactly zero; an utterance's result does not depend on what it is batched with."""
    rng = np.random.default_rng(100 * B + T_in)
    ids, lens = synth_ids(rng, B, T_in)
    ref = O.encode(ids, lens, O.cast_weights(weights, np.float64), hp)
    engine.set_bigru_mode(mode)
    try:
        got = _np(engine.encode(ids, lens))
        alone = _np(engine.encode(ids[B - 1:B], lens[B - 1:B]))
    finally:
        engine.set_bigru_mode(0)
    for b in range(B):
        assert np.all(got[b, lens[b]:] == 0)
    assert rel_l1(got, ref) < 1e-4, rel_l1(got, ref)
    assert np.array_equal(alone[0], got[B - 1])
    base = _np(engine.encode(ids, lens))
    assert rel_l1(got, base) < 1e-5


@pytest.mark.parametrize('mode', [1, 2])
@pytest.mark.parametrize('B,T', [(2, 15), (3, 100), (18, 60), (2, 1000)])
def test_postnet_bigru_mfma_mode(engine, hp, weights, B, T, mode):
    rng = np.random.default_rng(B + T)
    mel = (0.3 * rng.standard_normal((B, T, 80))).astype(np.float32)
    ref = O.postnet(mel.astype(np.float64), O.cast_weights(weights, np.float64), hp)
    engine.set_bigru_mode(mode)
    try:
        got = _np(engine.postnet(mel))
        one = _np(engine.postnet(mel[1:2]))
    finally:
        engine.set_bigru_mode(0)
    assert rel_l1(got, ref) < 1e-4, rel_l1(got, ref)
    assert np.array_equal(one[0], got[1])


@pytest.mark.parametrize('mode', [1, 2])
def test_bigru_mfma_mode_edge_lengths(engine, hp, weights, mode):
    """Throughput BiGRU scan: utterances of length 0, 1 and T among full ones in a batch that leaves a column group
    partly empty; against the one-utterance kernel (different summation order: 1e-5) and the masking rule (A.6)."""
    rng = np.random.default_rng(77)
    B, T_in = 21, 9
    ids, lens = synth_ids(rng, B, T_in)
    lens[1], lens[2], lens[5], lens[20] = 0, 1, T_in, 1
    base = _np(engine.encode(ids, lens))
    engine.set_bigru_mode(mode)
    try:
        got = _np(engine.encode(ids, lens))
    finally:
        engine.set_bigru_mode(0)
    for b in range(B):
        assert np.all(got[b, lens[b]:] == 0)
    assert np.all(np.isfinite(got))
    assert max_abs(got, base) < 1e-5 * max(1.0, float(np.abs(base).max()))
