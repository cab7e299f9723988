"""Host logic that runs without a GPU: text frontend against the reference-generated goldens,
hparams, weight specs, sharding, host DSP."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_text_matches_reference_goldens():
    """tests/golden/text_goldens.json was produced by importing /root/reference/text
    (tests/golden/make_text_goldens.py); this is the pinned part of the parity story."""
    from tacotron_amd.text.characters import chars, pad, eos
    from tacotron_amd.text.text_tools import text_to_onehot, onehot_to_text
    g = json.load(open(os.path.join(HERE, 'golden', 'text_goldens.json'), encoding='utf-8'))
    assert chars == g['chars'] and pad == g['pad'] and eos == g['eos'] and len(chars) == 95
    for c in g['cases']:
        ids = [int(i) for i in text_to_onehot(c['text'], 'basic_cleaners')]
        assert ids == c['ids'], c['text']
        assert ids[-1] == 1
        assert onehot_to_text(ids) == c['roundtrip']
    # the vector recorded in SURVEY.md 8(c)
    want = [35, 32, 39, 39, 42, 58, 64, 50, 42, 45, 39, 31, 54, 64, 82, 32, 47, 47, 28, 64, 32, 45, 64, 43, 45, 79,
            33, 64, 86, 87, 88, 64, 1]
    assert [int(i) for i in text_to_onehot("Hello,  World! Þetta er próf 123 #")] == want


def test_hparams_defaults_and_parse():
    from tacotron_amd.hparams import HParams
    hp = HParams()
    assert hp.get('max_iters') == 200 and hp.outputs_per_step == 5 and hp.num_freq == 1025
    assert hp.stft_params() == (2048, 300, 1200)          # audio.py:161-163
    hp.parse('max_iters=400,prenet_depths=[256,128],power=1.2')
    assert hp.max_iters == 400 and hp.prenet_depths == [256, 128] and hp.power == 1.2
    with pytest.raises(ValueError):
        hp.parse('no_such_param=1')
    with pytest.raises(ValueError):
        hp.parse('max_iters=1.5')
    assert HParams().max_iters == 200                     # instances do not share state


def test_hparams_match_reference_literals():
    """tests/golden/hparams_goldens.json holds the keyword literals of /root/reference/hparams.py:7-68, read by
    importing that file (tests/golden/make_refpin_goldens.py): every default of the product, name by name."""
    from tacotron_amd.hparams import HParams
    g = json.load(open(os.path.join(HERE, 'golden', 'hparams_goldens.json')))
    hp = HParams()
    mine = hp.values()
    assert set(mine) == set(g['values']), set(mine) ^ set(g['values'])
    for k, v in g['values'].items():
        assert mine[k] == v and type(mine[k]) is type(v), (k, mine[k], v)
        assert hp.get(k) == v and getattr(hp, k) == v
    assert list(hp.stft_params()) == g['stft_params']            # audio.py:150-164 evaluated by the reference
    assert len(g['values']) == 44


def _dsp_goldens():
    return np.load(os.path.join(HERE, 'golden', 'host_dsp_goldens.npz'))


@pytest.mark.parametrize('side', ['product', 'oracle'])
def test_host_dsp_matches_reference_outputs(side):
    """tests/golden/host_dsp_goldens.npz: inputs and outputs of the reference's own numpy/scipy helpers
    (tools/audio.py:15-17,81-88,91-116,212-224,254-274), produced by importing that file
    (tests/golden/make_refpin_goldens.py). Product and oracle are each compared with the reference's outputs."""
    g = _dsp_goldens()
    from oracle import taco_oracle as O
    from tacotron_amd.hparams import HParams
    from tacotron_amd.tools import audio
    hp = HParams()
    if side == 'product':
        pre_inv, pre, ep = audio.pre_emphasis_inv, audio.pre_emphasis, audio.find_endpoint
        to_i16, db2amp, amp2db = audio.wav_to_int16, audio._db_to_amp, audio._amp_to_db
        norm, norm_inv = audio._normalize, audio._normalize_inv
    else:
        pre_inv, pre = O.pre_emphasis_inv, O.pre_emphasis
        ep = lambda w, **k: O.find_endpoint(w, hp.sample_rate, **k)
        to_i16 = O.wav_to_int16
        db2amp = lambda x: np.power(10.0, x * 0.05)
        amp2db = lambda x: 20 * np.log10(np.maximum(1e-5, x))
        norm = lambda s: O._normalize(s, hp)
        norm_inv = lambda s: np.clip(s, 0, 1) * -float(hp.min_level_db) + hp.min_level_db
    # linear recurrences: same operations in the same order as lfilter -> equal to the last bits (allow 1e-12 relative)
    for x, want in ((g['pre_x'], g['pre_emphasis_inv']), (g['pre_x32'], g['pre_emphasis_inv_x32'])):
        got = pre_inv(x)
        assert got.dtype == np.float64 and np.allclose(got, want, rtol=1e-12, atol=1e-12)
    assert np.allclose(pre(g['pre_x']), g['pre_emphasis'], rtol=1e-12, atol=1e-12)
    for i in range(int(g['n_endpoint_cases'])):
        assert ep(g['ep_wav_%d' % i]) == int(g['ep_out_%d' % i]), i
    assert ep(g['ep_wav_7'], threshold_db=-30, min_silence_sec=0.5) == int(g['ep_out_7_m30_05'])
    assert np.array_equal(db2amp(g['db_in']), g['db_to_amp'])
    assert np.array_equal(amp2db(g['amp_in']), g['amp_to_db'])
    assert np.array_equal(norm(g['norm_in']), g['normalize'])
    assert np.array_equal(norm_inv(g['norm_inv_in']), g['normalize_inv'])
    for i in range(int(g['n_save_wav_cases'])):
        assert np.array_equal(to_i16(g['sw_in_%d' % i]), g['sw_pcm_%d' % i]), i
        assert int(g['sw_rate_%d' % i]) == hp.sample_rate
    if side == 'oracle':
        # the vocoder's input amplitudes (audio.py:53 / :63-64): the oracle's fused form against the reference's chain
        got = O.denorm_db_to_amp_pow(g['norm_inv_in'], hp)
        assert np.allclose(got, g['spec_amp_pow'], rtol=1e-14, atol=0)


def test_save_wav_file_holds_reference_pcm(tmp_path):
    """The file save_wav writes carries exactly the int16 samples the reference handed to librosa.output.write_wav."""
    import wave
    from tacotron_amd.tools import audio
    g = _dsp_goldens()
    for i in range(int(g['n_save_wav_cases'])):
        p = str(tmp_path / ('a%d.wav' % i))
        audio.save_wav(g['sw_in_%d' % i].copy(), p)
        with wave.open(p) as f:
            assert f.getframerate() == int(g['sw_rate_%d' % i]) and f.getsampwidth() == 2 and f.getnchannels() == 1
            pcm = np.frombuffer(f.readframes(f.getnframes()), '<i2')
        assert np.array_equal(pcm, g['sw_pcm_%d' % i])


def test_weight_specs_and_initialisers():
    from tacotron_amd.hparams import HParams
    from tacotron_amd.weights import init_weights, variable_specs, check_weights, num_params, save_npz, load_npz
    hp = HParams()
    specs = variable_specs(hp)
    names = [n for n, _, _ in specs]
    assert len(names) == len(set(names))
    w = init_weights(hp, seed=1)
    check_weights(w, hp)
    dec = sum(v.size for k, v in w.items() if k.startswith('decoder/'))
    assert dec == 1502736                                  # W_step / 4
    assert w['embedding'].shape == (95, 256) and np.abs(w['embedding']).max() <= 1.0 + 1e-6
    g = 'encoder_cbhg/bidirectional_rnn/fw/gru_cell/gates/'
    assert np.all(w[g + 'bias'] == 1.0) and w[g + 'kernel'].shape == (256, 256)
    assert np.all(w['encoder_cbhg/highway_1/T/bias'] == -1.0)
    k = w['encoder_cbhg/conv_bank/conv_bank_16/conv1d/kernel']
    assert k.shape == (16, 128, 128) and np.abs(k).max() <= np.sqrt(6.0 / (16 * 128 * 2))
    assert 6.8e6 < num_params(w) < 7.1e6
    bad = dict(w)
    bad.pop('dense/bias')
    with pytest.raises(ValueError):
        check_weights(bad, hp)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, 'model.ckpt-7.npz')
        save_npz(p, w)
        w2 = load_npz(p)
    assert set(w2) == set(w) and all(np.array_equal(w[k], w2[k]) for k in w)


def test_shard_helpers():
    from tacotron_amd.shard import shard_bounds, shard_by_length, gather_order
    assert shard_bounds(256, 8) == [(32 * i, 32 * i + 32) for i in range(8)]
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    lens = np.random.default_rng(0).integers(5, 150, size=37)
    shards = shard_by_length(lens, 4)
    cat = np.concatenate(shards)
    assert sorted(cat.tolist()) == list(range(37))
    assert all(lens[s].max() >= lens[t].max() for s, t in zip(shards, shards[1:]))
    inv = gather_order(shards)
    assert np.array_equal(cat[inv], np.arange(37))


def test_host_dsp():
    from oracle import taco_oracle as O
    from tacotron_amd.tools import audio
    rng = np.random.default_rng(0)
    x = rng.standard_normal(5000)
    assert np.allclose(audio.pre_emphasis_inv(x), O.pre_emphasis_inv(x), rtol=0, atol=1e-9)
    wav = np.concatenate([0.5 * np.sin(np.arange(30000) / 10.0), np.zeros(40000)])
    e = audio.find_endpoint(wav)
    assert e == O.find_endpoint(wav) and 30000 <= e < 40000
    assert audio.find_endpoint(np.ones(50000)) == 50000
    import tempfile, wave
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, 'a.wav')
        audio.save_wav(wav[:1000].copy(), p)
        with wave.open(p) as f:
            assert f.getframerate() == 24000 and f.getsampwidth() == 2 and f.getnframes() == 1000
            pcm = np.frombuffer(f.readframes(1000), '<i2')
    assert np.array_equal(pcm, O.wav_to_int16(wav[:1000]))


def test_batch_and_pad():
    from tacotron_amd.data.batch import Batch, pad_inputs
    ids, lens = pad_inputs([[5, 6, 1], [7, 1]])
    assert ids.tolist() == [[5, 6, 1], [7, 1, 0]] and lens.tolist() == [3, 2]
    b = Batch((ids, lens, None, None), prep=False)
    assert b.get_size() == 2 and b.get_all()[2] is None and b.get_input_lengths() is lens
    with pytest.raises(NotImplementedError):
        Batch([], prep=True)


def test_split_text_and_plot(tmp_path):
    from tacotron_amd.synthesizer import split_text
    from tacotron_amd.tools.plot import plot_alignment
    assert split_text('One. Two!  Three? four') == ['One.', 'Two!', 'Three?', 'four']
    long = 'alpha beta, ' * 30 + 'omega.'
    pieces = split_text(long, max_chars=50)
    assert all(0 < len(p) <= 50 for p in pieces) and ' '.join(pieces) == long.strip()
    assert split_text('x' * 120, max_chars=50) == ['x' * 50, 'x' * 50, 'x' * 20]
    assert split_text('   ') == []
    path = str(tmp_path / 'a.png')
    plot_alignment(np.random.default_rng(0).random((20, 30)), path, info='step 7')
    assert open(path, 'rb').read(4) == b'\x89PNG'


def test_wav_io_and_trim_silence(tmp_path):
    from tacotron_amd.tools import audio
    rng = np.random.default_rng(2)
    y = 0.3 * np.sin(2 * np.pi * 220 * np.arange(24000) / 24000) + 0.01 * rng.standard_normal(24000)
    path = str(tmp_path / 'a.wav')
    audio.save_wav(y.copy(), path)
    z = audio.load_wav(path)
    assert z.dtype == np.float32 and z.shape == y.shape
    assert np.allclose(z, y / np.abs(y).max(), atol=2e-4)           # save_wav scales to full range (audio.py:16)
    # 0.5 s of digital silence on both sides is trimmed to the frame grid (hop 512)
    padded = np.concatenate([np.zeros(12000), y, np.zeros(12000)]).astype(np.float32)
    t = audio.trim_silence(padded)
    assert 24000 <= len(t) <= 24000 + 2 * 2048 and abs(float(np.abs(t).max()) - float(np.abs(y).max())) < 1e-6
    assert len(audio.trim_silence(np.zeros(5000, np.float32))) == 6 * 512       # all frames equal the reference level: kept, cut to the frame grid
    assert np.array_equal(audio.trim_silence(y.astype(np.float32)[:100]), y.astype(np.float32)[:100])
