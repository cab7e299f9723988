"""GPU tests of the host-side API mirror (Tacotron.initialize, Synthesizer, tools.audio) and of
the BASELINE.json configurations at full size (parity where the oracle finishes in seconds,
size-independent properties otherwise)."""
import os

import numpy as np
import pytest

from oracle import taco_oracle as O
from tests.util import rel_l1, synth_ids

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def test_tacotron_initialize_api(hp, weights):
    from tacotron_amd.data.batch import Batch
    from tacotron_amd.model.tacotron import Tacotron
    from tacotron_amd.hparams import HParams
    hp2 = HParams(max_iters=6)
    ids, lens = synth_ids(np.random.default_rng(1), 2, 9)
    m = Tacotron(hp2, weights=weights)
    m.initialize(Batch((ids, lens, None, None), prep=False))
    ref = O.tacotron_forward(ids, lens, weights, hp2)
    assert m.inputs is ids and m.input_lengths is lens and m.mel_targets is None
    assert tuple(m.mel_outputs.shape) == (2, 30, 80) and tuple(m.linear_outputs.shape) == (2, 30, 1025)
    assert tuple(m.alignments.shape) == (2, 9, 6)
    assert rel_l1(_np(m.mel_outputs), ref['mel']) < 1e-3 and rel_l1(_np(m.linear_outputs), ref['linear']) < 1e-3
    m2 = Tacotron(hp2, weights=weights).initialize(ids, lens)      # convenience form
    assert np.array_equal(_np(m2.mel_outputs), _np(m.mel_outputs))
    with pytest.raises(ValueError):
        m.initialize(np.full((1, 4), 95, np.int32), np.array([4], np.int32))
    with pytest.raises(ValueError):
        m.initialize(ids, np.array([9, 10], np.int32))
    m.engine.close(); m2.engine.close()


def test_synthesizer_writes_reference_layout(tmp_path, monkeypatch):
    import wave
    from tacotron_amd import synthesizer as S
    from tacotron_amd.hparams import hparams
    old = hparams.max_iters, hparams.griffin_lim_iters
    hparams.max_iters, hparams.griffin_lim_iters = 8, 3
    try:
        s = S.Synthesizer().load_random(seed=3)
        d = str(tmp_path / 'synthesized')
        os.makedirs(d)
        s.synthesize('Halló heimur, this is a test.', d)
        s.synthesize('second sentence', d)
        lines = open(os.path.join(d, 'index.txt')).read().splitlines()
        assert len(lines) == 2 and lines[1].endswith('|second sentence')
        assert lines[0].split('|')[:2] == [os.path.join(d, 'text', 'text-001.txt'), os.path.join(d, 'wavs', 'synth-001.wav')]
        assert open(os.path.join(d, 'text', 'text-002.txt')).read() == 'second sentence'
        with wave.open(os.path.join(d, 'wavs', 'synth-001.wav')) as f:
            assert f.getframerate() == 24000 and f.getnframes() > 0
        with pytest.raises(FileNotFoundError):
            S.Synthesizer().load(str(tmp_path / 'model.ckpt'), 5)
        s.model.engine.close()
    finally:
        hparams.max_iters, hparams.griffin_lim_iters = old


def test_checkpoint_roundtrip_through_synthesizer(tmp_path, hp, weights):
    from tacotron_amd import synthesizer as S
    from tacotron_amd.hparams import hparams
    from tacotron_amd.weights import save_npz
    save_npz(str(tmp_path / 'model.ckpt-12.npz'), weights)
    old = hparams.max_iters
    hparams.max_iters = 3
    try:
        s = S.Synthesizer().load(str(tmp_path / 'model.ckpt'), 12)
        ids, lens = synth_ids(np.random.default_rng(2), 1, 6)
        s.model.initialize(ids, lens)
        ref = O.tacotron_forward(ids, lens, weights, hp, max_iters=3)
        assert rel_l1(_np(s.model.linear_outputs), ref['linear']) < 1e-3
        s.model.engine.close()
    finally:
        hparams.max_iters = old


def test_tf_checkpoint_through_synthesizer(tmp_path, hp, weights):
    """Synthesizer.load on a TF V2 bundle under the reference's variable names (synthesizer.py:32-34); the bundle
    comes from this package's own writer (no TF here: real-checkpoint compatibility is unpinned)."""
    from tacotron_amd import synthesizer as S
    from tacotron_amd.hparams import hparams
    from tacotron_amd.tf_checkpoint import save_tacotron_weights
    save_tacotron_weights(str(tmp_path / 'model.ckpt-77'), weights)
    old = hparams.max_iters
    hparams.max_iters = 3
    try:
        s = S.Synthesizer().load(str(tmp_path / 'model.ckpt'), 77)
        ids, lens = synth_ids(np.random.default_rng(3), 2, 9)
        s.model.initialize(ids, lens)
        ref = O.tacotron_forward(ids, lens, weights, hp, max_iters=3)
        assert rel_l1(_np(s.model.linear_outputs), ref['linear']) < 1e-3
        s.model.engine.close()
    finally:
        hparams.max_iters = old


def test_full_length_parity_cfg1(engine, hp, weights):
    """BASELINE configs[0] shape (T_in=150, 200 decoder steps -> 1000 frames) against the fp64 oracle after all 200
    feedback steps: mel/linear within 1e-3 relative L1; attention argmax identical wherever the oracle's margin
    is above fp32 resolution."""
    ids, lens = synth_ids(np.random.default_rng(0), 2, 150, ragged=False)
    ref = O.tacotron_forward(ids, lens, weights, hp)
    o = engine.finish(engine.synthesize(ids, lens))
    assert o['steps_run'] == 200
    e_mel, e_lin = rel_l1(_np(o['mel']), ref['mel']), rel_l1(_np(o['linear']), ref['linear'])
    al, al_r = _np(o['alignments']), ref['alignments']
    top = np.sort(al_r, axis=1)
    decided = (top[:, -1] - top[:, -2]) > 1e-6
    same = al.argmax(1) == al_r.argmax(1)
    print('cfg1: mel %.2e linear %.2e argmax same %d/%d decided %d' % (e_mel, e_lin, same.sum(), same.size, decided.sum()))
    assert e_mel < 1e-3 and e_lin < 1e-3
    assert np.all(same[decided])


def test_full_length_parity_tile_decoder(engine, hp, weights):
    """The same configuration through the throughput form of the decode loop (mode 3, one workgroup per 16-utterance
    tile): all 200 feedback steps against the fp64 oracle, to the same bar."""
    ids, lens = synth_ids(np.random.default_rng(0), 2, 150, ragged=False)
    ref = O.tacotron_forward(ids, lens, weights, hp)
    try:
        engine.set_decode_graph(3)
        o = engine.finish(engine.synthesize(ids, lens))
    finally:
        engine.set_decode_graph(2)
    assert o['steps_run'] == 200
    e_mel, e_lin = rel_l1(_np(o['mel']), ref['mel']), rel_l1(_np(o['linear']), ref['linear'])
    al, al_r = _np(o['alignments']), ref['alignments']
    top = np.sort(al_r, axis=1)
    decided = (top[:, -1] - top[:, -2]) > 1e-6
    same = al.argmax(1) == al_r.argmax(1)
    print('cfg1 (mode 3): mel %.2e linear %.2e argmax same %d/%d decided %d' % (e_mel, e_lin, same.sum(), same.size, decided.sum()))
    assert e_mel < 1e-3 and e_lin < 1e-3
    assert np.all(same[decided])


def test_ragged_batch_padding_semantics(engine, hp, weights):
    """A.6/A.8: pad id 0 embeds to a learned row, only the encoder BiGRU masks, attention runs over padded steps."""
    ids, lens = synth_ids(np.random.default_rng(9), 4, 60, ragged=True, min_len=20)
    ref = O.tacotron_forward(ids, lens, weights, hp, max_iters=12)
    o = engine.finish(engine.synthesize(ids, lens, max_iters=12))
    for k in ('enc_out', 'mel', 'linear', 'alignments'):
        assert rel_l1(_np(o[k]), ref[k]) < 1e-3, k
    # padded encoder steps still receive attention mass (no masking in the reference graph)
    b = int(np.argmin(lens))
    assert _np(o['alignments'])[b, lens[b]:, :].sum() > 0


def test_cfg2_full_size_properties(engine, hp):
    """configs[1] (B=32, T_in=150, 200 steps): oracle-free properties at full size."""
    import torch
    ids, lens = synth_ids(np.random.default_rng(0), 32, 150, ragged=False)
    o = engine.synthesize(ids, lens)
    torch.cuda.synchronize()
    assert int(o['steps'].item()) == 200
    mel, lin, al = o['mel'], o['linear'], o['alignments']
    assert tuple(mel.shape) == (32, 1000, 80) and tuple(lin.shape) == (32, 1000, 1025) and tuple(al.shape) == (32, 150, 200)
    assert bool(torch.isfinite(mel).all()) and bool(torch.isfinite(lin).all())
    assert torch.allclose(al.sum(1), torch.ones_like(al.sum(1)), atol=1e-5)
    # shard invariance: any sub-batch gives bit-identical per-utterance results
    mel_c, lin_c = mel.clone(), lin.clone()
    o2 = engine.synthesize(ids[8:12], lens[8:12])
    assert torch.equal(o2['mel'], mel_c[8:12]) and torch.equal(o2['linear'], lin_c[8:12])
    # determinism: same call twice, bit-identical
    o3 = engine.synthesize(ids[8:12], lens[8:12])
    assert torch.equal(o3['linear'], lin_c[8:12])


def test_cfg4_long_form(engine, hp, weights):
    """configs[3]: Ivona long-form, max_iters=400 (2000 frames), 400-character encoder memory."""
    ids, lens = synth_ids(np.random.default_rng(4), 1, 400, ragged=False)
    ref = O.tacotron_forward(ids, lens, weights, hp, max_iters=400)
    o = engine.finish(engine.synthesize(ids, lens, max_iters=400))
    assert o['steps_run'] == 400 and tuple(o['mel'].shape) == (1, 2000, 80)
    assert rel_l1(_np(o['mel']), ref['mel']) < 1e-3 and rel_l1(_np(o['linear']), ref['linear']) < 1e-3


def test_cfg5_icelandic_text(hp, weights):
    """configs[4]: Icelandic charset through the same 95-symbol table (ids 75-84 present), B=1, in the hipGraph-captured
    step chain the config names (decode mode 1) and in the default persistent kernel."""
    from tacotron_amd.model.tacotron import Tacotron
    from tacotron_amd.hparams import HParams
    from tacotron_amd.text.text_tools import text_to_onehot
    ids = np.asarray(text_to_onehot('Það er gott veður í dag, þökk sé æðri máttarvöldum.'), np.int32)[None]
    assert ((ids >= 75) & (ids <= 84)).any()
    lens = np.array([ids.shape[1]], np.int32)
    hp2 = HParams(max_iters=20)
    m = Tacotron(hp2, weights=weights)
    ref = O.tacotron_forward(ids, lens, weights, hp2)
    for mode in (1, 2):
        m.engine.set_decode_graph(mode)
        m.initialize(ids, lens)
        assert rel_l1(_np(m.mel_outputs), ref['mel']) < 1e-3 and rel_l1(_np(m.linear_outputs), ref['linear']) < 1e-3
        assert rel_l1(_np(m.alignments), ref['alignments']) < 1e-3
    m.engine.close()


def test_griffin_lim_full_length(engine, hp):
    """One utterance at the default length (T=1000 frames -> 300 900 samples), all 60 iterations."""
    rng = np.random.default_rng(11)
    t = np.linspace(0, 1, 1000)[:, None]
    f = np.linspace(0, 1, 1025)[None, :]
    lin = np.clip(0.55 + 0.35 * np.sin(40 * t + 9 * f) * np.exp(-2 * f) + 0.05 * rng.standard_normal((1000, 1025)), 0, 1)
    lin = lin.astype(np.float32)
    ref = O.spectrogram_tensorflow_inv(lin.astype(np.float64), hp)
    got = _np(engine.griffin_lim(lin))
    assert got.shape == ref.shape == (300900,)
    err = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    print('GL full length: relL2 %.2e' % err)
    assert err < 1e-3
    # audio tools wrapper + host post-processing run end to end
    from tacotron_amd.tools import audio
    wav = audio.pre_emphasis_inv(_np(audio.inv_spectrogram(lin, engine)))
    assert wav.shape == (300900,) and np.isfinite(wav).all()
    assert 0 < audio.find_endpoint(wav) <= len(wav)


def test_griffin_lim_row_slice_input(engine, hp):
    """The vocoder's first kernel reads the caller's spectrogram four values at a time when it can: a row slice of a
    larger device tensor (4-byte aligned only) must give the bits of an aligned copy."""
    import torch
    rng = np.random.default_rng(21)
    lin = rng.random((2, 7, hp.num_freq)).astype(np.float32)
    big = torch.zeros(2 * 7 * hp.num_freq + 3, device='cuda')
    view = big[3:].view(2, 7, hp.num_freq)
    view.copy_(torch.from_numpy(lin))
    assert view.data_ptr() % 16 != 0
    a = _np(engine.griffin_lim(view, iters=3)).copy()
    b = _np(engine.griffin_lim(lin, iters=3))
    assert np.array_equal(a, b)


@pytest.mark.parametrize('B,T,iters', [(1, 3, 3), (4, 3, 2), (6, 4, 2), (5, 5, 2), (3, 6, 2), (1, 16, 2), (2, 19, 3),
                                       (1, 17, 3), (2, 18, 2), (3, 40, 4), (1, 8, 1), (2, 40, 0), (4, 27, 3), (5, 200, 4),
                                       (33, 100, 3), (2, 1000, 5)])
def test_griffin_lim_run_records_match_frame_state(engine, hp, monkeypatch, B, T, iters):
    """Between iterations the vocoder keeps the overlap-added signal as run records (the sums of a run of frames plus
    the raw head frames its neighbours need) instead of the windowed frames; the additions per sample and their order
    are the same, so the waveform must equal the frame-state result bit for bit: runs of 8 and of 16 frames, whole and
    short last runs, utterances of a single run, last runs of one or two frames and utterances whose records would not
    fit the frame buffers (which keep the frame state)."""
    rng = np.random.default_rng(1000 * B + T)
    lin = rng.random((B, T, hp.num_freq)).astype(np.float32)
    monkeypatch.setenv('TACO_GL_FRAME_STATE', '1')
    monkeypatch.delenv('TACO_GL_RUN', raising=False)
    ref = _np(engine.griffin_lim(lin, iters=iters)).copy()
    assert np.isfinite(ref).all()
    monkeypatch.setenv('TACO_GL_FRAME_STATE', '0')
    for run in ('8', '16'):
        monkeypatch.setenv('TACO_GL_RUN', run)
        got = _np(engine.griffin_lim(lin, iters=iters))
        assert np.array_equal(ref, got), 'run records of %s frames differ from the frame state' % run


def test_pipeline_in_flight_matches_lone_engine(engine, hp, weights):
    """SynthesisPipeline: 5 different batches through 3 engines in flight come back in order and bit-identical to the
    same batches run one at a time (no kernel's reduction order depends on what else is running)."""
    from tacotron_amd.pipeline import SynthesisPipeline
    rng = np.random.default_rng(77)
    batches = [synth_ids(rng, B, T_in) for B, T_in in [(3, 20), (1, 33), (4, 12), (2, 20), (3, 27)]]
    S = 7
    pipe = SynthesisPipeline(weights, hp, depth=3)
    try:
        got = list(pipe.run(iter(batches), max_iters=S, to_wav=True, gl_iters=2))
    finally:
        pipe.close()
    assert len(got) == len(batches)
    for (ids, lens), o in zip(batches, got):
        ref = engine.finish(engine.synthesize(ids, lens, max_iters=S, to_wav=True, gl_iters=2), to_wav=True, gl_iters=2)
        assert o['steps_run'] == ref['steps_run'] == S
        for name in ('mel', 'linear', 'alignments', 'wav'):
            assert np.array_equal(o[name].numpy(), _np(ref[name])), name


def test_persist_budget_serialises_decodes_without_changing_results(engine, hp, weights):
    """taco_set_persist_budget: with room for one 16-utterance decode at a time, decodes of two engines issued
    back to back still both complete (the second waits on its stream for the first) and give the bits of the
    default budget; a decode larger than the budget is admitted alone; negative arguments are refused."""
    import torch
    from tacotron_amd.engine import TacoEngine, TacoError
    rng = np.random.default_rng(5)
    ids, lens = synth_ids(rng, 16, 24)
    ids2, lens2 = synth_ids(rng, 20, 24)
    S = 6
    ref = _np(engine.synthesize(ids, lens, max_iters=S)['mel']).copy()
    ref2 = _np(engine.synthesize(ids2, lens2, max_iters=S)['mel']).copy()
    other = TacoEngine(weights, hp)
    try:
        engine.set_persist_budget(32)
        a = engine.synthesize(ids, lens, max_iters=S)
        b = other.synthesize(ids, lens, max_iters=S)
        c = other.synthesize(ids2, lens2, max_iters=S)          # 64 workgroups > budget: runs alone
        torch.cuda.synchronize()
        engine.decode_status(); other.decode_status()
        assert np.array_equal(_np(a['mel']), ref) and np.array_equal(_np(b['mel']), ref)
        assert np.array_equal(_np(c['mel']), ref2)
        with pytest.raises(TacoError):
            engine.set_persist_budget(-1)
    finally:
        engine.set_persist_budget(0)
        other.close()


def test_synthesizer_extensions(tmp_path):
    """synthesize_long (split + one batch + join), synthesize_many (pipeline, input order kept), plot=True."""
    from tacotron_amd import synthesizer as S
    from tacotron_amd.hparams import hparams
    old = hparams.max_iters, hparams.griffin_lim_iters
    hparams.max_iters, hparams.griffin_lim_iters = 6, 2
    try:
        s = S.Synthesizer().load_random(seed=3)
        text = 'First sentence here. And a second one! Is there a third?'
        pieces = S.split_text(text)
        assert pieces == ['First sentence here.', 'And a second one!', 'Is there a third?']
        singles = s.synthesize_batch(pieces)
        long = s.synthesize_long(text, pause_s=0.01)
        assert len(long) == sum(len(w) for w in singles) + 2 * int(0.01 * hparams.sample_rate)
        assert np.array_equal(long[:len(singles[0])], singles[0])
        texts = ['text number %d' % i for i in range(7)]
        many = list(s.synthesize_many(texts, batch_size=3, depth=2))
        assert [t for t, _ in many] == texts
        # batch composition does not change an utterance's result beyond padding effects: same text alone
        alone = s.synthesize_batch([texts[4]])[0]
        assert many[4][1].shape == alone.shape and np.allclose(many[4][1], alone, atol=1e-4)
        d = str(tmp_path / 'synthesized')
        os.makedirs(d)
        s.synthesize('plot me', d, plot=True)
        png = os.path.join(d, 'alignments', 'align-001.png')
        assert os.path.getsize(png) > 1000 and open(png, 'rb').read(4) == b'\x89PNG'
    finally:
        hparams.max_iters, hparams.griffin_lim_iters = old


def test_c_abi_error_paths(engine, hp, weights):
    """Error behaviour at the C boundary (include/taco.h conventions): <0 + a message in taco_last_error, nothing
    launched, the handle stays usable afterwards."""
    import ctypes
    import torch
    from tacotron_amd import _lib
    lib, h = engine.lib, engine.handle
    ids, lens = synth_ids(np.random.default_rng(3), 2, 9)
    d_ids = torch.from_numpy(ids).cuda()
    d_lens = torch.from_numpy(lens).cuda()
    enc = torch.empty(2, 9, 256, device='cuda')
    ws = torch.empty(int(lib.taco_workspace_bytes(h, 2, 9, 4)), dtype=torch.uint8, device='cuda')
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def err():
        return _lib.last_error()

    # null pointers, empty batch, workspace too small
    assert lib.taco_encode(h, None, d_lens.data_ptr(), 2, 9, enc.data_ptr(), ws.data_ptr(), ws.numel(), s) < 0
    assert 'null' in err()
    assert lib.taco_encode(h, d_ids.data_ptr(), d_lens.data_ptr(), 0, 9, enc.data_ptr(), ws.data_ptr(), ws.numel(), s) < 0
    assert 'B=0' in err()
    assert lib.taco_encode(h, d_ids.data_ptr(), d_lens.data_ptr(), 2, 9, enc.data_ptr(), ws.data_ptr(), 1024, s) < 0
    assert 'workspace' in err()
    assert lib.taco_workspace_bytes(h, 0, 9, 4) == 0 and lib.taco_workspace_bytes(None, 2, 9, 4) == 0
    lin = torch.rand(1, 6, hp.num_freq, device='cuda')
    wav = torch.empty(1, 5 * 300 + 1200, device='cuda')
    assert lib.taco_griffin_lim(h, lin.data_ptr(), 1, 6, -1, wav.data_ptr(), ws.data_ptr(), ws.numel(), s) < 0
    assert 'iters' in err()
    assert lib.taco_griffin_lim_librosa(h, lin.data_ptr(), None, 1, 4, 1, wav.data_ptr(), ws.data_ptr(), ws.numel(), s) < 0
    assert 'reflect' in err()
    assert lib.taco_set_decode_graph(h, 7) < 0
    # taco_create: a missing variable and a wrong shape are reported by name (Saver.restore's NotFound / shape errors)
    from tacotron_amd.engine import TacoEngine, TacoError
    w2 = dict(weights)
    name = sorted(k for k in w2 if 'attention' in k or 'gru' in k.lower())[0]
    del w2[name]
    with pytest.raises((TacoError, KeyError, ValueError)) as ei:
        TacoEngine(w2, hp)
    assert name.split('/')[-1] in str(ei.value) or name in str(ei.value)
    w3 = dict(weights)
    w3[name] = np.zeros((3, 3), np.float32)
    with pytest.raises((TacoError, ValueError)) as ei:
        TacoEngine(w3, hp)
    assert name in str(ei.value)
    # the handle still works
    ref = O.encode(ids, lens, weights, hp)
    assert rel_l1(_np(engine.encode(ids, lens)), ref) < 1e-3


@pytest.mark.parametrize('overrides', [
    dict(outputs_per_step=3, encoder_K=5, decoder_K=4, encoder_num_highway_layers=2, decoder_num_highway_layers=3),
    dict(prenet_depths=[128, 64], attention_depth=128, decoder_depth=128, postnet_depth=128, embedded_depth=128,
         encoder_bank_num_filters=64, decoder_bank_num_filters=96, encoder_proj_num_filters=[96, 64],
         decoder_proj_num_filters=[128, 80], encoder_highway_depth=64, decoder_highway_depth=96, outputs_per_step=2),
    # max-pool windows other than the reference's 2 (hparams.py:32,42): 1 = identity, wider = a kernel of its own
    dict(encoder_pooling_width=1, decoder_pooling_width=3),
    dict(encoder_pooling_width=4, decoder_pooling_width=5, encoder_K=6, decoder_K=3),
    # BiGRU cell counts other than 128 (hparams.py:37,47): the generic scan; the encoder's also sets the memory depth
    dict(encoder_gru_num_cells=64, decoder_gru_num_cells=96),
    dict(encoder_gru_num_cells=160, decoder_gru_num_cells=32, outputs_per_step=4),
])
def test_non_default_hparams(overrides):
    """The reference's layer sizes are hparams (`--hparams k=v`): a model with other sizes runs through the general
    kernels (the persistent decoder is specialised to the default sizes and steps aside) and matches the oracle."""
    from tacotron_amd.engine import TacoEngine
    from tacotron_amd.hparams import HParams
    from tacotron_amd.weights import init_weights
    hp2 = HParams(max_iters=5, griffin_lim_iters=2, **overrides)
    w = init_weights(hp2, seed=77, randomize_bn=True, randomize_bias=True)
    ids, lens = synth_ids(np.random.default_rng(5), 3, 13)
    ref = O.tacotron_forward(ids, lens, w, hp2)
    eng = TacoEngine(w, hp2)
    try:
        o = eng.finish(eng.synthesize(ids, lens, max_iters=5))
        r = hp2.outputs_per_step
        assert tuple(o['mel'].shape) == (3, 5 * r, hp2.num_mels) and o['steps_run'] == 5
        for name in ('enc_out', 'mel', 'linear', 'alignments'):
            assert rel_l1(_np(o[name]), ref[name]) < 1e-3, name
    finally:
        eng.close()


def test_non_default_frame_geometry(weights):
    """frame_shift_ms / frame_length_ms are hparams (hop 240, window 960 here instead of 300 / 1200): the Griffin-Lim
    and STFT kernels' run-time-geometry instantiations against the oracle."""
    from tacotron_amd.engine import TacoEngine
    from tacotron_amd.hparams import HParams
    hp2 = HParams(frame_shift_ms=10, frame_length_ms=40)
    assert hp2.stft_params() == (2048, 240, 960)
    eng = TacoEngine(weights, hp2)
    try:
        rng = np.random.default_rng(12)
        lin = rng.uniform(0.3, 1.0, size=(2, 23, 1025)).astype(np.float32)
        ref = O.spectrogram_tensorflow_inv(lin.astype(np.float64), hp2, iters=3)
        got = _np(eng.griffin_lim(lin, iters=3))
        assert got.shape == ref.shape == (2, 22 * 240 + 960)
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-3
        u = rng.random((2, 23, 1025))
        ref = np.stack([O.griffin_lim_librosa(lin[b].T.astype(np.float64), hp2, u[b].T, 2) for b in range(2)])
        got = _np(eng.griffin_lim_librosa(lin, u.astype(np.float32), iters=2))
        assert got.shape == ref.shape and np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-3
        y = (0.1 * rng.standard_normal(5000)).astype(np.float32)
        l, m = eng.spectrogram(y)
        assert np.abs(_np(l)[0].T - O.spectrogram_librosa(y.astype(np.float64), hp2)).max() < 2e-3
        assert np.abs(_np(m)[0].T - O.mel_spectrogram_librosa(y.astype(np.float64), hp2)).max() < 2e-3
    finally:
        eng.close()


def test_audio_tools_reference_signatures(hp):
    """tools.audio with the reference's own call shapes (no engine argument): spectrogram(y), mel_spectrogram(y),
    spectrogram_inv(spect), inv_spectrogram(spect) run on the default engine."""
    from tacotron_amd.tools import audio
    rng = np.random.default_rng(3)
    y = (0.1 * rng.standard_normal(6000)).astype(np.float32)
    lin = audio.spectrogram(y)
    mel = audio.mel_spectrogram(y)
    assert lin.shape == (1025, 21) and mel.shape == (80, 21)
    assert np.abs(lin - O.spectrogram_librosa(y.astype(np.float64), hp)).max() < 2e-3
    np.random.seed(7)
    wav = audio.spectrogram_inv(lin)
    np.random.seed(7)
    u = np.random.rand(1025, 21)
    ref = O.pre_emphasis_inv(O.griffin_lim_librosa(lin.astype(np.float64), hp, u), hp.preemphasis)
    assert wav.shape == ref.shape == (20 * 300,) and np.linalg.norm(wav - ref) / np.linalg.norm(ref) < 2e-2
    w2 = audio.inv_spectrogram(lin.T)
    assert tuple(w2.shape) == (20 * 300 + 1200,)


def test_process_utterance_writes_reference_format(tmp_path, hp):
    """data/data_load.py:114-146: a wav file -> <prefix>-spec-%05d.npy [T, 1025] and <prefix>-mel-%05d.npy [T, 80],
    float32, loadable without pickle; values against the oracle's librosa-flavoured analysis of the same samples."""
    from tacotron_amd.data.data_load import process_utterance
    from tacotron_amd.tools import audio
    rng = np.random.default_rng(4)
    y = 0.2 * np.sin(2 * np.pi * 330 * np.arange(9000) / 24000) + 0.02 * rng.standard_normal(9000)
    wav_path = str(tmp_path / 'u.wav')
    audio.save_wav(y.copy(), wav_path)
    out = process_utterance(str(tmp_path), 7, wav_path, 'some text', prefix='ljs', speaker='a')
    assert out == ('ljs-spec-00007.npy', 'ljs-mel-00007.npy', 31, 'some text', 'a')
    spec = np.load(str(tmp_path / out[0]), allow_pickle=False)
    mel = np.load(str(tmp_path / out[1]), allow_pickle=False)
    assert spec.dtype == mel.dtype == np.float32 and spec.shape == (31, 1025) and mel.shape == (31, 80)
    w = audio.load_wav(wav_path).astype(np.float64)
    assert np.abs(spec.T - O.spectrogram_librosa(w, hp)).max() < 2e-3
    assert np.abs(mel.T - O.mel_spectrogram_librosa(w, hp)).max() < 2e-3


def test_persistent_abort_costs_one_timeout_then_mode1(hp, weights):
    """Failure path of the persistent decoder (include/taco.h, taco_decode_status): a launch whose bounded wait ran
    out sets the handle's sticky word; launches queued behind it return at once (not one time-out each), the next
    taco_decode refuses once (the previous outputs were invalid), later calls run the per-stage path and are correct,
    and a mode-1 decode never reports the stale status of an earlier persistent launch (the case of round 1's t5.log).
    The time-out itself is simulated with taco_debug_poison_decoder: no GPU is hung for this test."""
    import time
    import torch
    from tacotron_amd.engine import TacoEngine, TacoError
    eng = TacoEngine(weights, hp)
    try:
        rng = np.random.default_rng(21)
        enc = (0.5 * rng.standard_normal((3, 20, 256))).astype(np.float32)
        S = 6
        ref_mel, ref_al, _ = O.decode(enc.astype(np.float64), weights, hp, max_iters=S)
        m0, a0, _ = eng.decode(enc, max_iters=S)
        torch.cuda.synchronize()
        eng.decode_status()
        assert rel_l1(_np(m0), ref_mel) < 1e-3
        m0 = _np(m0).copy()
        # ---- path A: nobody asks for the status; the next decode call finds out by itself
        assert eng.lib.taco_debug_poison_decoder(eng.handle) == 0
        t0 = time.perf_counter()
        refused = False
        for _ in range(3):                       # queued persistent launches: each must leave at once
            try:
                eng.decode(enc, max_iters=S)
            except TacoError as ex:              # (the host may already have seen the first one's abort)
                assert 'persistent decoder' in str(ex)
                refused = True
                break
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 5.0    # one wait is bounded at 10 s; three time-outs would be 30 s
        if not refused:                          # the first call after the aborted launch completed refuses, once
            with pytest.raises(TacoError, match='persistent decoder'):
                eng.decode(enc, max_iters=S)
        m1, a1, _ = eng.decode(enc, max_iters=S)             # per-stage path from now on
        torch.cuda.synchronize()
        eng.decode_status()                                   # no stale status from the aborted launches
        assert rel_l1(_np(m1), ref_mel) < 1e-3 and rel_l1(_np(a1), ref_al) < 1e-3
        assert np.abs(_np(m1) - m0).max() < 1e-4
        # ---- re-arm: persistent mode again, bit-identical to the first run
        eng.set_decode_graph(2)
        m2, _, _ = eng.decode(enc, max_iters=S)
        torch.cuda.synchronize()
        eng.decode_status()
        assert np.array_equal(_np(m2), m0)
        # ---- path B: the caller asks (engine.finish does): decode_status reports, the handle falls back
        assert eng.lib.taco_debug_poison_decoder(eng.handle) == 0
        eng.decode(enc, max_iters=S)
        torch.cuda.synchronize()
        with pytest.raises(TacoError, match='persistent decoder'):
            eng.decode_status()
        m3, _, _ = eng.decode(enc, max_iters=S)
        torch.cuda.synchronize()
        eng.decode_status()
        assert np.array_equal(_np(m3), _np(m1))
    finally:
        eng.close()


def test_cfg3_full_size_to_waveform(engine, hp):
    """configs[2] per GPU (a shard of 32 of the 256 utterances): B=32, T_in=150, 200 decoder steps, all 60
    Griffin-Lim iterations, end to end to the waveform in one taco_synthesize (synthesizer.py:26,39-45 run for a
    batch). Full-size properties: shapes, finiteness, utterance 0 bit-identical to the same utterance run alone
    (shard invariance through the vocoder); and the waveform of that utterance against the fp64 oracle's
    Griffin-Lim of the same linear spectrogram."""
    import torch
    ids, lens = synth_ids(np.random.default_rng(0), 32, 150, ragged=False)
    o = engine.finish(engine.synthesize(ids, lens, to_wav=True), to_wav=True)
    assert o['steps_run'] == 200
    wav, lin = o['wav'], o['linear']
    assert tuple(wav.shape) == (32, 999 * 300 + 1200) and tuple(lin.shape) == (32, 1000, 1025)
    assert bool(torch.isfinite(wav).all()) and float(wav.abs().max()) > 0
    wav0, lin0 = wav[0].clone(), _np(lin[0]).copy()
    o1 = engine.finish(engine.synthesize(ids[:1], lens[:1], to_wav=True), to_wav=True)
    assert np.array_equal(_np(o1['linear'][0]), lin0)
    assert torch.equal(o1['wav'][0], wav0)
    ref = O.spectrogram_tensorflow_inv(lin0.astype(np.float64), hp)
    err = float(np.linalg.norm(_np(wav0) - ref) / np.linalg.norm(ref))
    print('cfg3: wav of utterance 0 vs oracle Griffin-Lim, relL2 %.2e' % err)
    assert err < 1e-3


def test_sharded_job_matches_single_device(engine, hp, weights):
    """Multi-GPU product entry: SynthesisPipeline(devices=[0, 0]).synthesize_sharded length-sorts a job,
    deals it to one engine set per listed device (here two sets on the one GPU of the box), runs batches in flight and
    restores the input order; every utterance is bit-identical to the same job run as one batch on one engine."""
    from tacotron_amd.pipeline import SynthesisPipeline
    ids, lens = synth_ids(np.random.default_rng(5), 10, 24, ragged=True, min_len=6)
    S = 6
    ref = engine.finish(engine.synthesize(ids, lens, max_iters=S, to_wav=True, gl_iters=2), to_wav=True, gl_iters=2)
    pipe = SynthesisPipeline(weights, hp, depth=2, devices=[0, 0])
    try:
        got = pipe.synthesize_sharded(ids, lens, batch_size=3, max_iters=S, to_wav=True, gl_iters=2,
                                      keys=('mel', 'linear', 'alignments', 'wav'))
        assert len(got) == 10
        for b, r in enumerate(got):
            assert r['steps_run'] == S
            for name in ('mel', 'linear', 'alignments', 'wav'):
                assert np.array_equal(r[name], _np(ref[name][b])), (b, name)
        # trim=True: each batch padded to its own longest utterance = the reference run on that smaller batch
        got2 = pipe.synthesize_sharded(ids, lens, batch_size=10, max_iters=S, trim=True, keys=('mel',))
        order = np.argsort(-lens, kind='stable')
        half = order[5:]                                # the shorter shard: padded width < 24
        assert lens[half].max() < 24
        o_half = engine.finish(engine.synthesize(ids[half][:, :lens[half].max()], lens[half], max_iters=S))
        for j, b in enumerate(half):
            assert np.array_equal(got2[b]['mel'], _np(o_half['mel'][j]))
        # default keys: the waveform alone for a to-waveform job (synthesizer.py:43 fetches nothing else); alignments
        # of a trimmed job have the batch's own width
        got3 = pipe.synthesize_sharded(ids, lens, batch_size=4, max_iters=S, to_wav=True, gl_iters=2)
        assert all(set(r) == {'wav', 'steps_run'} for r in got3)
        assert all(np.array_equal(r['wav'], _np(ref['wav'][b])) for b, r in enumerate(got3))
        got4 = pipe.synthesize_sharded(ids, lens, batch_size=10, max_iters=S, trim=True)
        assert set(got4[0]) == {'mel', 'linear', 'alignments', 'steps_run'}
        for j, b in enumerate(half):
            assert got4[b]['alignments'].shape == (int(lens[half].max()), S)
            assert np.array_equal(got4[b]['alignments'], _np(o_half['alignments'][j]))
    finally:
        pipe.close()
