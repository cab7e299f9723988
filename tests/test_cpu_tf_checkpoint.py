"""TF V2 tensor-bundle reader/writer (tacotron_amd/tf_checkpoint.py). No TensorFlow and no checkpoint of the
reference exist here, so apart from the published known answers (CRC32C test vector, LevelDB table magic) the reader
is checked against this package's own writer: parity with real TF files is unpinned."""
import os
import struct

import numpy as np
import pytest

from tacotron_amd import tf_checkpoint as C
from tacotron_amd.hparams import HParams
from tacotron_amd.weights import init_weights, variable_specs


def test_crc32c_known_answers():
    assert C.crc32c(b'123456789') == 0xE3069283            # the standard CRC-32C check value
    assert C.crc32c(b'') == 0
    assert C.crc32c(bytes(32)) == 0x8A9136AA                # RFC 3720 B.4: 32 bytes of zeros
    assert C.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43       # RFC 3720 B.4: 32 bytes of ones
    assert C.mask_crc(0) == 0xa282ead8


def test_roundtrip_generic(tmp_path):
    rng = np.random.default_rng(0)
    t = {'a/b/kernel': rng.standard_normal((3, 5, 7)).astype(np.float32),
         'a/b/bias': rng.standard_normal(7).astype(np.float32),
         'a/step': np.array(1234, np.int64),
         'z': rng.integers(0, 100, (4, 2)).astype(np.int32)}
    for i in range(300):                                       # several data blocks, shared key prefixes
        t['scope/layer_%03d/kernel' % i] = rng.standard_normal((2, 3)).astype(np.float32)
    prefix = str(tmp_path / 'model.ckpt-7')
    C.write_checkpoint(prefix, t, block_bytes=512)
    idx, header = C.read_index(prefix)
    assert header['num_shards'] == 1 and set(idx) == set(t)
    assert idx['a/b/kernel']['shape'] == (3, 5, 7) and idx['a/step']['shape'] == ()
    back = C.read_checkpoint(prefix, verify_crc=True)
    assert set(back) == set(t)
    for k in t:
        assert back[k].dtype == t[k].dtype and np.array_equal(back[k], t[k]), k
    with open(prefix + '.index', 'rb') as f:
        assert struct.unpack('<Q', f.read()[-8:])[0] == 0xdb4775248b80fb57


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / 'm')
    C.write_checkpoint(prefix, {'x': np.arange(10, dtype=np.float32)})
    raw = bytearray(open(prefix + '.index', 'rb').read())
    raw[5] ^= 0x40
    open(prefix + '.index', 'wb').write(bytes(raw))
    with pytest.raises(C.CheckpointError):
        C.read_index(prefix)
    C.write_checkpoint(prefix, {'x': np.arange(10, dtype=np.float32)})
    d = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    d[3] ^= 1
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(d))
    with pytest.raises(C.CheckpointError):
        C.read_checkpoint(prefix, verify_crc=True)
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(d[:-4]))
    with pytest.raises(C.CheckpointError):
        C.read_checkpoint(prefix)


def test_tacotron_variables_roundtrip_and_diagnostics(tmp_path, capsys):
    """A training checkpoint as model/tacotron.py:169 leaves it: inference variables + Adam slots + counters."""
    hp = HParams()
    w = init_weights(hp, seed=3)
    prefix = str(tmp_path / 'model.ckpt-1000')
    full = {C.GRAPH_PREFIX + k: v for k, v in w.items()}
    k0 = C.GRAPH_PREFIX + 'embedding'
    full[k0 + '/Adam'] = np.zeros_like(w['embedding'])
    full[k0 + '/Adam_1'] = np.zeros_like(w['embedding'])
    full[C.GRAPH_PREFIX + 'global_step'] = np.array(1000, np.int32)
    full['model/optimizer/beta1_power'] = np.array(0.9, np.float32)
    C.write_checkpoint(prefix, full)
    back = C.load_tacotron_weights(prefix, hp)
    assert set(back) == {n for n, _, _ in variable_specs(hp)}
    for k in w:
        assert np.array_equal(back[k], w[k]), k
    # a missing variable and a wrong shape are named
    broken = dict(full)
    del broken[C.GRAPH_PREFIX + 'memory_layer/kernel']
    broken[C.GRAPH_PREFIX + 'dense/bias'] = np.zeros(7, np.float32)
    C.write_checkpoint(prefix, broken)
    with pytest.raises(Exception) as ei:
        C.load_tacotron_weights(prefix, hp)
    assert 'memory_layer/kernel' in str(ei.value) and 'dense/bias' in str(ei.value)
    # an unknown variable inside the graph scope is listed, not silently dropped and not fatal (strict: fatal)
    odd = dict(full)
    odd[C.GRAPH_PREFIX + 'post_cbhg/extra/kernel'] = np.zeros((2, 2), np.float32)
    C.write_checkpoint(prefix, odd)
    rep = {}
    back = C.load_tacotron_weights(prefix, hp, report=rep)
    assert rep['leftovers'] == [C.GRAPH_PREFIX + 'post_cbhg/extra/kernel'] and not rep['renamed']
    assert 'post_cbhg/extra/kernel' in capsys.readouterr().out
    assert np.array_equal(back['dense/kernel'], w['dense/kernel'])
    with pytest.raises(C.CheckpointError) as ei:
        C.load_tacotron_weights(prefix, hp, strict=True)
    assert 'post_cbhg/extra/kernel' in str(ei.value)
    # a leftover that has the form of a graph layer = a checkpoint with more layers than hp: refused by default
    for extra_name, shape in (('post_cbhg/highway_5/H/kernel', (128, 128)),
                              ('encoder_cbhg/conv_bank/conv_bank_17/conv1d/kernel', (17, 128, 128)),
                              ('prenet/prenet_dense_3/kernel', (128, 128))):
        more = dict(full)
        more[C.GRAPH_PREFIX + extra_name] = np.zeros(shape, np.float32)
        C.write_checkpoint(prefix, more)
        with pytest.raises(C.CheckpointError) as ei:
            C.load_tacotron_weights(prefix, hp)
        assert extra_name in str(ei.value) and 'more layers' in str(ei.value)
        assert set(C.load_tacotron_weights(prefix, hp, allow_extra_layers=True)) == set(w)
    capsys.readouterr()
    # a surplus batch-norm scope (a 17th conv-bank BN) must not shift the order-based assignment silently
    more = dict(full)
    for part in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
        more[C.GRAPH_PREFIX + 'encoder_cbhg/conv_bank/batch_normalization_16/' + part] = np.ones(128, np.float32)
    C.write_checkpoint(prefix, more)
    with pytest.raises(C.CheckpointError) as ei:
        C.load_tacotron_weights(prefix, hp)
    assert '17 batch_normalization scopes' in str(ei.value)
    # save side
    C.save_tacotron_weights(prefix, w)
    assert np.array_equal(C.load_tacotron_weights(prefix, hp)['dense/kernel'], w['dense/kernel'])


@pytest.mark.parametrize('spelling', ['tf1.15', 'residual_wrapper', 'global_bn'])
def test_variable_name_spellings(tmp_path, spelling, capsys):
    """the `residual_wrapper/` segment of cell_1 / cell_2 (model/decoder.py:28-32) is
    TF-version dependent and batch-norm scopes are auto-numbered; the loader matches by pattern and shape, so every
    spelling loads to the same weight set, and `verbose` prints the name mapping."""
    hp = HParams()
    w = init_weights(hp, seed=5, randomize_bn=True)
    prefix = str(tmp_path / 'model.ckpt-9')
    C.save_tacotron_weights(prefix, w, spelling=spelling)
    names, _ = C.read_index(prefix)
    if spelling == 'residual_wrapper':
        assert any('/cell_1/residual_wrapper/gru_cell/' in n for n in names)
        assert not any('/cell_1/gru_cell/' in n for n in names)
    if spelling == 'global_bn':
        assert C.GRAPH_PREFIX + 'post_cbhg/batch_normalization_27/gamma' in names
        assert C.GRAPH_PREFIX + 'post_cbhg/batch_normalization_1/gamma' not in names
    rep = {}
    back = C.load_tacotron_weights(prefix, hp, verbose=True, report=rep)
    assert set(back) == set(w) and not rep['leftovers']
    for k in w:
        assert np.array_equal(back[k], w[k]), k
    assert bool(rep['renamed']) == (spelling != 'tf1.15')
    out = capsys.readouterr().out
    assert 'dense/kernel <- model/inference/dense/kernel' in out
    if spelling != 'tf1.15':
        assert '(renamed)' in out


def test_mismatched_bn_count_and_shapes_are_named(tmp_path):
    hp = HParams()
    w = init_weights(hp, seed=5)
    prefix = str(tmp_path / 'm')
    full = {C.GRAPH_PREFIX + k: v for k, v in w.items()}
    for part in ('gamma', 'beta', 'moving_mean', 'moving_variance'):      # one conv-bank BN short
        del full[C.GRAPH_PREFIX + 'encoder_cbhg/conv_bank/batch_normalization_15/' + part]
    full[C.GRAPH_PREFIX + 'post_cbhg/batch_normalization/gamma'] = np.ones(7, np.float32)
    C.write_checkpoint(prefix, full)
    with pytest.raises(C.CheckpointError) as ei:
        C.load_tacotron_weights(prefix, hp)
    msg = str(ei.value)
    assert 'missing encoder_cbhg/conv_bank/batch_normalization_15/gamma' in msg
    assert 'shape post_cbhg/batch_normalization/gamma' in msg


def test_tensor_crc_checked_by_default(tmp_path):
    """A flipped bit in the data shard of a real-size weight set is caught by the default load (vectorised CRC32C
    agrees with the byte-wise definition on large inputs)."""
    rng = np.random.default_rng(1)
    blob = rng.integers(0, 256, 300001, dtype=np.uint8).tobytes()
    assert C.crc32c(blob) == C._crc_scalar(0xFFFFFFFF, blob) ^ 0xFFFFFFFF
    assert C.crc32c(blob[1000:], C.crc32c(blob[:1000])) == C.crc32c(blob)
    hp = HParams()
    w = init_weights(hp, seed=2)
    prefix = str(tmp_path / 'model.ckpt-3')
    C.save_tacotron_weights(prefix, w)
    path = prefix + '.data-00000-of-00001'
    d = bytearray(open(path, 'rb').read())
    d[len(d) // 2] ^= 4
    open(path, 'wb').write(bytes(d))
    with pytest.raises(C.CheckpointError, match='CRC32C'):
        C.load_tacotron_weights(prefix, hp)
    assert C.load_tacotron_weights(prefix, hp, verify_crc=False).keys() == w.keys()
