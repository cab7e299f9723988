"""Weight container for the synthesis path.

The reference keeps its parameters in TF variables restored by name from `model.ckpt-<step>`
(synthesizer.py:32-34); without a checkpoint they hold TF's default initial values
(synthesizer.py:30 runs `global_variables_initializer`). This module owns

* the list of variables the inference graph creates (names as TF would scope them under
  `model/inference/`) with their shapes as a function of `hparams`,
* `init_weights()` - "random-init weights" drawn with the TF default initialisers
  (glorot-uniform kernels, zero biases, GRU gate bias 1.0, highway T bias -1.0
  (modules.py:121-122), truncated-normal(0.5) embedding (batch.py:37-39), BN gamma=1 beta=0
  mean=0 var=1) from a seeded numpy generator,
* save/load of the flat `.npz` form used for fixtures.

A weight set is a plain `dict[str, np.ndarray(float32)]`; nothing in here touches the GPU.
"""
import numpy as np

from .hparams import hparams as _default_hparams
from .text.characters import chars

PREFIX = 'model/inference/'
_DEC = 'decoder/output_projection_wrapper/'
_CELL0 = _DEC + 'multi_rnn_cell/cell_0/output_projection_wrapper/'
_ATTW = _CELL0 + 'concat_output_and_attention_wrapper/attention_wrapper/'


def _bn_name(scope, idx):
    return scope + 'batch_normalization' + ('' if idx == 0 else '_%d' % idx)


def _cbhg_specs(scope, in_depth, K, bank_filters, proj_filters, proj_width, n_highway, highway_depth, gru_cells):
    specs = []
    for k in range(1, K + 1):
        s = '%s/conv_bank/conv_bank_%d/conv1d/' % (scope, k)
        specs.append((s + 'kernel', (k, in_depth, bank_filters), 'conv'))
        specs.append((s + 'bias', (bank_filters,), 'zeros'))
        specs += _bn_specs(_bn_name(scope + '/conv_bank/', k - 1), bank_filters)
    c_in = K * bank_filters
    for i, c_out in enumerate(proj_filters):
        s = '%s/proj_%d/conv1d/' % (scope, i + 1)
        specs.append((s + 'kernel', (proj_width, c_in, c_out), 'conv'))
        specs.append((s + 'bias', (c_out,), 'zeros'))
        specs += _bn_specs(_bn_name(scope + '/', i), c_out)
        c_in = c_out
    if c_in != highway_depth:  # modules.py:83-84
        specs.append((scope + '/dense/kernel', (c_in, highway_depth), 'glorot'))
        specs.append((scope + '/dense/bias', (highway_depth,), 'zeros'))
    for i in range(1, n_highway + 1):
        for g, binit in (('H', 'zeros'), ('T', 'minus_one')):
            s = '%s/highway_%d/%s/' % (scope, i, g)
            specs.append((s + 'kernel', (highway_depth, highway_depth), 'glorot'))
            specs.append((s + 'bias', (highway_depth,), binit))
    for d in ('fw', 'bw'):
        specs += _gru_specs('%s/bidirectional_rnn/%s/gru_cell/' % (scope, d), highway_depth, gru_cells)
    return specs


def _bn_specs(scope, c):
    return [(scope + '/gamma', (c,), 'ones'), (scope + '/beta', (c,), 'zeros'),
            (scope + '/moving_mean', (c,), 'zeros'), (scope + '/moving_variance', (c,), 'ones')]


def _gru_specs(scope, in_depth, cells):
    return [(scope + 'gates/kernel', (in_depth + cells, 2 * cells), 'glorot'),
            (scope + 'gates/bias', (2 * cells,), 'ones'),
            (scope + 'candidate/kernel', (in_depth + cells, cells), 'glorot'),
            (scope + 'candidate/bias', (cells,), 'zeros')]


def variable_specs(hp=None):
    """[(name, shape, initialiser)] for every variable of the inference graph, in creation order."""
    hp = hp or _default_hparams
    E = hp.embedded_depth
    pre = list(hp.prenet_depths)
    A = hp.attention_depth
    D = hp.decoder_depth
    M = hp.num_mels
    r = hp.outputs_per_step
    specs = [('embedding', (len(chars), E), 'embedding')]
    d = E
    for i, n in enumerate(pre):
        specs.append(('prenet/prenet_dense_%d/kernel' % (i + 1), (d, n), 'glorot'))
        specs.append(('prenet/prenet_dense_%d/bias' % (i + 1), (n,), 'zeros'))
        d = n
    specs += _cbhg_specs('encoder_cbhg', d, hp.encoder_K, hp.encoder_bank_num_filters,
                         hp.encoder_proj_num_filters, hp.encoder_proj_filter_width,
                         hp.encoder_num_highway_layers, hp.encoder_highway_depth, hp.encoder_gru_num_cells)
    enc_out = 2 * hp.encoder_gru_num_cells
    specs.append(('memory_layer/kernel', (enc_out, A), 'glorot'))
    # decoder cell stack (decoder.py:18-36)
    d = M + enc_out  # AttentionWrapper feeds concat([inputs, attention]) to the cell
    for i, n in enumerate(pre):
        s = _ATTW + 'decoder_prenet_wrapper/decoder_prenet/prenet_dense_%d/' % (i + 1)
        specs.append((s + 'kernel', (d, n), 'glorot'))
        specs.append((s + 'bias', (n,), 'zeros'))
        d = n
    specs += _gru_specs(_ATTW + 'decoder_prenet_wrapper/gru_cell/', d, A)
    specs.append((_ATTW + 'bahdanau_attention/query_layer/kernel', (A, A), 'glorot'))
    specs.append((_ATTW + 'bahdanau_attention/attention_v', (A,), 'glorot'))
    specs.append((_CELL0 + 'kernel', (A + enc_out, D), 'glorot'))
    specs.append((_CELL0 + 'bias', (D,), 'zeros'))
    for c in (1, 2):
        specs += _gru_specs(_DEC + 'multi_rnn_cell/cell_%d/gru_cell/' % c, D, D)
    specs.append((_DEC + 'kernel', (D, M * r), 'glorot'))
    specs.append((_DEC + 'bias', (M * r,), 'zeros'))
    specs += _cbhg_specs('post_cbhg', M, hp.decoder_K, hp.decoder_bank_num_filters,
                         hp.decoder_proj_num_filters, hp.decoder_proj_filter_width,
                         hp.decoder_num_highway_layers, hp.decoder_highway_depth, hp.decoder_gru_num_cells)
    specs.append(('dense/kernel', (2 * hp.decoder_gru_num_cells, hp.num_freq), 'glorot'))
    specs.append(('dense/bias', (hp.num_freq,), 'zeros'))
    return specs


def _draw(rng, shape, kind):
    if kind in ('glorot', 'conv'):
        if len(shape) == 1:
            fan_in = fan_out = shape[0]
        elif len(shape) == 2:
            fan_in, fan_out = shape
        else:
            rf = int(np.prod(shape[:-2]))
            fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)
    if kind == 'zeros':
        return np.zeros(shape, np.float32)
    if kind == 'ones':
        return np.ones(shape, np.float32)
    if kind == 'minus_one':
        return np.full(shape, -1.0, np.float32)
    if kind == 'embedding':  # truncated normal, stddev 0.5, resample beyond 2 sigma
        x = rng.standard_normal(size=shape)
        bad = np.abs(x) > 2.0
        while bad.any():
            x[bad] = rng.standard_normal(size=int(bad.sum()))
            bad = np.abs(x) > 2.0
        return (0.5 * x).astype(np.float32)
    raise ValueError(kind)


def init_weights(hp=None, seed=1234, randomize_bn=False, randomize_bias=False, attention_sharpness=1.0):
    """Random-init weight set. `randomize_bn` / `randomize_bias` replace the trivial BN
    statistics / zero biases by random ones (test fixtures that exercise the BN fold and the
    bias adds); `attention_sharpness` scales `attention_v` to give a peaked attention."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape, kind in variable_specs(hp):
        w[name] = _draw(rng, shape, kind)
    if randomize_bias:
        for name in w:
            if name.endswith('/bias'):
                w[name] = (w[name] + rng.uniform(-0.1, 0.1, size=w[name].shape)).astype(np.float32)
    if randomize_bn:
        for name in w:
            if name.endswith('/gamma'):
                w[name] = rng.uniform(0.5, 1.5, size=w[name].shape).astype(np.float32)
            elif name.endswith('/beta') or name.endswith('/moving_mean'):
                w[name] = rng.uniform(-0.2, 0.2, size=w[name].shape).astype(np.float32)
            elif name.endswith('/moving_variance'):
                w[name] = rng.uniform(0.5, 2.0, size=w[name].shape).astype(np.float32)
    if attention_sharpness != 1.0:
        k = _ATTW + 'bahdanau_attention/attention_v'
        w[k] = (w[k] * attention_sharpness).astype(np.float32)
    return w


def check_weights(w, hp=None):
    """Raise with a full list of missing / mis-shaped variables (loaders call this)."""
    problems = []
    for name, shape, _ in variable_specs(hp):
        if name not in w:
            problems.append('missing %s %s' % (name, shape))
        elif tuple(w[name].shape) != tuple(shape):
            problems.append('shape %s: got %s want %s' % (name, tuple(w[name].shape), shape))
    if problems:
        raise ValueError('weight set does not match hparams:\n  ' + '\n  '.join(problems))


def num_params(w):
    return int(sum(int(v.size) for v in w.values()))


def save_npz(path, w):
    np.savez(path, **{k.replace('/', '|'): v for k, v in w.items()})


def load_npz(path):
    with np.load(path, allow_pickle=False) as z:
        return {k.replace('|', '/'): np.asarray(z[k], np.float32) for k in z.files}
