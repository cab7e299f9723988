"""Hyper-parameters of the synthesis path.

Mirrors the reference's single global `hparams` object (reference hparams.py:7-68), which is a
`tf.contrib.training.HParams`; here it is a plain object with the same three access styles the
reference uses: `hparams.get('name')` (encoder.py:53-61, decoder.py:19-54, audio.py:161-163),
attribute access (tacotron.py:43-46) and `hparams.parse("k=v,k2=v2")` (runner.py:72).
Only values that shape the inference path are kept; training values are carried for
completeness of `get()` but nothing here reads them.
"""
import ast
import copy

_DEFAULTS = dict(
    cleaners='basic_cleaners',
    # Audio (hparams.py:13-21)
    num_mels=80,
    num_freq=1025,
    n_fft=2048,
    sample_rate=24000,
    frame_length_ms=50,
    frame_shift_ms=12.5,
    preemphasis=0.97,
    min_level_db=-100,
    ref_level_db=20,
    # Model (hparams.py:25-51)
    embedded_depth=256,
    outputs_per_step=5,
    prenet_depths=[256, 128],
    encoder_K=16,
    encoder_bank_num_filters=128,
    encoder_pooling_stride=1,
    encoder_pooling_width=2,
    encoder_proj_num_filters=[128, 128],
    encoder_proj_filter_width=3,
    encoder_num_highway_layers=4,
    encoder_highway_depth=128,
    encoder_gru_num_cells=128,
    decoder_K=8,
    decoder_bank_num_filters=128,
    decoder_pooling_stride=1,
    decoder_pooling_width=2,
    decoder_proj_num_filters=[256, 80],
    decoder_proj_filter_width=3,
    decoder_num_highway_layers=4,
    decoder_highway_depth=128,
    decoder_gru_num_cells=128,
    postnet_depth=256,
    attention_depth=256,
    decoder_depth=256,
    pad_value=0,
    # Training values (hparams.py:57-62), unused on the synthesis path
    batch_size=32,
    superbatch_size=32,
    adam_beta1=0.9,
    adam_beta2=0.999,
    initial_learning_rate=0.002,
    decay_learning_rate=True,
    # Eval (hparams.py:65-67)
    max_iters=200,
    griffin_lim_iters=60,
    power=1.5,
)


class HParams:
    def __init__(self, **kw):
        object.__setattr__(self, '_values', copy.deepcopy(_DEFAULTS))
        for k, v in kw.items():
            self.set_hparam(k, v)

    def get(self, name, default=None):
        return self._values.get(name, default)

    def __getattr__(self, name):
        try:
            return object.__getattribute__(self, '_values')[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self.set_hparam(name, value)

    def set_hparam(self, name, value):
        if name not in self._values:
            raise ValueError('Unknown hyperparameter: %s' % name)
        old = self._values[name]
        if isinstance(old, bool):
            value = bool(value)
        elif isinstance(old, int) and not isinstance(value, bool):
            if float(value) != int(value):
                raise ValueError('%s expects an integer, got %r' % (name, value))
            value = int(value)
        elif isinstance(old, float):
            value = float(value)
        elif isinstance(old, list):
            value = [type(old[0])(x) for x in value]
        self._values[name] = value

    def parse(self, values):
        """`name=value,name2=[a,b]` overrides, as `HParams.parse` accepts (runner.py:72)."""
        if not values:
            return self
        depth, cur, parts = 0, '', []
        for ch in values:
            if ch == '[':
                depth += 1
            elif ch == ']':
                depth -= 1
            if ch == ',' and depth == 0:
                parts.append(cur)
                cur = ''
            else:
                cur += ch
        if cur.strip():
            parts.append(cur)
        for p in parts:
            if '=' not in p:
                raise ValueError('Could not parse hparam override: %r' % p)
            k, v = p.split('=', 1)
            k, v = k.strip(), v.strip()
            try:
                val = ast.literal_eval(v)
            except (ValueError, SyntaxError):
                val = v
            if isinstance(val, tuple):
                val = list(val)
            self.set_hparam(k, val)
        return self

    def values(self):
        return copy.deepcopy(self._values)

    def copy(self):
        return HParams(**self._values)

    # Derived STFT parameters (audio.py:150-164)
    def stft_params(self):
        n_fft = self.n_fft
        hop = int(self.frame_shift_ms / 1000 * self.sample_rate)
        win = int(self.frame_length_ms / 1000 * self.sample_rate)
        return n_fft, hop, win


hparams = HParams()
