"""`Tacotron(hparams).initialize(batch)` (reference model/tacotron.py:22-57), inference branch.

The reference builds a TF graph whose tensors are evaluated later by `session.run`; here there
is no graph: `initialize` runs the HIP engine on the given ids at once and leaves the results
in the attributes the reference exports (`inputs, input_lengths, mel_outputs [B,T,80],
linear_outputs [B,T,1025], alignments [B,T_in,steps]`) as device tensors. Weights come from
`load_weights` (a dict keyed by the TF variable names) - the stand-in for `Saver.restore`."""
import numpy as np

from ..data.batch import Batch
from ..engine import TacoEngine
from ..hparams import hparams as _hparams


class Tacotron:
    def __init__(self, hparams=_hparams, weights=None, device=0):
        self._hparams = hparams
        self._device = device
        self.engine = None
        if weights is not None:
            self.load_weights(weights)

    def load_weights(self, weights):
        if self.engine is not None:
            self.engine.close()
        self.engine = TacoEngine(weights, self._hparams, self._device)
        return self

    def initialize(self, batch, input_lengths=None, to_wav=False):
        """`batch` is a Batch (as the reference) or an ids array with `input_lengths` given."""
        if self.engine is None:
            raise RuntimeError('Tacotron.initialize: no weights loaded (call load_weights first)')
        if not isinstance(batch, Batch):
            batch = Batch((batch, input_lengths, None, None), prep=False)
        if batch._lin_targets is not None:
            raise NotImplementedError('training graph (linear targets given) is out of scope')
        inputs, lengths, _, _ = batch.get_all()
        inputs = np.asarray(inputs, np.int32)
        lengths = np.asarray(lengths, np.int32)
        if inputs.ndim != 2 or lengths.shape != (inputs.shape[0],):
            raise ValueError('inputs must be [B, T_in] and input_lengths [B]')
        if inputs.size and (inputs.min() < 0 or inputs.max() >= len(_chars())):
            raise ValueError('character id out of range [0, %d)' % len(_chars()))
        if lengths.min() < 1 or lengths.max() > inputs.shape[1]:
            raise ValueError('input_lengths must lie in [1, T_in]')
        o = self.engine.finish(self.engine.synthesize(inputs, lengths, to_wav=to_wav), to_wav=to_wav)
        self.inputs, self.input_lengths, self.mel_targets, self.linear_targets = batch.get_all()
        self.encoder_outputs = o['enc_out']
        self.mel_outputs = o['mel']
        self.linear_outputs = o['linear']
        self.alignments = o['alignments']
        self.wav_outputs = o.get('wav') if to_wav else None
        self.steps_run = o['steps_run']
        return self


def _chars():
    from ..text.characters import chars
    return chars
