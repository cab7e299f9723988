"""`Batch` as the synthesis path uses it (reference data/batch.py:10-47 with prep=False):
a holder of (inputs, input_lengths, mel_targets, linear_targets). The training-side padding
helpers of the reference (batch.py:61-170) are out of scope; `pad_inputs` keeps only the input
layout they define (right-pad with id 0, batch.py:121-134) for batched inference."""
import numpy as np

from ..hparams import hparams


class Batch:
    def __init__(self, data, prep=False):
        if prep:
            raise NotImplementedError('Batch(prep=True) prepares training targets; only the synthesis '
                                      'path (prep=False, synthesizer.py:22) is implemented')
        self._inputs, self._input_lengths, self._mel_targets, self._lin_targets = data

    def get_input_lengths(self):
        return self._input_lengths

    def get_size(self):
        return int(np.shape(self._inputs)[0])

    def get_embedds(self):
        """The reference returns embedding_lookup(embedding, inputs) (batch.py:36-40). Here the
        lookup is fused with the encoder prenet into one table gather inside taco_encode, so
        the ids themselves are what the encoder consumes."""
        return self._inputs

    def get_all(self):
        return (self._inputs, self._input_lengths, self._mel_targets, self._lin_targets)

    def get_inputs(self):
        return self._inputs

    def get_mel_targets(self):
        return self._mel_targets


def pad_inputs(seqs, pad_value=None):
    """List of id sequences -> ([B, T_max] int32 right-padded with pad_value, [B] int32 lengths)."""
    pad_value = hparams.pad_value if pad_value is None else pad_value
    lens = np.asarray([len(s) for s in seqs], np.int32)
    out = np.full((len(seqs), int(lens.max())), pad_value, np.int32)
    for i, s in enumerate(seqs):
        out[i, :len(s)] = s
    return out, lens
