"""`Synthesizer.load / .synthesize` and the `--text` CLI (reference synthesizer.py:16-76).

Same call shapes and the same on-disk results (`<synth_dir>/wavs/synth-%03d.wav`,
`text/text-%03d.txt`, `index.txt` lines `txt_path|wav_path|text`). The session.run of
synthesizer.py:39-43 becomes one taco_synthesize call on the GPU; text -> ids before and
de-emphasis / endpointing / wav writing after stay on the host, as in the reference.

Checkpoints: `<checkpoint_dir>-<step>.index` + `.data-*` as tf.train.Saver writes them (tf_checkpoint.py), or
`<checkpoint_dir>-<step>.npz` in this package's flat format (weights.save_npz).

Extensions: `synthesize_batch` (texts as one batch), `synthesize_long` (a long text split at
sentence ends, the pieces synthesised as one batch and joined), `synthesize_many` (a stream of texts through
`SynthesisPipeline`, several batches in flight), `plot=True` for the alignment PNG next to the wav."""
import argparse
import os
import re

import numpy as np

from .data.batch import Batch, pad_inputs
from .hparams import hparams
from .model.tacotron import Tacotron
from .text.text_tools import text_to_onehot
from .tools import audio
from .weights import init_weights, load_npz


class Synthesizer:
    def load(self, checkpoint_dir, restore_step, model_name='tacotron', device=0):
        print('Constructing model: %s' % model_name)
        restore_dir = '%s-%d' % (checkpoint_dir, restore_step)
        print('Loading checkpoint: %s' % restore_dir)
        # a checkpoint the reference's tf.train.Saver wrote (model/tacotron.py:169: <dir>/model.ckpt-<step>.index +
        # .data-*), or this package's flat .npz form
        if os.path.exists(restore_dir + '.index'):
            from .tf_checkpoint import load_tacotron_weights
            weights = load_tacotron_weights(restore_dir, hparams)
        elif os.path.exists(restore_dir + '.npz'):
            weights = load_npz(restore_dir + '.npz')
        else:
            raise FileNotFoundError('neither %s.index (TF checkpoint) nor %s.npz found' % (restore_dir, restore_dir))
        self._weights, self._device = weights, device
        self.model = Tacotron(hparams=hparams, weights=weights, device=device)
        audio.set_default_engine(self.model.engine)
        return self

    def load_random(self, seed=1234, device=0):
        """Random-init weights (what synthesizer.py:30 leaves in the variables before restore)."""
        self._weights, self._device = init_weights(hparams, seed=seed), device
        self.model = Tacotron(hparams=hparams, weights=self._weights, device=device)
        return self

    def synthesize_batch(self, texts):
        """Extension: a list of texts in one batch -> list of float64 waveforms (de-emphasised, trimmed)."""
        seqs = [np.asarray(text_to_onehot(t, 'basic_cleaners'), np.int32) for t in texts]
        ids, lens = pad_inputs(seqs)
        self.model.initialize(Batch((ids, lens, None, None), prep=False), to_wav=True)
        wavs = self.model.wav_outputs.cpu().numpy()
        out = []
        for w in wavs:
            w = audio.pre_emphasis_inv(w)
            out.append(w[:audio.find_endpoint(w)])
        return out

    def synthesize_long(self, text, max_chars=150, pause_s=0.15):
        """Extension: a text longer than the decoder's reach (max_iters * r frames ~ 12.5 s) is split at sentence
        ends (then at commas / spaces if a sentence is still longer than max_chars), all pieces run as ONE batch,
        and the trimmed waveforms are joined with `pause_s` of silence."""
        pieces = split_text(text, max_chars)
        wavs = self.synthesize_batch(pieces)
        gap = np.zeros(int(pause_s * hparams.sample_rate))
        out = []
        for i, w in enumerate(wavs):
            if i:
                out.append(gap)
            out.append(w)
        return np.concatenate(out) if out else np.zeros(0)

    def synthesize_many(self, texts, batch_size=32, depth=4, devices=None):
        """Extension: an iterable of texts -> generator of (text, waveform) in input order, batches of `batch_size`
        with up to `depth` batches in flight on this GPU (pipeline.SynthesisPipeline).
        `devices` (a list of GPU ids): the texts are taken as one job, length-sorted and sharded over those GPUs
        (SynthesisPipeline.synthesize_sharded: one engine set per device, no collective), each batch padded to its
        own longest text; results still come back in input order."""
        from .pipeline import SynthesisPipeline
        if devices is not None:
            texts = list(texts)
            key = ('sharded', tuple(devices), depth)
            if getattr(self, '_pipeline_key', None) != key:
                if getattr(self, '_pipeline', None) is not None:
                    self._pipeline.close()
                self._pipeline = SynthesisPipeline(self._weights, hparams, depth=depth, devices=devices)
                self._pipeline_key = key
            ids, lens = self._ids(texts)
            res = self._pipeline.synthesize_sharded(ids, lens, batch_size=batch_size, to_wav=True, keys=('wav',),
                                                    trim=True)
            for t, r in zip(texts, res):
                w = audio.pre_emphasis_inv(r['wav'])
                yield t, w[:audio.find_endpoint(w)]
            return
        if getattr(self, '_pipeline', None) is None or getattr(self, '_pipeline_key', None) != ('stream', depth):
            if getattr(self, '_pipeline', None) is not None:
                self._pipeline.close()
            self._pipeline = SynthesisPipeline(self._weights, hparams, device=self._device, depth=depth)
            self._pipeline_key = ('stream', depth)
        groups = []

        def batches():
            group = []
            for t in texts:
                group.append(t)
                if len(group) == batch_size:
                    groups.append(group)
                    yield self._ids(group)
                    group = []
            if group:
                groups.append(group)
                yield self._ids(group)

        for o in self._pipeline.run(batches(), to_wav=True):
            group = groups.pop(0)
            for t, w in zip(group, o['wav'].numpy()):
                w = audio.pre_emphasis_inv(w)
                yield t, w[:audio.find_endpoint(w)]

    @staticmethod
    def _ids(texts):
        return pad_inputs([np.asarray(text_to_onehot(t, 'basic_cleaners'), np.int32) for t in texts])

    def synthesize(self, text, synth_dir, plot=False):
        wav = self.synthesize_batch([text])[0]
        os.makedirs(os.path.join(synth_dir, 'wavs'), exist_ok=True)
        os.makedirs(os.path.join(synth_dir, 'text'), exist_ok=True)
        index_path = os.path.join(synth_dir, 'index.txt')
        with open(index_path, 'a+') as index_file:
            index_file.seek(0)
            index = sum(1 for _ in index_file) + 1
            wav_path = os.path.join(synth_dir, 'wavs', 'synth-%03d.wav' % index)
            txt_path = os.path.join(synth_dir, 'text', 'text-%03d.txt' % index)
            index_file.write(txt_path + '|' + wav_path + '|' + text + '\n')
        with open(txt_path, 'w') as txt_file:
            txt_file.write(text)
        audio.save_wav(wav, wav_path)
        if plot:
            from .tools.plot import plot_alignment
            os.makedirs(os.path.join(synth_dir, 'alignments'), exist_ok=True)
            plot_alignment(self.model.alignments[0], os.path.join(synth_dir, 'alignments', 'align-%03d.png' % index),
                           info='%s, %d steps' % (text[:60], self.model.steps_run))
        print('Sentence has been synthesized and is available at: ', wav_path)
        return wav_path


def split_text(text, max_chars=150):
    """Sentence-end split ('.', '!', '?' followed by space), then ', ' and finally spaces for pieces that are still
    longer than max_chars. Pieces keep their punctuation; empty pieces are dropped."""
    pieces = []
    for sent in re.split(r'(?<=[.!?])\s+', text.strip()):
        while len(sent) > max_chars:
            cut = sent.rfind(', ', 0, max_chars)
            cut = cut + 1 if cut > 0 else sent.rfind(' ', 0, max_chars)
            if cut <= 0:
                cut = max_chars
            pieces.append(sent[:cut].strip())
            sent = sent[cut:].strip()
        if sent:
            pieces.append(sent)
    return pieces


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--restore_step', help='Select a valid checkpoint to synthesize from', default=0)
    parser.add_argument('--input_dir', help='base training output directory')
    parser.add_argument('--model_name', help='Select the name of the model', default='tacotron')
    parser.add_argument('--text', help='Text to be synthesized')
    parser.add_argument('--hparams', default='', help='comma separated name=value overrides (e.g. max_iters=400)')
    parser.add_argument('--plot', action='store_true', help='also save the alignment plot (tools/plot.py)')
    parser.add_argument('--random_init', action='store_true', help='use random-init weights instead of a checkpoint')
    args = parser.parse_args(argv)
    hparams.parse(args.hparams)
    in_dir = os.path.join(args.input_dir, args.model_name)
    synth_dir = os.path.join(in_dir, 'synthesized')
    os.makedirs(synth_dir, exist_ok=True)
    s = Synthesizer()
    if args.random_init:
        s.load_random()
    else:
        s.load(os.path.join(in_dir, 'meta', 'model.ckpt'), int(args.restore_step), model_name=args.model_name)
    s.synthesize(args.text, synth_dir, plot=args.plot)


if __name__ == '__main__':
    main()
