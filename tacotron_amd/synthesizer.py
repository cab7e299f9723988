"""`Synthesizer.load / .synthesize` and the `--text` CLI (reference synthesizer.py:16-76).

Same call shapes and the same on-disk results (`<synth_dir>/wavs/synth-%03d.wav`,
`text/text-%03d.txt`, `index.txt` lines `txt_path|wav_path|text`). The session.run of
synthesizer.py:39-43 becomes one taco_synthesize call on the GPU; text -> ids before and
de-emphasis / endpointing / wav writing after stay on the host, as in the reference.

Checkpoints: `<checkpoint_dir>-<step>.npz` in this package's flat format (weights.save_npz).
Reading TF tensor-bundle checkpoints is not implemented yet."""
import argparse
import os

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
        self.model = Tacotron(hparams=hparams, weights=weights, device=device)
        return self

    def load_random(self, seed=1234, device=0):
        """Random-init weights (what synthesizer.py:30 leaves in the variables before restore)."""
        self.model = Tacotron(hparams=hparams, weights=init_weights(hparams, seed=seed), device=device)
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

    def synthesize(self, text, synth_dir):
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
        print('Sentence has been synthesized and is available at: ', wav_path)
        return wav_path


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--restore_step', help='Select a valid checkpoint to synthesize from', default=0)
    parser.add_argument('--input_dir', help='base training output directory')
    parser.add_argument('--model_name', help='Select the name of the model', default='tacotron')
    parser.add_argument('--text', help='Text to be synthesized')
    parser.add_argument('--hparams', default='', help='comma separated name=value overrides (e.g. max_iters=400)')
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
    s.synthesize(args.text, synth_dir)


if __name__ == '__main__':
    main()
