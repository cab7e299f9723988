"""`_process_utterance` of the reference's preprocessing (data/data_load.py:114-146) on the GPU analysis path:
one waveform file -> `<prefix>-spec-%05d.npy` [T, num_freq] and `<prefix>-mel-%05d.npy` [T, num_mels], float32,
np.save(allow_pickle=False) - the on-disk format the reference's DataFeeder reads. Both spectrograms come from one
taco_spectrogram call (one STFT) instead of the reference's two. Dataset walking / metadata stay out of scope."""
import os

import numpy as np

from ..tools import audio


def process_utterance(out_dir, index, wav_path, text, prefix='data', trim_silence=False, engine=None, **kwargs):
    """Returns (spect_filename, melspect_filename, n_frames, text) + the extra keyword values, as the reference."""
    wav = audio.load_wav(wav_path)
    if trim_silence:
        wav = audio.trim_silence(wav)
    eng = engine or audio.default_engine()
    lin, mel = eng.spectrogram(np.asarray(wav, np.float32)[None])
    spect = lin[0].cpu().numpy().astype(np.float32)          # [T, num_freq]: what the reference stores (spect.T)
    melspect = mel[0].cpu().numpy().astype(np.float32)
    n_frames = spect.shape[0]
    spect_filename = prefix + '-spec-%05d.npy' % index
    melspect_filename = prefix + '-mel-%05d.npy' % index
    np.save(os.path.join(out_dir, spect_filename), spect, allow_pickle=False)
    np.save(os.path.join(out_dir, melspect_filename), melspect, allow_pickle=False)
    return (spect_filename, melspect_filename, n_frames, text) + tuple(kwargs.values())


_process_utterance = process_utterance
