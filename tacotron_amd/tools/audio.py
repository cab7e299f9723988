"""Audio tools of the synthesis path (reference tools/audio.py).

Device side: `spectrogram_tensorflow_inv` (audio.py:57-64) runs the HIP Griffin-Lim
(taco_griffin_lim). Host side, as in the reference (synthesizer.py:44-57): `pre_emphasis_inv`
(audio.py:111-116), `find_endpoint` (audio.py:81-88), `save_wav` (audio.py:15-17).
The analysis half of the reference module (load_wav, spectrogram, mel_spectrogram, ...) belongs
to preprocessing and is out of scope."""
import wave

import numpy as np

from ..hparams import hparams


def spectrogram_tensorflow_inv(spect, engine, iters=None):
    """[T, 1025] or [B, T, 1025] normalised linear spectrogram -> waveform(s) on the GPU.
    Does NOT invert the pre-emphasis (the caller does, as synthesizer.py:44)."""
    return engine.griffin_lim(spect, iters)


inv_spectrogram = spectrogram_tensorflow_inv


def _db_to_amp(x):
    return np.power(10.0, x * 0.05)


def pre_emphasis_inv(x, coef=None):
    """scipy.signal.lfilter([1], [1, -preemphasis], x): y[n] = x[n] + a*y[n-1] (float64)."""
    a = float(hparams.get('preemphasis')) if coef is None else float(coef)
    x = np.asarray(x, np.float64)
    try:
        from scipy import signal
        return signal.lfilter([1], [1, -a], x)
    except ImportError:  # pragma: no cover
        y = np.empty_like(x)
        acc = 0.0
        for n in range(len(x)):
            acc = x[n] + a * acc
            y[n] = acc
        return y


def find_endpoint(wav, threshold_db=-40, min_silence_sec=0.8):
    window_length = int(hparams.get('sample_rate') * min_silence_sec)
    hop_length = int(window_length / 4)
    threshold = _db_to_amp(threshold_db)
    for x in range(hop_length, len(wav) - window_length, hop_length):
        if np.max(wav[x:x + window_length]) < threshold:   # signed max, as the reference
            return x + hop_length
    return len(wav)


def wav_to_int16(wav):
    wav = np.asarray(wav, np.float64)
    peak = np.max(np.abs(wav)) if len(wav) else 0.0
    return (wav * (32767 / max(0.01, peak))).astype(np.int16)


def save_wav(wav, path):
    """Scale to int16 full range and write 16-bit PCM mono at hparams.sample_rate (audio.py:15-17;
    librosa.output.write_wav wrote the same PCM via scipy.io.wavfile)."""
    pcm = wav_to_int16(wav)
    with wave.open(path, 'wb') as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(hparams.get('sample_rate')))
        f.writeframes(pcm.astype('<i2').tobytes())
