"""Audio tools of the synthesis path (reference tools/audio.py).

Every function keeps the reference's signature; the optional `engine` argument (default: the engine registered by
`Synthesizer.load`, or one created on first use) names the GPU handle to run on.
Device side: `spectrogram_tensorflow_inv` (audio.py:57-64) runs the HIP Griffin-Lim
(taco_griffin_lim). Host side, as in the reference (synthesizer.py:44-57): `pre_emphasis_inv`
(audio.py:111-116), `find_endpoint` (audio.py:81-88), `save_wav` (audio.py:15-17). `spectrogram_inv` (audio.py:44-55,
the numpy/librosa Griffin-Lim train.py uses for evaluation audio) runs taco_griffin_lim_librosa.
Analysis half (audio.py:23-42,66-79: what data/data_load.py:114-146 stores as `*-spec-*.npy` / `*-mel-*.npy`):
`spectrogram` / `mel_spectrogram` run the centred STFT, dB scaling and mel projection on the GPU
(taco_spectrogram); `build_mel_basis` restates librosa.filters.mel (librosa 0.5.1: Slaney scale, area-normalised)
on the host - librosa is not installed here, so that restatement is unpinned. `load_wav` reads PCM WAV with the
standard library, `trim_silence` restates librosa.effects.trim (unpinned)."""
import wave

import numpy as np

from ..hparams import hparams


_default_engine = None


def set_default_engine(engine):
    """The engine the functions below use when none is passed, so that they can be called with the reference's own
    signatures (`audio.spectrogram(y)`, `audio.spectrogram_inv(spect)`); `Synthesizer.load` registers its model's."""
    global _default_engine
    _default_engine = engine


def default_engine():
    """The registered engine, or one created on first use: the audio kernels need no trained weights, only a handle
    (hparams + the derived mel filterbank), so randomly initialised ones do."""
    global _default_engine
    if _default_engine is None or getattr(_default_engine, 'handle', None) is None:     # none yet, or closed
        from ..engine import TacoEngine
        from ..weights import init_weights
        _default_engine = TacoEngine(init_weights(hparams, seed=0), hparams)
    return _default_engine


def spectrogram_tensorflow_inv(spect, engine=None, iters=None):
    """[T, 1025] or [B, T, 1025] normalised linear spectrogram -> waveform(s) on the GPU.
    Does NOT invert the pre-emphasis (the caller does, as synthesizer.py:44)."""
    return (engine or default_engine()).griffin_lim(spect, iters)


inv_spectrogram = spectrogram_tensorflow_inv


def spectrogram_inv(spect, engine=None, iters=None, rng=None):
    """audio.py:44-55: [num_freq, T] normalised linear spectrogram (the reference passes `spectrogram.T`) -> waveform
    [hop * (T - 1)] float64, numpy/librosa-flavoured Griffin-Lim on the GPU (taco_griffin_lim_librosa) and the
    de-emphasis on the host. The initial phases are np.random.rand(*spect.shape) as at audio.py:175 (global numpy
    state, so np.random.seed reproduces the reference's draw), or rng.random(spect.shape) when a Generator is given."""
    spect = np.asarray(spect)
    u = np.random.rand(*spect.shape) if rng is None else rng.random(spect.shape)
    y = (engine or default_engine()).griffin_lim_librosa(np.ascontiguousarray(spect.T, np.float32), np.ascontiguousarray(u.T, np.float32),
                                   iters)
    return pre_emphasis_inv(y.cpu().numpy())


def _db_to_amp(x):
    return np.power(10.0, x * 0.05)


def _amp_to_db(x):
    """audio.py:212-217."""
    return 20 * np.log10(np.maximum(1e-5, x))


def _normalize(S):
    """audio.py:254-264."""
    return np.clip((S - hparams.get('min_level_db')) / -float(hparams.get('min_level_db')), 0, 1)


def _normalize_inv(S):
    """audio.py:266-274."""
    return (np.clip(S, 0, 1) * -float(hparams.get('min_level_db'))) + hparams.get('min_level_db')


def _stft_params():
    """audio.py:150-164: (n_fft, hop_length, win_length)."""
    return hparams.stft_params()


def pre_emphasis(x, coef=None):
    """audio.py:91-109: scipy.signal.lfilter([1, -preemphasis], [1], x): y[n] = x[n] - a*x[n-1] (float64)."""
    from scipy import signal
    a = float(hparams.get('preemphasis')) if coef is None else float(coef)
    return signal.lfilter([1, -a], [1], x)


def _hz_to_mel(f):
    """librosa.hz_to_mel(htk=False): linear below 1 kHz (200/3 Hz per mel), logarithmic above (Slaney)."""
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def build_mel_basis(hp=None):
    """audio.py:246-252: librosa.filters.mel(sample_rate, n_fft, n_mels=num_mels) -> [num_mels, 1 + n_fft/2]
    (fmin 0, fmax sr/2, htk=False, norm=1: triangles on the Slaney mel scale, each scaled by 2 / its bandwidth)."""
    hp = hp or hparams
    sr, n_fft, n_mels = float(hp.sample_rate), int(hp.n_fft), int(hp.num_mels)
    fftfreqs = np.linspace(0.0, sr / 2, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(np.float32)


def load_wav(path):
    """audio.py:8-13 (librosa.core.load(path, sr=hparams.sample_rate)[0]): mono float32 waveform in [-1, 1] at the
    model's sample rate. PCM WAV files are read with the standard library; a file at another rate is resampled with
    scipy's polyphase filter, which is not librosa's resampler (unpinned; files at the model's rate are exact)."""
    with wave.open(path, 'rb') as f:
        ch, width, rate, n = f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()
        raw = f.readframes(n)
    if width == 2:
        y = np.frombuffer(raw, '<i2').astype(np.float32) / 32768.0
    elif width == 4:
        y = np.frombuffer(raw, '<i4').astype(np.float32) / 2147483648.0
    elif width == 1:
        y = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError('%s: unsupported sample width %d bytes' % (path, width))
    if ch > 1:
        y = y.reshape(-1, ch).mean(axis=1)                      # librosa's to_mono
    sr = int(hparams.get('sample_rate'))
    if rate != sr:
        from math import gcd
        from scipy import signal
        g = gcd(sr, rate)
        y = signal.resample_poly(y, sr // g, rate // g).astype(np.float32)
    return y


def trim_silence(audio_signal, top_db=60, frame_length=2048, hop_length=512):
    """audio.py:19-21 (librosa.effects.trim, librosa 0.5.1 defaults, restated: unpinned): drop leading and trailing
    frames whose mean square is more than top_db below the loudest frame's."""
    y = np.asarray(audio_signal)
    if len(y) < frame_length:
        return y
    nfr = 1 + (len(y) - frame_length) // hop_length
    idx = hop_length * np.arange(nfr)[:, None] + np.arange(frame_length)[None, :]
    mse = np.mean(np.square(y[idx].astype(np.float64)), axis=1)
    db = 10.0 * np.log10(np.maximum(1e-10, mse)) - 10.0 * np.log10(max(1e-10, float(mse.max())))
    nz = np.flatnonzero(db > -top_db)
    if nz.size == 0:
        return y[:0]
    return y[int(nz[0]) * hop_length:min(len(y), (int(nz[-1]) + 1) * hop_length)]


def spectrogram(y, engine=None):
    """audio.py:23-42: normalised linear-scale dB spectrogram [1 + n_fft/2, frames] of one waveform."""
    lin, _ = (engine or default_engine()).spectrogram(np.asarray(y, np.float32)[None], want_mel=False)
    return lin[0].T.cpu().numpy()


def mel_spectrogram(y, engine=None):
    """audio.py:66-79: normalised mel-scale dB spectrogram [num_mels, frames]."""
    _, mel = (engine or default_engine()).spectrogram(np.asarray(y, np.float32)[None], want_linear=False)
    return mel[0].T.cpu().numpy()


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
    """audio.py:16-17: `wav *= 32767 / max(0.01, max|wav|)` then the int16 cast, in the dtype of the array handed in
    (the reference scales in place: a float32 waveform is scaled in float32, the float64 output of pre_emphasis_inv in
    float64 - pinned by tests/golden/host_dsp_goldens.npz). The caller's array is left alone here."""
    wav = np.array(wav, copy=True)
    if not np.issubdtype(wav.dtype, np.floating):
        wav = wav.astype(np.float64)
    if wav.size:
        wav *= 32767 / max(0.01, np.max(np.abs(wav)))
    return wav.astype(np.int16)


def save_wav(wav, path):
    """Scale to int16 full range and write 16-bit PCM mono at hparams.sample_rate (audio.py:15-17;
    librosa.output.write_wav wrote the same PCM via scipy.io.wavfile)."""
    pcm = wav_to_int16(wav)
    with wave.open(path, 'wb') as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(hparams.get('sample_rate')))
        f.writeframes(pcm.astype('<i2').tobytes())
