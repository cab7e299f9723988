"""CPU ORACLE (test infrastructure, NOT product code) - numpy restatement of the
cadia-lvl/tacotron synthesis path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module. The product (`tacotron_amd/`) never does: its arithmetic is the HIP library.

PARITY UNPINNED for the numeric path: the reference has no tests, golden vectors or
checkpoints, and its arithmetic lives in un-vendored third-party packages that are absent
here (tensorflow==1.15.4 requirements.txt:33, librosa==0.5.1 requirements.txt:15), so this
file restates their published op semantics along the reference's own
call sites, cited per function as `file:line` into /root/reference. The text frontend IS
pinned: tests/golden/text_goldens.json was produced by importing the reference's text/
package (tests/golden/make_text_goldens.py).

Every function takes/returns numpy arrays; `dtype` selects float64 (parity oracle) or float32
(CPU timing baseline, closest to what TF-CPU would compute).
"""
import numpy as np

# ----------------------------------------------------------------------------------------
# elementary ops (TF 1.15 semantics)
# ----------------------------------------------------------------------------------------


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def dense(x, kernel, bias=None, act=None):
    """tf.layers.dense: act(x @ kernel[in,out] + bias) (modules.py:26,84,120-121, decoder.py:57)."""
    y = x @ kernel
    if bias is not None:
        y = y + bias
    if act == 'relu':
        y = np.maximum(y, 0)
    elif act == 'sigmoid':
        y = sigmoid(y)
    elif act is not None:
        raise ValueError(act)
    return y


def conv1d_same(x, kernel, bias):
    """tf.layers.conv1d(padding='same', stride 1) on [B,T,C_in] with kernel [K,C_in,C_out]
    (modules.py:138-139). Cross-correlation, zero padding pad_left=(K-1)//2, rest right."""
    B, T, C = x.shape
    K, Cin, Cout = kernel.shape
    assert C == Cin
    pl = (K - 1) // 2
    xp = np.zeros((B, T + K - 1, C), x.dtype)
    xp[:, pl:pl + T] = x
    cols = np.concatenate([xp[:, j:j + T] for j in range(K)], axis=-1)  # [B,T,K*C], tap-major
    return cols @ kernel.reshape(K * Cin, Cout) + bias


def batch_norm_inference(x, gamma, beta, mean, var, eps=1e-3):
    """tf.layers.batch_normalization(training=False), axis -1, epsilon 1e-3 (modules.py:140)."""
    inv = gamma / np.sqrt(var + eps)
    return x * inv + (beta - mean * inv)


def convolute(x, w, scope, bn_scope, act):
    """modules.py:125-140: conv1d -> activation (inside the conv layer) -> batch norm."""
    y = conv1d_same(x, w[scope + '/conv1d/kernel'], w[scope + '/conv1d/bias'])
    if act == 'relu':
        y = np.maximum(y, 0)
    return batch_norm_inference(y, w[bn_scope + '/gamma'], w[bn_scope + '/beta'],
                                w[bn_scope + '/moving_mean'], w[bn_scope + '/moving_variance'])


def maxpool2_same(x):
    """tf.layers.max_pooling1d(pool_size=2, strides=1, padding='same') (modules.py:70-71):
    out[t] = max(x[t], x[t+1]); the single right pad is ignored by the max."""
    y = x.copy()
    y[:, :-1] = np.maximum(x[:, :-1], x[:, 1:])
    return y


def maxpool_same(x, width, stride=1):
    """tf.layers.max_pooling1d(pool_size=width, strides=stride, padding='same') over time (modules.py:70-71): window
    [t - (width-1)//2, t + width-1 - (width-1)//2] clipped to the sequence (the padding never wins the max). Only
    stride 1 keeps the length the residual of modules.py:80 needs."""
    if stride != 1:
        raise ValueError('pooling stride %d: the CBHG residual (modules.py:80) needs stride 1' % stride)
    T = x.shape[1]
    pl = (width - 1) // 2
    y = np.empty_like(x)
    for t in range(T):
        y[:, t] = x[:, max(t - pl, 0):min(t + width - 1 - pl, T - 1) + 1].max(axis=1)
    return y


def highwaynet(x, w, scope):
    """modules.py:100-123."""
    H = dense(x, w[scope + '/H/kernel'], w[scope + '/H/bias'], 'relu')
    T = dense(x, w[scope + '/T/kernel'], w[scope + '/T/bias'], 'sigmoid')
    return H * T + x * (1.0 - T)


def gru_cell(x, h, w, scope):
    """tf GRUCell.call (A.5): reset gate first, applied to the state BEFORE the candidate
    matmul; new_h = u*h + (1-u)*c. Used at modules.py:93-94 and decoder.py:19,30-31."""
    H = h.shape[-1]
    g = sigmoid(np.concatenate([x, h], -1) @ w[scope + 'gates/kernel'] + w[scope + 'gates/bias'])
    r, u = g[..., :H], g[..., H:]
    c = np.tanh(np.concatenate([x, r * h], -1) @ w[scope + 'candidate/kernel'] + w[scope + 'candidate/bias'])
    return u * h + (1.0 - u) * c


def _dynamic_rnn(x, lengths, w, scope, H):
    """tf.nn.dynamic_rnn with sequence_length: for t >= len the output is 0 and the state is
    carried unchanged (A.6)."""
    B, T, _ = x.shape
    h = np.zeros((B, H), x.dtype)
    out = np.zeros((B, T, H), x.dtype)
    for t in range(T):
        hn = gru_cell(x[:, t], h, w, scope)
        if lengths is None:
            h = hn
            out[:, t] = hn
        else:
            live = (t < lengths)[:, None]
            h = np.where(live, hn, h)
            out[:, t] = np.where(live, hn, 0)
    return out


def _reverse_sequence(x, lengths):
    if lengths is None:
        return x[:, ::-1]
    y = x.copy()
    for b, n in enumerate(lengths):
        y[b, :n] = x[b, :n][::-1]
    return y


def bidirectional_gru(x, lengths, w, scope, H):
    """tf.nn.bidirectional_dynamic_rnn(GRUCell, GRUCell, sequence_length) + concat (modules.py:92-98)."""
    fw = _dynamic_rnn(x, lengths, w, scope + '/bidirectional_rnn/fw/gru_cell/', H)
    bw = _dynamic_rnn(_reverse_sequence(x, lengths), lengths, w, scope + '/bidirectional_rnn/bw/gru_cell/', H)
    return np.concatenate([fw, _reverse_sequence(bw, lengths)], -1)


def prenet(x, w, scope, n_layers=2):
    """modules.py:4-28 with training=False: dropout is the identity."""
    for i in range(n_layers):
        s = '%s/prenet_dense_%d/' % (scope, i + 1)
        x = dense(x, w[s + 'kernel'], w[s + 'bias'], 'relu')
    return x


def cbhg(x, lengths, w, scope, K, n_highway, highway_depth, gru_cells, stages=None, pool_width=2, pool_stride=1):
    """modules.py:30-98. BN variables auto-number in the enclosing scope."""
    bank = np.concatenate(
        [convolute(x, w, '%s/conv_bank/conv_bank_%d' % (scope, k),
                   '%s/conv_bank/batch_normalization%s' % (scope, '' if k == 1 else '_%d' % (k - 1)), 'relu')
         for k in range(1, K + 1)], -1)                                            # :63-67
    pooled = maxpool2_same(bank) if (pool_width, pool_stride) == (2, 1) else maxpool_same(bank, pool_width, pool_stride)   # :70-71
    p1 = convolute(pooled, w, scope + '/proj_1', scope + '/batch_normalization', 'relu')   # :74-75
    p2 = convolute(p1, w, scope + '/proj_2', scope + '/batch_normalization_1', None)       # :76-77
    hw = p2 + x                                                                    # :80
    if hw.shape[2] != highway_depth:                                               # :83-84
        hw = dense(hw, w[scope + '/dense/kernel'], w[scope + '/dense/bias'])
    for i in range(n_highway):                                                     # :87-88
        hw = highwaynet(hw, w, '%s/highway_%d' % (scope, i + 1))
    out = bidirectional_gru(hw, lengths, w, scope, gru_cells)                      # :92-98
    if stages is not None:
        stages.update(bank=bank, proj1=p1, proj2_res=p2 + x, highway=hw, rnn=out)
    return out


# ----------------------------------------------------------------------------------------
# model path
# ----------------------------------------------------------------------------------------

_DEC = 'decoder/output_projection_wrapper/'
_CELL0 = _DEC + 'multi_rnn_cell/cell_0/output_projection_wrapper/'
_ATTW = _CELL0 + 'concat_output_and_attention_wrapper/attention_wrapper/'


def cast_weights(w, dtype):
    return {k: np.asarray(v, dtype) for k, v in w.items()}


def encode(ids, lengths, w, hp, stages=None):
    """batch.py:36-40 (embedding lookup) -> encoder.py:13-63 (prenet + CBHG K=16)."""
    emb = w['embedding'][ids]
    pre = prenet(emb, w, 'prenet', len(hp.prenet_depths))
    if stages is not None:
        stages['enc_prenet'] = pre
    return cbhg(pre, np.asarray(lengths), w, 'encoder_cbhg', hp.encoder_K, hp.encoder_num_highway_layers,
                hp.encoder_highway_depth, hp.encoder_gru_num_cells, stages,
                pool_width=hp.encoder_pooling_width, pool_stride=hp.encoder_pooling_stride)


def decode(enc_out, w, hp, max_iters=None, return_scores=False, targets=None):
    """decoder.py:18-42 + rnn_wrappers.py:24-26,51-53 + helpers.py:66-78 + TF's AttentionWrapper /
    BahdanauAttention / dynamic_decode (A.7). Returns (mel [B,steps*r,M], alignments
    [B,T_in,steps], steps_run[, scores [steps,B,T_in]]).
    targets [B,T_out,M]: teacher-forced as TrainingHelper does (helpers.py:6-43): the fed frames are
    targets[:, r-1::r], one step per fed frame, no stop test."""
    dt = enc_out.dtype
    B, T_in, _ = enc_out.shape
    A = hp.attention_depth
    D = hp.decoder_depth
    M = hp.num_mels
    r = hp.outputs_per_step
    S = hp.max_iters if max_iters is None else max_iters
    fed = None
    if targets is not None:
        fed = np.asarray(targets, dt)[:, r - 1::r, :]                 # helpers.py:15
        S = min(S, fed.shape[1]) if max_iters is not None else fed.shape[1]   # helpers.py:18-19,41
    keys = enc_out @ w['memory_layer/kernel']                    # BahdanauAttention memory_layer, no bias
    values = enc_out
    Wq = w[_ATTW + 'bahdanau_attention/query_layer/kernel']
    v = w[_ATTW + 'bahdanau_attention/attention_v']
    h_att = np.zeros((B, A), dt)
    ctx = np.zeros((B, enc_out.shape[2]), dt)
    h1 = np.zeros((B, D), dt)
    h2 = np.zeros((B, D), dt)
    prev = np.zeros((B, M), dt)                                   # helpers.py:80-85 <GO> frame
    finished = np.zeros(B, bool)
    outs, aligns, scores = [], [], []
    steps = 0
    while steps < S and not finished.all():
        p = prenet(np.concatenate([prev, ctx], -1), w, _ATTW + 'decoder_prenet_wrapper/decoder_prenet',
                   len(hp.prenet_depths))                         # rnn_wrappers.py:25
        h_att = gru_cell(p, h_att, w, _ATTW + 'decoder_prenet_wrapper/gru_cell/')   # rnn_wrappers.py:26
        q = h_att @ Wq
        e = np.sum(v * np.tanh(keys + q[:, None, :]), -1)         # _bahdanau_score, no mask (A.8)
        ex = np.exp(e - e.max(-1, keepdims=True))
        a = ex / ex.sum(-1, keepdims=True)
        ctx = np.einsum('bt,btd->bd', a, values)
        d0 = np.concatenate([h_att, ctx], -1) @ w[_CELL0 + 'kernel'] + w[_CELL0 + 'bias']  # rnn_wrappers.py:53, decoder.py:29
        h1 = gru_cell(d0, h1, w, _DEC + 'multi_rnn_cell/cell_1/gru_cell/')
        o1 = h1 + d0                                              # ResidualWrapper, decoder.py:30
        h2 = gru_cell(o1, h2, w, _DEC + 'multi_rnn_cell/cell_2/gru_cell/')
        o2 = h2 + o1                                              # decoder.py:31
        out = o2 @ w[_DEC + 'kernel'] + w[_DEC + 'bias']          # decoder.py:35
        outs.append(out)
        aligns.append(a)
        scores.append(e)
        if fed is None:
            prev = out[:, -M:]                                    # helpers.py:77
            finished = finished | np.all(out == 0.0, axis=1)      # helpers.py:75 + dynamic_decode's OR
        else:
            prev = fed[:, steps, :]                               # helpers.py:42 (time = steps)
        steps += 1
    mel = np.stack(outs, 1).reshape(B, steps * r, M)              # decoder.py:42
    align = np.stack(aligns, 0).transpose(1, 2, 0)                # tacotron.py:51
    if return_scores:
        return mel, align, steps, np.stack(scores, 0)
    return mel, align, steps


def postnet(mel, w, hp, stages=None):
    """decoder.py:44-57: post-CBHG (K=8, no sequence lengths) + dense to num_freq."""
    post = cbhg(mel, None, w, 'post_cbhg', hp.decoder_K, hp.decoder_num_highway_layers,
                hp.decoder_highway_depth, hp.decoder_gru_num_cells, stages,
                pool_width=hp.decoder_pooling_width, pool_stride=hp.decoder_pooling_stride)
    return dense(post, w['dense/kernel'], w['dense/bias'])


def tacotron_forward(ids, lengths, w, hp, dtype=np.float64, max_iters=None):
    """tacotron.py:26-57 with is_training=False. Returns dict(mel, linear, alignments, enc_out, steps)."""
    w = cast_weights(w, dtype)
    enc = encode(np.asarray(ids), lengths, w, hp)
    mel, align, steps = decode(enc, w, hp, max_iters)
    lin = postnet(mel, w, hp)
    return dict(enc_out=enc, mel=mel, linear=lin, alignments=align, steps=steps)


# ----------------------------------------------------------------------------------------
# Griffin-Lim, TF variant (tools/audio.py:57-64,183-205,226-227,276-277; A.9)
# ----------------------------------------------------------------------------------------


def hann_periodic(n, dtype=np.float64):
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(dtype)


def stft_tf(y, n_fft, hop, win):
    """tf.contrib.signal.stft(signals, win, hop, n_fft, pad_end=False) (audio.py:203-205)."""
    L = y.shape[-1]
    M = 1 + (L - win) // hop
    idx = hop * np.arange(M)[:, None] + np.arange(win)[None, :]
    frames = y[..., idx] * hann_periodic(win, y.dtype)
    return np.fft.rfft(frames, n_fft, axis=-1)


def istft_tf(X, n_fft, hop, win):
    """tf.contrib.signal.inverse_stft(stfts, win, hop, n_fft) with TF 1.15's default
    window_fn = periodic Hann and no window-sum normalisation (audio.py:199-201)."""
    frames = np.fft.irfft(X, n_fft, axis=-1)[..., :win]
    frames = frames * hann_periodic(win, frames.dtype)
    M = frames.shape[-2]
    y = np.zeros(frames.shape[:-2] + ((M - 1) * hop + win,), frames.dtype)
    for m in range(M):
        y[..., m * hop:m * hop + win] += frames[..., m, :]
    return y


def denorm_db_to_amp_pow(spect, hp):
    """audio.py:63-64: (10 ** ((clip(S,0,1)*100 - 100 + 20) * 0.05)) ** power."""
    S = np.clip(spect, 0, 1) * -float(hp.min_level_db) + hp.min_level_db
    S = np.power(10.0, (S + hp.ref_level_db) * 0.05)
    return np.power(S, hp.power)


def griffin_lim_tf(mag, hp, iters=None, dtype=np.float64):
    """audio.py:183-196 on [..., T, 1025] magnitudes: zero initial phase, `iters` iterations."""
    n_fft, hop, win = hp.stft_params()
    iters = hp.griffin_lim_iters if iters is None else iters
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    mag = mag.astype(dtype)
    y = istft_tf(mag.astype(cdt), n_fft, hop, win).astype(dtype)
    for _ in range(iters):
        est = stft_tf(y, n_fft, hop, win).astype(cdt)
        ang = est / np.maximum(1e-8, np.abs(est)).astype(dtype)
        y = istft_tf((mag * ang).astype(cdt), n_fft, hop, win).astype(dtype)
    return y


def spectrogram_tensorflow_inv(spect, hp, iters=None, dtype=np.float64):
    """audio.py:57-64 on a normalised linear spectrogram [..., T, 1025]; no de-emphasis."""
    return griffin_lim_tf(denorm_db_to_amp_pow(spect.astype(dtype), hp), hp, iters, dtype)


# ----------------------------------------------------------------------------------------
# host DSP (audio.py:15-17,81-88,111-116; A.11)
# ----------------------------------------------------------------------------------------


def pre_emphasis_inv(x, coef=0.97):
    """scipy.signal.lfilter([1], [1, -coef], x): y[n] = x[n] + coef*y[n-1], in float64."""
    y = np.empty(len(x), np.float64)
    acc = 0.0
    for n, v in enumerate(np.asarray(x, np.float64)):
        acc = v + coef * acc
        y[n] = acc
    return y


def find_endpoint(wav, sample_rate=24000, threshold_db=-40, min_silence_sec=0.8):
    window = int(sample_rate * min_silence_sec)
    hop = int(window / 4)
    thr = np.power(10.0, threshold_db * 0.05)
    for x in range(hop, len(wav) - window, hop):
        if np.max(wav[x:x + window]) < thr:
            return x + hop
    return len(wav)


def wav_to_int16(wav):
    """save_wav's scaling (audio.py:16-17), in place on a copy so that it happens in the input's own float type."""
    w = np.asarray(wav)
    w = w.astype(w.dtype if w.dtype.kind == 'f' else np.float64, copy=True)
    w *= 32767 / max(0.01, np.max(np.abs(w)))
    return w.astype(np.int16)


# ----------------------------------------------------------------------------------------
# numpy/librosa-flavoured Griffin-Lim (audio.py:44-55,119-148,167-181); secondary path
# ----------------------------------------------------------------------------------------


def _hann_padded(n_fft, win):
    wpad = np.zeros(n_fft)
    lp = (n_fft - win) // 2
    wpad[lp:lp + win] = hann_periodic(win)       # scipy get_window('hann', fftbins=True), centre-padded
    return wpad


def stft_librosa(y, n_fft, hop, win):
    """librosa.stft(center=True, reflect pad) -> [1+n_fft/2, frames] complex128."""
    yp = np.pad(y, n_fft // 2, mode='reflect')
    M = 1 + (len(yp) - n_fft) // hop
    idx = hop * np.arange(M)[:, None] + np.arange(n_fft)[None, :]
    return np.fft.rfft(yp[idx] * _hann_padded(n_fft, win), axis=-1).T


def istft_librosa(X, hop, win):
    """librosa 0.5.1 istft: window-sum-square normalised overlap-add, trimmed by n_fft//2."""
    n_fft = 2 * (X.shape[0] - 1)
    M = X.shape[1]
    w = _hann_padded(n_fft, win)
    L = n_fft + hop * (M - 1)
    y = np.zeros(L)
    wss = np.zeros(L)
    frames = np.fft.irfft(X.T, n_fft, axis=-1) * w
    for m in range(M):
        y[m * hop:m * hop + n_fft] += frames[m]
        wss[m * hop:m * hop + n_fft] += w * w
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2:-(n_fft // 2)]


def griffin_lim_librosa(spect, hp, u, iters=None):
    """audio.py:53,167-181 on [1025, T]: `u` [1025, T] are the uniform draws of np.random.rand (audio.py:175)."""
    n_fft, hop, win = hp.stft_params()
    iters = hp.griffin_lim_iters if iters is None else iters
    S = np.power(10.0, ((np.clip(spect, 0, 1) * -float(hp.min_level_db)) + hp.min_level_db + hp.ref_level_db) * 0.05)
    S = S ** hp.power
    ang = np.exp(2j * np.pi * u)
    Sc = np.abs(S).astype(complex)
    y = istft_librosa(Sc * ang, hop, win)
    for _ in range(iters):
        ang = np.exp(1j * np.angle(stft_librosa(y, n_fft, hop, win)))
        y = istft_librosa(Sc * ang, hop, win)
    return y


def spectrogram_inv_librosa(spect, hp, rng, iters=None):
    """audio.py:44-55 on [1025, T]; `rng` replaces the reference's unseeded np.random.rand."""
    return pre_emphasis_inv(griffin_lim_librosa(spect, hp, rng.random(spect.shape), iters), hp.preemphasis)


# ----------------------------------------------------------------------------------------
# analysis side (audio.py:23-42,66-79,91-109,212-264): what data_load._process_utterance stores
# ----------------------------------------------------------------------------------------
def pre_emphasis(x, coef=0.97):
    """scipy.signal.lfilter([1, -coef], [1], x): y[n] = x[n] - coef*x[n-1], zero initial state (audio.py:109)."""
    x = np.asarray(x, np.float64)
    y = x.copy()
    y[1:] -= coef * x[:-1]
    return y


def mel_basis_librosa(sr, n_fft, n_mels):
    """librosa 0.5.1 filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm=1), written filter by filter
    (an independent formulation of tacotron_amd.tools.audio.build_mel_basis). Library knowledge: unpinned."""
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz2mel(f):
        return min_log_mel + np.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp

    def mel2hz(m):
        return min_log_hz * np.exp(logstep * (m - min_log_mel)) if m >= min_log_mel else f_sp * m

    mmax = hz2mel(sr / 2.0)
    edges = [mel2hz(mmax * i / (n_mels + 1)) for i in range(n_mels + 2)]
    freqs = [sr / 2.0 * k / (n_fft // 2) for k in range(1 + n_fft // 2)]
    W = np.zeros((n_mels, len(freqs)))
    for i in range(n_mels):
        lo, ce, hi = edges[i], edges[i + 1], edges[i + 2]
        for k, f in enumerate(freqs):
            up = (f - lo) / (ce - lo)
            down = (hi - f) / (hi - ce)
            W[i, k] = max(0.0, min(up, down)) * 2.0 / (hi - lo)
    return W


def _normalize(S, hp):
    return np.clip((S - hp.min_level_db) / -float(hp.min_level_db), 0, 1)


def spectrogram_librosa(y, hp):
    """audio.py:23-42 -> [1025, frames] float64."""
    n_fft, hop, win = hp.stft_params()
    D = stft_librosa(pre_emphasis(y, hp.preemphasis), n_fft, hop, win)
    return _normalize(20 * np.log10(np.maximum(1e-5, np.abs(D))) - hp.ref_level_db, hp)


def mel_spectrogram_librosa(y, hp):
    """audio.py:66-79 -> [num_mels, frames] float64."""
    n_fft, hop, win = hp.stft_params()
    D = stft_librosa(pre_emphasis(y, hp.preemphasis), n_fft, hop, win)
    M = mel_basis_librosa(hp.sample_rate, n_fft, hp.num_mels) @ np.abs(D)
    return _normalize(20 * np.log10(np.maximum(1e-5, M)) - hp.ref_level_db, hp)
