"""TacoEngine: the Python face of one libtaco_hip handle (one GPU).

PyTorch-ROCm is used for exactly three things: device buffers (`torch.empty(..., device=)` +
`.data_ptr()`), a HIP stream, and host<->device copies. All arithmetic happens inside the
C-ABI calls (include/taco.h). Mirrors what one `tf.Session` does for the reference
(synthesizer.py:29-43): holds the restored weights and runs the graph on fed ids."""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .hparams import hparams as _default_hparams
from .weights import check_weights


class TacoError(RuntimeError):
    pass


class TacoEngine:
    def __init__(self, weights, hp=None, device=0):
        self.hp = hp or _default_hparams
        self.lib = _lib.load_library()
        if not torch.cuda.is_available():
            raise TacoError('TacoEngine needs a GPU (torch.cuda.is_available() is False); there is no CPU path')
        check_weights(weights, self.hp)
        self.device = torch.device('cuda', device)
        self._hp_struct = _lib.make_hparams_struct(self.hp)
        weights = dict(weights)
        if 'audio/mel_basis' not in weights:      # not a TF variable: derived from hparams (tools/audio.py:246-252)
            from .tools.audio import build_mel_basis
            weights['audio/mel_basis'] = build_mel_basis(self.hp)
        names = sorted(weights)
        arrays = [np.ascontiguousarray(weights[n], dtype=np.float32) for n in names]
        entries = (_lib.TacoWeightEntry * len(names))()
        for e, n, a in zip(entries, names, arrays):
            e.name = n.encode()
            e.data = a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            e.ndim = a.ndim
            for i, d in enumerate(a.shape):
                e.dims[i] = d
        h = ctypes.c_void_p()
        if self.lib.taco_create(ctypes.byref(self._hp_struct), entries, len(names), device, ctypes.byref(h)) != 0:
            raise TacoError(_lib.last_error())
        self.handle = h
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream(device=self.device)
        self._ws = None

    def close(self):
        if getattr(self, 'handle', None):
            torch.cuda.synchronize(self.device)
            self.lib.taco_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def _check(self, rc):
        if rc != 0:
            raise TacoError(_lib.last_error())

    def _workspace(self, B, T_in, S):
        need = int(self.lib.taco_workspace_bytes(self.handle, B, T_in, S))
        if need == 0:
            raise TacoError('taco_workspace_bytes rejected B=%d T_in=%d max_iters=%d' % (B, T_in, S))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = self._empty((need,), torch.uint8)
        return self._ws

    def _empty(self, shape, dtype):
        """torch.empty on the engine's device, marked as in use on the engine's stream: the caching allocator, which
        hands freed blocks straight back to the allocating (current) stream, must not recycle a buffer that kernels
        of this engine's stream may still be writing - a caller that drops an output while the engine runs detached
        would otherwise let the next allocation (another engine's workspace, say) alias it."""
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        t.record_stream(self.stream)
        return t

    def _dev(self, x, dtype):
        t = torch.as_tensor(np.ascontiguousarray(x) if isinstance(x, np.ndarray) else x)
        if t.device.type == 'cpu':
            # host data: the upload is issued on the engine's own stream, so it is ordered before the kernels that
            # read it also when the engine runs detached from torch's current stream
            with torch.cuda.stream(self.stream):
                t = t.to(device=self.device, dtype=dtype).contiguous()
        else:
            t = t.to(device=self.device, dtype=dtype).contiguous()
        t.record_stream(self.stream)
        return t

    # By default every call is ordered after torch's current stream and torch's current stream after it, so
    # callers can mix engine calls with ordinary torch code. `detached = True` drops both edges: the engine's
    # stream then runs independently (several engines = several batches in flight on one GPU) and the caller
    # synchronises explicitly (torch.cuda.synchronize / engine.stream.synchronize).
    detached = False

    def _enter(self):
        if not self.detached:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def _leave(self):
        if not self.detached:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)

    @property
    def _s(self):
        return ctypes.c_void_p(self.stream.cuda_stream)

    def set_decode_graph(self, mode):
        """0/False = eager per-stage kernels, 1/True = hipGraph replay of them, 2 = persistent kernel (default)."""
        self._check(self.lib.taco_set_decode_graph(self.handle, int(mode)))

    def set_persist_budget(self, workgroups):
        """CUs that persistent decodes of this process may hold on this engine's device at a time (0 = all): see
        include/taco.h. One decode at a time (32 * ceil(B / 16)) is the better setting when vocoder launches of
        other batches run beside the decodes."""
        self._check(self.lib.taco_set_persist_budget(int(self.device.index or 0), int(workgroups)))

    def set_gl_split(self, groups):
        """Griffin-Lim of a batch as 1..4 utterance groups on streams of the handle's own (include/taco.h): 2 is the
        better setting for a batch that has the GPU to itself, 1 (default) with several batches in flight."""
        self._check(self.lib.taco_set_gl_split(self.handle, int(groups)))

    def set_bigru_mode(self, mode):
        """CBHG BiGRU scan: 0 = one utterance per workgroup (default, and the fastest for every job shape measured),
        1 / 2 = sixteen / four utterances per workgroup on fp32 MFMA (less CU time, longer chain). include/taco.h."""
        self._check(self.lib.taco_set_bigru_mode(self.handle, int(mode)))

    def get_persist_budget(self):
        return int(self.lib.taco_get_persist_budget(int(self.device.index or 0)))

    def decode_status(self):
        """After a synchronisation: raises if the last persistent decode hit a barrier timeout."""
        self._check(self.lib.taco_decode_status(self.handle))

    # ---- stages (tensors in, tensors out; all on self.device)
    def encode(self, ids, lens):
        ids = self._dev(ids, torch.int32)
        lens = self._dev(lens, torch.int32)
        B, T_in = ids.shape
        if lens.shape != (B,):
            raise TacoError('input_lengths must have shape [B]')
        ws = self._workspace(B, T_in, self.hp.max_iters)
        enc = self._empty((B, T_in, 2 * self.hp.encoder_gru_num_cells,), torch.float32)
        self._enter()
        self._check(self.lib.taco_encode(self.handle, ids.data_ptr(), lens.data_ptr(), B, T_in, enc.data_ptr(),
                                         ws.data_ptr(), ws.numel(), self._s))
        self._leave()
        return enc

    def decode(self, enc_out, max_iters=None, dropout_keep=None, targets=None):
        """Free-running decode (TestingHelper), or teacher-forced when `targets` [B, T_out, num_mels] is given
        (TrainingHelper, model/helpers.py:6-43: T_out // r steps, step t > 0 is fed targets[:, t*r - 1])."""
        enc_out = self._dev(enc_out, torch.float32)
        B, T_in, _ = enc_out.shape
        r, M = self.hp.outputs_per_step, self.hp.num_mels
        if targets is not None:
            targets = self._dev(targets, torch.float32)
            if targets.dim() != 3 or targets.shape[0] != B or targets.shape[2] != M or targets.shape[1] < r:
                raise TacoError('targets must be [B, T_out >= r, num_mels]')
            # len(targets[:, r-1::r]) steps (helpers.py:15,18), bounded by maximum_iterations (decoder.py:40)
            S = (targets.shape[1] - (r - 1) + r - 1) // r
            if max_iters:
                S = min(S, int(max_iters))
            if targets.shape[1] < S * r:      # the last group is incomplete: only its first frames exist
                pad = torch.zeros(B, S * r - targets.shape[1], M, dtype=torch.float32, device=self.device)
                targets = torch.cat([targets, pad], 1)
            targets = targets[:, :S * r].contiguous()
            targets.record_stream(self.stream)
        else:
            S = int(max_iters or self.hp.max_iters)
        ws = self._workspace(B, T_in, S)
        mel = self._empty((B, S * r, M,), torch.float32)
        align = self._empty((B, T_in, S,), torch.float32)
        steps = torch.zeros(1, dtype=torch.int32, device=self.device)
        steps.record_stream(self.stream)
        keep = None
        if dropout_keep is not None:
            keep = self._dev(dropout_keep, torch.float32)
            if tuple(keep.shape) != (S, B, sum(self.hp.prenet_depths)):
                raise TacoError('dropout_keep must be [max_iters, B, %d]' % sum(self.hp.prenet_depths))
        self._enter()
        if targets is not None:
            self._check(self.lib.taco_decode_teacher(self.handle, enc_out.data_ptr(), B, T_in, S, targets.data_ptr(),
                                                     keep.data_ptr() if keep is not None else None,
                                                     mel.data_ptr(), align.data_ptr(), steps.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), self._s))
        else:
            self._check(self.lib.taco_decode(self.handle, enc_out.data_ptr(), B, T_in, S,
                                             keep.data_ptr() if keep is not None else None,
                                             mel.data_ptr(), align.data_ptr(), steps.data_ptr(),
                                             ws.data_ptr(), ws.numel(), self._s))
        self._leave()
        return mel, align, steps

    def postnet(self, mel):
        mel = self._dev(mel, torch.float32)
        B, T, _ = mel.shape
        S = -(-T // self.hp.outputs_per_step)
        ws = self._workspace(B, 1, S)
        lin = self._empty((B, T, self.hp.num_freq,), torch.float32)
        self._enter()
        self._check(self.lib.taco_postnet(self.handle, mel.data_ptr(), B, T, lin.data_ptr(), ws.data_ptr(),
                                          ws.numel(), self._s))
        self._leave()
        return lin

    def griffin_lim(self, linear, iters=None):
        linear = self._dev(linear, torch.float32)
        squeeze = linear.dim() == 2
        if squeeze:
            linear = linear[None]
        B, T, _ = linear.shape
        iters = self.hp.griffin_lim_iters if iters is None else int(iters)
        _, hop, win = self.hp.stft_params()
        S = -(-T // self.hp.outputs_per_step)
        ws = self._workspace(B, 1, S)
        wav = self._empty((B, (T - 1) * hop + win,), torch.float32)
        self._enter()
        self._check(self.lib.taco_griffin_lim(self.handle, linear.data_ptr(), B, T, iters, wav.data_ptr(),
                                              ws.data_ptr(), ws.numel(), self._s))
        self._leave()
        return wav[0] if squeeze else wav

    def griffin_lim_noalloc(self, linear, iters=None, out=None):
        """taco_griffin_lim_noalloc: the same vocoder on the library's static device scratch (B*T <= 1024 frames); the
        call allocates nothing when `out` [B, (T-1)*hop+win] is given. Calls that overlap on one GPU race."""
        linear = self._dev(linear, torch.float32)
        squeeze = linear.dim() == 2
        if squeeze:
            linear = linear[None]
        B, T, _ = linear.shape
        iters = self.hp.griffin_lim_iters if iters is None else int(iters)
        _, hop, win = self.hp.stft_params()
        wav = out if out is not None else self._empty((B, (T - 1) * hop + win,), torch.float32)
        self._enter()
        self._check(self.lib.taco_griffin_lim_noalloc(self.handle, linear.data_ptr(), B, T, iters, wav.data_ptr(),
                                                      self._s))
        self._leave()
        return wav[0] if squeeze and out is None else wav

    def griffin_lim_librosa(self, linear, phase_u=None, iters=None):
        """tools/audio.py:44-55,167-181 without the final pre_emphasis_inv: linear [B, T, num_freq] (or [T, num_freq])
        normalised spectrogram, phase_u same shape of uniform [0, 1) draws (None = zero phase) ->
        wav [B, hop * (T - 1)]."""
        linear = self._dev(linear, torch.float32)
        squeeze = linear.dim() == 2
        if squeeze:
            linear = linear[None]
        B, T, _ = linear.shape
        if phase_u is not None:
            phase_u = self._dev(phase_u, torch.float32).reshape(linear.shape)
        iters = self.hp.griffin_lim_iters if iters is None else int(iters)
        _, hop, win = self.hp.stft_params()
        S = -(-T // self.hp.outputs_per_step)
        ws = self._workspace(B, 1, S)
        wav = self._empty((B, (T - 1) * hop,), torch.float32)
        self._enter()
        self._check(self.lib.taco_griffin_lim_librosa(self.handle, linear.data_ptr(),
                                                      phase_u.data_ptr() if phase_u is not None else None, B, T, iters,
                                                      wav.data_ptr(), ws.data_ptr(), ws.numel(), self._s))
        self._leave()
        return wav[0] if squeeze else wav

    def spectrogram(self, wav, want_linear=True, want_mel=True):
        """Analysis side (tools/audio.py:23-42,66-79): wav [B, n] (or [n]) -> (linear [B, T, num_freq],
        mel [B, T, num_mels]) normalised dB spectrograms, T = 1 + n // hop; either may be skipped (None)."""
        wav = self._dev(wav, torch.float32)
        if wav.dim() == 1:
            wav = wav[None]
        B, n = wav.shape
        _, hop, _ = self.hp.stft_params()
        T = 1 + n // hop
        need = int(self.lib.taco_spectrogram_workspace_bytes(self.handle, B, n))
        if need == 0:
            raise TacoError('taco_spectrogram_workspace_bytes rejected B=%d n=%d' % (B, n))
        ws = self._empty((need,), torch.uint8)
        lin = self._empty((B, T, self.hp.num_freq), torch.float32) if want_linear else None
        mel = self._empty((B, T, self.hp.num_mels), torch.float32) if want_mel else None
        self._enter()
        self._check(self.lib.taco_spectrogram(self.handle, wav.data_ptr(), B, n, float(self.hp.preemphasis),
                                              lin.data_ptr() if lin is not None else None,
                                              mel.data_ptr() if mel is not None else None,
                                              ws.data_ptr(), ws.numel(), self._s))
        self._leave()
        return lin, mel

    def synthesize(self, ids, lens, max_iters=None, to_wav=False, gl_iters=None, out=None):
        """encode -> decode -> postnet [-> Griffin-Lim] in one C call. Returns a dict of device
        tensors; `steps` is read back (one int) to trim outputs if the stop test fired.
        `out` may hold preallocated tensors from a previous call (bench loop)."""
        ids = self._dev(ids, torch.int32)
        lens = self._dev(lens, torch.int32)
        B, T_in = ids.shape
        S = int(max_iters or self.hp.max_iters)
        r, M = self.hp.outputs_per_step, self.hp.num_mels
        T = S * r
        _, hop, win = self.hp.stft_params()
        ws = self._workspace(B, T_in, S)
        o = out if out is not None else {}
        def buf(name, shape, dtype=torch.float32):
            t = o.get(name)
            if t is None or tuple(t.shape) != tuple(shape):
                t = self._empty(tuple(shape), dtype)
                o[name] = t
            return t
        enc = buf('enc_out', (B, T_in, 2 * self.hp.encoder_gru_num_cells))
        mel = buf('mel', (B, T, M))
        lin = buf('linear', (B, T, self.hp.num_freq))
        align = buf('alignments', (B, T_in, S))
        steps = buf('steps', (1,), torch.int32)
        wav = buf('wav', (B, (T - 1) * hop + win)) if to_wav else None
        gl = self.hp.griffin_lim_iters if gl_iters is None else int(gl_iters)
        self._enter()
        self._check(self.lib.taco_synthesize(self.handle, ids.data_ptr(), lens.data_ptr(), B, T_in, S, gl,
                                             enc.data_ptr(), mel.data_ptr(), lin.data_ptr(), align.data_ptr(),
                                             wav.data_ptr() if wav is not None else None, steps.data_ptr(),
                                             ws.data_ptr(), ws.numel(), self._s))
        self._leave()
        return o

    def finish(self, o, to_wav=False, gl_iters=None):
        """Host-side check of the stop test (helpers.py:75): if dynamic_decode would have stopped
        early, trim and redo the length-dependent stages on the shorter mel."""
        steps = int(o['steps'].item())
        self.decode_status()
        S = o['alignments'].shape[2]
        if steps < S:
            r = self.hp.outputs_per_step
            o = dict(o)
            o['mel'] = o['mel'][:, :steps * r].contiguous()
            o['alignments'] = o['alignments'][:, :, :steps].contiguous()
            o['linear'] = self.postnet(o['mel'])
            if to_wav:
                o['wav'] = self.griffin_lim(o['linear'], gl_iters)
        o['steps_run'] = steps
        return o

    # ---- measurement hooks
    def kernel_kinds(self):
        n = self.lib.taco_num_kernel_kinds()
        return [self.lib.taco_kernel_name(i).decode() for i in range(n)]

    def handoff_probe(self, rounds=4000):
        """us per dependent all-to-all granule hand-off among 32 workgroups on this device (include/taco.h)."""
        us = ctypes.c_double()
        self._check(self.lib.taco_debug_handoff_probe(int(self.device.index or 0), int(rounds), ctypes.byref(us)))
        return us.value

    def timing_begin(self, kind):
        kinds = self.kernel_kinds()
        self._check(self.lib.taco_timing_begin(self.handle, kinds.index(kind) if isinstance(kind, str) else kind))

    def timing_end(self):
        n = ctypes.c_int32()
        ms = ctypes.c_double()
        self._check(self.lib.taco_timing_end(self.handle, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value
