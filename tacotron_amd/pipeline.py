"""Batches in flight: several TacoEngine handles on one GPU, each with its own HIP stream.

One synthesis is a chain of dependent launches (the decoder loop alone is 200 sequential steps on ~40 of the 256
CUs), so a single batch leaves most of the chip idle; `depth` independent batches overlap encoder / post-net GEMMs
of one batch with the decoder loop of the others (bench.py measures the headline number this way). The reference has
no counterpart (one tf.Session, one `session.run` at a time, synthesizer.py:39-43); results per batch are identical
to a lone `engine.synthesize` because no kernel's reduction order depends on what else runs.

More than 4 streams only overlap when the process may open more hardware queues than HIP's default:
set GPU_MAX_HW_QUEUES (8 is what bench.py uses) before the HIP runtime loads, i.e. before `import torch`."""
import collections

import numpy as np
import torch

from .engine import TacoEngine
from .shard import gather_order, plan_batches


class SynthesisPipeline:
    """`devices`: the GPUs of the node to shard a job over (utterances are independent, weights are
    replicated, no collective; BASELINE.json north_star: "utterance batches shard trivially across the 8 GPUs of one
    node"). One set of `depth` engines is created per entry (an id may repeat: two sets on one GPU); `run` uses the
    first set, `synthesize_sharded` all of them."""

    def __init__(self, weights, hp=None, device=0, depth=6, decode_mode=2, devices=None):
        if depth < 1:
            raise ValueError('depth must be >= 1')
        self.devices = [device] if devices is None else list(devices)
        if not self.devices:
            raise ValueError('devices must name at least one GPU')
        self.sets = [[TacoEngine(weights, hp, device=d) for _ in range(depth)] for d in self.devices]
        self.engines = self.sets[0]
        self.hp = self.engines[0].hp
        for es in self.sets:
            for e in es:
                e.set_decode_graph(decode_mode)
                e.detached = True
        self._set_outs = [[{} for _ in es] for es in self.sets]
        self._budget_before = {}
        self._outs = self._set_outs[0]

    def _set_budget(self, to_wav, batch):
        """Decodes side by side: as many as fit the chip for mel + linear jobs; one at a time when the vocoder's
        full-chip launches of other batches run beside them (include/taco.h, taco_set_persist_budget). The knob is
        per process and device, not per pipeline: the value found at the first call is put back by close(); two
        pipelines that run jobs of different kinds on one GPU at the same time share the last value set (throughput
        only: results do not depend on it)."""
        for es in self.sets:
            dev = int(es[0].device.index or 0)
            if dev not in self._budget_before:
                self._budget_before[dev] = es[0].get_persist_budget()
            es[0].set_persist_budget(32 * ((batch + 15) // 16) if to_wav else 0)

    def close(self):
        for es in self.sets:
            if es:
                dev = int(es[0].device.index or 0)
                if dev in self._budget_before:       # (the budget is per process and device: leave what was found)
                    es[0].set_persist_budget(self._budget_before.pop(dev))
            for e in es:
                e.close()
        self.sets, self.engines = [], []

    def synthesize_sharded(self, ids, lens, batch_size=32, max_iters=None, to_wav=False, gl_iters=None,
                           keys=None, trim=False):
        """One job of N utterances (ids [N, T_in] int32 right-padded with 0, lens [N]) over all engine sets:
        the utterances are length-sorted and dealt into one contiguous shard per set (shard.shard_by_length), each
        shard runs as batches of `batch_size` with up to `depth` batches in flight per set, and the results come back
        in the ORIGINAL order (shard.gather_order) as a list of per-utterance dicts of host arrays (`keys` that were
        produced, plus `steps_run`). No device talks to another. `keys` defaults to what the job is for: ('wav',) when
        to_wav (what synthesizer.py:43 fetches), else ('mel', 'linear', 'alignments').
        The padded width is part of the reference's semantics (convolutions and the unmasked attention see the padding,
        SURVEY.md A.6/A.8), so by default every batch keeps the job's T_in and an utterance's result is bit-identical
        to the unsharded run; trim=True pads each batch only to its own longest utterance (what a length-sorted
        feeder is for: less work, the result of the reference run on that smaller batch).

        Host side (round 3): the results of a shard live in ONE pinned host tensor per key, [len(shard), ...]; the
        download of a batch is enqueued on its engine's own stream right behind the synthesis (non-blocking copies
        into that batch's rows), so nothing on the host stands between a device's batches: the host thread only launches,
        and waits on a batch's event when that engine slot comes up for reuse. The returned arrays are views of the
        pinned tensors (no second copy). The stop test (helpers.py:75) is checked per batch from the downloaded step
        count; a batch it fired in (never, in practice) is redone the slow way through TacoEngine.finish."""
        ids = np.ascontiguousarray(ids, np.int32)
        lens = np.ascontiguousarray(lens, np.int32)
        if ids.ndim != 2 or lens.shape != (ids.shape[0],):
            raise ValueError('ids must be [N, T_in] and lens [N]')
        if batch_size < 1:
            raise ValueError('batch_size must be >= 1')
        if keys is None:
            keys = ('wav',) if to_wav else ('mel', 'linear', 'alignments')
        keys = tuple(k for k in keys if k != 'wav' or to_wav)
        hp = self.hp
        S = int(max_iters or hp.max_iters)
        T = S * hp.outputs_per_step
        _, hop, win = hp.stft_params()
        T_in = ids.shape[1]
        row_shape = {'mel': (T, hp.num_mels), 'linear': (T, hp.num_freq), 'alignments': (T_in, S),
                     'wav': ((T - 1) * hop + win,), 'enc_out': (T_in, 2 * hp.encoder_gru_num_cells)}
        for k in keys:
            if k not in row_shape:
                raise ValueError('unknown key %r' % (k,))
        per_batch = {k for k in keys if trim and k in ('alignments', 'enc_out')}     # row shape varies with the batch
        shards, plan = plan_batches(lens, len(self.sets), batch_size)
        self._set_budget(to_wav, batch_size)
        host = [{k: torch.empty((len(idx),) + row_shape[k], dtype=torch.float32, pin_memory=True)
                 for k in keys if k not in per_batch} if len(idx) else {} for idx in shards]
        tasks = collections.deque(plan)
        results = [[None] * len(idx) for idx in shards]                 # per set, in shard order
        launched_rows = [0] * len(self.sets)
        inflight = [collections.deque() for _ in self.sets]
        launched = [0] * len(self.sets)

        def collect(si):
            ei, ev, n, row0, steps_h, extra = inflight[si].popleft()
            ev.synchronize()
            steps = int(steps_h.item())
            e = self.sets[si][ei]
            if steps < S:
                # the stop test fired: redo the length-dependent stages on the trimmed mel (TacoEngine.finish)
                with torch.cuda.device(e.device), torch.cuda.stream(e.stream):
                    o = e.finish(self._set_outs[si][ei], to_wav=to_wav, gl_iters=gl_iters)
                    slow = {k: o[k][:n].cpu().numpy() for k in keys if k in o and torch.is_tensor(o[k])}
                e.stream.synchronize()
            else:
                slow = None
            for b in range(n):
                if slow is not None:
                    r = {k: v[b] for k, v in slow.items()}
                else:
                    r = {k: (extra[k][b] if k in per_batch else host[si][k][row0 + b]).numpy() for k in keys}
                r['steps_run'] = steps
                results[si][row0 + b] = r

        while tasks:
            si, idx = tasks.popleft()
            es = self.sets[si]
            if len(inflight[si]) == len(es):
                collect(si)
            ei = launched[si] % len(es)
            launched[si] += 1
            e = es[ei]
            b_ids, b_lens = ids[idx], lens[idx]
            if trim:
                b_ids = np.ascontiguousarray(b_ids[:, :int(b_lens.max())])
            n, row0 = len(idx), launched_rows[si]
            launched_rows[si] += n
            with torch.cuda.device(e.device):
                o = e.synthesize(b_ids, b_lens, max_iters=max_iters, to_wav=to_wav, gl_iters=gl_iters,
                                 out=self._set_outs[si][ei])
                steps_h = torch.empty(1, dtype=torch.int32, pin_memory=True)
                extra = {}
                with torch.cuda.stream(e.stream):       # downloads ride the engine's stream, behind its kernels
                    for k in keys:
                        if k in per_batch:
                            extra[k] = torch.empty(tuple(o[k][:n].shape), dtype=torch.float32, pin_memory=True)
                            extra[k].copy_(o[k][:n], non_blocking=True)
                        else:
                            host[si][k][row0:row0 + n].copy_(o[k][:n], non_blocking=True)
                    steps_h.copy_(o['steps'], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(e.stream)
            inflight[si].append((ei, ev, n, row0, steps_h, extra))
        for si in range(len(self.sets)):
            while inflight[si]:
                collect(si)
        # a persistent decode whose bounded wait ran out is reported by the next decode of its handle (no
        # synchronisation, include/taco.h); the last one of each engine is asked for here, once, after its stream is idle
        for si, es in enumerate(self.sets):
            for e in es[:launched[si]]:
                e.stream.synchronize()
                e.decode_status()
        flat = [r for rs in results for r in rs]
        return [flat[j] for j in gather_order(shards)]

    def run(self, batches, max_iters=None, to_wav=False, gl_iters=None, to_host=True):
        """`batches`: iterable of (ids [B, T_in] int32, lens [B] int32). Yields one result dict per batch, in order
        (`mel`, `linear`, `alignments`, `steps_run`, `wav` if to_wav), while up to `depth` later batches already run.
        With to_host (default) the tensors are copied to host memory before the engine's buffers are reused; with
        to_host=False they are device tensors owned by the pipeline, valid until the next result is requested."""
        inflight = collections.deque()
        n = len(self.engines)
        k = 0

        def collect():
            i, ev = inflight.popleft()
            ev.synchronize()
            e = self.engines[i]
            with torch.cuda.stream(e.stream):       # a trimmed re-run (stop test fired) stays on this engine's stream
                o = e.finish(self._outs[i], to_wav=to_wav, gl_iters=gl_iters)
                if to_host:
                    o = {name: (t.cpu() if torch.is_tensor(t) else t) for name, t in o.items()}
            e.stream.synchronize()
            return o

        budget_set = False
        for ids, lens in batches:
            if not budget_set:
                self._set_budget(to_wav, int(ids.shape[0]))
                budget_set = True
            if len(inflight) == n:
                yield collect()
            i = k % n
            k += 1
            e = self.engines[i]
            e.synthesize(ids, lens, max_iters=max_iters, to_wav=to_wav, gl_iters=gl_iters, out=self._outs[i])
            ev = torch.cuda.Event()
            ev.record(e.stream)
            inflight.append((i, ev))
        while inflight:
            yield collect()
