"""Batches in flight: several TacoEngine handles on one GPU, each with its own HIP stream.

One synthesis is a chain of dependent launches (the decoder loop alone is 200 sequential steps on ~40 of the 256
CUs), so a single batch leaves most of the chip idle; `depth` independent batches overlap encoder / post-net GEMMs
of one batch with the decoder loop of the others (bench.py measures the headline number this way). The reference has
no counterpart (one tf.Session, one `session.run` at a time, synthesizer.py:39-43); results per batch are identical
to a lone `engine.synthesize` because no kernel's reduction order depends on what else runs.

More than 4 streams only overlap when the process may open more hardware queues than HIP's default:
set GPU_MAX_HW_QUEUES (8 is what bench.py uses) before the HIP runtime loads, i.e. before `import torch`."""
import collections

import torch

from .engine import TacoEngine


class SynthesisPipeline:
    def __init__(self, weights, hp=None, device=0, depth=6, decode_mode=2):
        if depth < 1:
            raise ValueError('depth must be >= 1')
        self.engines = [TacoEngine(weights, hp, device=device) for _ in range(depth)]
        self.hp = self.engines[0].hp
        for e in self.engines:
            e.set_decode_graph(decode_mode)
            e.detached = True
        self._outs = [{} for _ in self.engines]

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []

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

        for ids, lens in batches:
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
