"""The C-ABI library loads on a machine without a GPU and exports every symbol include/taco.h
declares; the product path refuses to run without a GPU (no CPU fallback); the N>1 launch path
(rank env, shard selection, barrier + max-over-ranks) works under gloo with world_size 2."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, 'include', 'taco.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(taco_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from tacotron_amd import _lib
    lib = _lib.load_library()
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names
    assert lib.taco_num_kernel_kinds() == 13
    assert lib.taco_kernel_name(4) == b'dec_gemm'


def test_hparams_struct_matches_header():
    import ctypes
    from tacotron_amd import _lib
    from tacotron_amd.hparams import HParams
    s = _lib.make_hparams_struct(HParams())
    assert ctypes.sizeof(s) == 38 * 4 and s.struct_size == 38 * 4
    assert (s.num_chars, s.prenet_depth2, s.encoder_K, s.decoder_proj2, s.hop_length, s.win_length) == (95, 128, 16, 80, 300, 1200)
    hdr = open(os.path.join(ROOT, 'include', 'taco.h')).read()
    body = hdr[hdr.index('typedef struct TacoHParams {'):hdr.index('} TacoHParams;')]
    fields = re.findall(r'^\s*(u?int32_t|float)\s+(\w+);', body, flags=re.M)
    ctype = {'uint32_t': ctypes.c_uint32, 'int32_t': ctypes.c_int32, 'float': ctypes.c_float}
    assert [(n, ctype[t]) for t, n in fields] == list(_lib.TacoHParams._fields_)
    assert fields[0] == ('uint32_t', 'struct_size')


def test_integration_stub_struct_matches_header():
    """The reference-side binding shown in INTEGRATION.md declares TacoHParams field for field as include/taco.h does:
    the stub's class is extracted from the document, executed, and compared with the header (names, types, order)."""
    import ctypes
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    start = md.index('class TacoHParams(ctypes.Structure):')
    stub = md[start:md.index('class TacoWeightEntry(ctypes.Structure):', start)]
    ns = {'ctypes': ctypes}
    exec(stub, ns)
    hdr = open(os.path.join(ROOT, 'include', 'taco.h')).read()
    body = hdr[hdr.index('typedef struct TacoHParams {'):hdr.index('} TacoHParams;')]
    ctype = {'uint32_t': ctypes.c_uint32, 'int32_t': ctypes.c_int32, 'float': ctypes.c_float}
    want = [(n, ctype[t]) for t, n in re.findall(r'^\s*(u?int32_t|float)\s+(\w+);', body, flags=re.M)]
    assert list(ns['TacoHParams']._fields_) == want and len(want) == 38
    from tacotron_amd import _lib
    assert ctypes.sizeof(ns['TacoHParams']) == ctypes.sizeof(_lib.TacoHParams)


def test_create_rejects_wrong_struct_size():
    """A binding built against an older (shorter) or newer header is refused before any other field is read."""
    import ctypes
    from tacotron_amd import _lib
    from tacotron_amd.hparams import HParams
    lib = _lib.load_library()
    for size in (0, 33 * 4, 37 * 4, 39 * 4):
        s = _lib.make_hparams_struct(HParams())
        s.struct_size = size
        entries = (_lib.TacoWeightEntry * 1)()
        h = ctypes.c_void_p()
        assert lib.taco_create(ctypes.byref(s), entries, 0, 0, ctypes.byref(h)) < 0
        assert 'struct_size' in _lib.last_error() and not h.value


def test_create_rejects_pooling_stride():
    """hparams.py:31,41 -> modules.py:70-71,80: any pooling window runs, but a stride other than 1 shortens the pooled
    sequence, so the residual `proj2 + inputs` cannot be formed (the reference graph does not build either): refused at
    taco_create (before any device call) instead of silently computing a different graph."""
    import ctypes
    from tacotron_amd import _lib
    from tacotron_amd.hparams import HParams
    lib = _lib.load_library()
    for name, val in (('encoder_pooling_stride', 2), ('decoder_pooling_stride', 3), ('encoder_pooling_width', 0),
                      ('decoder_pooling_width', 65)):
        hp = HParams()
        hp.set_hparam(name, val)
        s = _lib.make_hparams_struct(hp)
        entries = (_lib.TacoWeightEntry * 1)()
        h = ctypes.c_void_p()
        assert lib.taco_create(ctypes.byref(s), entries, 0, 0, ctypes.byref(h)) < 0
        assert 'max-pool' in _lib.last_error() and not h.value


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from tacotron_amd.engine import TacoEngine, TacoError
    from tacotron_amd.hparams import HParams
    from tacotron_amd.weights import init_weights
    with pytest.raises(TacoError):
        TacoEngine(init_weights(HParams(), seed=0), HParams())


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'tacotron_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(d, f), encoding='utf-8').read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, os.path.join(d, f)


_WORKER = r'''
import os, sys, json
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %r)
from tacotron_amd.shard import shard_bounds
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
lo, hi = shard_bounds(64, world)[rank]
mine = torch.zeros(64, dtype=torch.int32); mine[lo:hi] = 1
dist.barrier()
t = torch.tensor([0.5 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.all_reduce(mine, op=dist.ReduceOp.SUM)
if rank == 0:
    print(json.dumps({'max_t': float(t), 'covered': int(mine.sum()), 'once': bool((mine == 1).all()), 'per_rank': hi - lo}))
dist.barrier(); dist.destroy_process_group()
''' % ROOT


def test_two_rank_gloo_shard_and_timing(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', '29611', str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out == {'max_t': 1.5, 'covered': 64, 'once': True, 'per_rank': 32}


def test_plan_batches_covers_every_utterance_once():
    """shard.plan_batches (the launch plan of SynthesisPipeline.synthesize_sharded): contiguous length-sorted shards,
    batches issued round-robin over the ranks, every utterance exactly once; gather_order restores the input order."""
    import numpy as np
    from tacotron_amd.shard import gather_order, plan_batches
    rng = np.random.default_rng(0)
    for n, world, bs in [(10, 2, 3), (1, 2, 4), (0, 3, 2), (17, 8, 2), (64, 8, 32), (7, 1, 7)]:
        lens = rng.integers(2, 50, n)
        shards, tasks = plan_batches(lens, world, bs)
        assert len(shards) == world and sorted(np.concatenate(shards).tolist()) == list(range(n))
        seen = [[] for _ in range(world)]
        for r, idx in tasks:
            assert 1 <= len(idx) <= bs
            seen[r] += idx.tolist()
        for r in range(world):
            assert seen[r] == shards[r].tolist()                   # shard order kept, nothing twice
            assert all(lens[a] >= lens[b] for a, b in zip(seen[r], seen[r][1:]))   # longest first
        if world > 1 and n >= world:
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
        ranks = [r for r, _ in tasks]
        first = {r: ranks.index(r) for r in set(ranks)}
        assert all(first[r] < world for r in first)                # every device gets its first batch up front
        flat = np.concatenate([lens[s] for s in shards]) if n else np.zeros(0, int)
        assert np.array_equal(flat[gather_order(shards)], lens)


def test_gl_hand_issued_lds_reads_are_waited_for(tmp_path):
    """gl_iter2_kernel issues its exchange reads as inline asm, which the compiler's wait-count insertion does not see;
    the wait is written into GL2_LAND. Compile the source to assembly and check that nothing touches a register between
    its hand-issued read and that wait (tools/check_asm_reads.py)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc here')
    asm = tmp_path / 'gl.s'
    src = os.path.join(ROOT, 'tacotron_amd', 'csrc', 'griffinlim.hip')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-S', '--cuda-device-only', '-o',
                    str(asm), src], check=True, capture_output=True, timeout=600)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_asm_reads.py'), str(asm)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    m = re.search(r'checked[^:]*: (\d+), violations: 0', r.stdout)
    assert m and int(m.group(1)) >= 300, r.stdout[-500:]
