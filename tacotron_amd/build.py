"""Builds tacotron_amd/libtaco_hip.so (gfx950) from tacotron_amd/csrc/*.hip with hipcc.

In-tree and explicit (no JIT cache): the shared object travels to the GPU box with the
source snapshot. `python -m tacotron_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', 'build')
LIB = os.path.join(HERE, 'libtaco_hip.so')
SOURCES = ['api.hip', 'gemm.hip', 'rnn.hip', 'decoder.hip', 'dec_persistent.hip', 'dec_tile.hip', 'griffinlim.hip']
HEADERS = ['kernels.h', 'fft_core.h', 'dec_common.h', os.path.join('..', '..', 'include', 'taco.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            return o, True
        return o, False

    with ThreadPoolExecutor(max_workers=min(7, len(SOURCES))) as ex:
        res = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
