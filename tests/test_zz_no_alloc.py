"""taco_griffin_lim_noalloc (include/taco.h): the vocoder entry point that allocates nothing - its working set is a
static device array inside the library - against taco_griffin_lim, which takes a caller workspace. Same kernels, so
the bar is bit-identity. Two calls in a row, the second with a smaller input (the array keeps the first call's
contents); the size limit is enforced with a clear error. Collected last (file name) so that a failure here cannot hide
the other tests from `pytest -x`; passes alone in a fresh process and after the other tests."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_noalloc_griffin_lim_matches_workspace_entry(engine, hp):
    import torch
    from tacotron_amd import _lib
    lib, h = engine.lib, engine.handle
    rng = np.random.default_rng(31)
    dev = engine.device
    s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _, hop, win = hp.stft_params()

    def both(B, T, iters):
        lin = torch.from_numpy(rng.uniform(0.2, 1.0, size=(B, T, hp.num_freq)).astype(np.float32)).to(dev)
        L = (T - 1) * hop + win
        want = torch.empty(B, L, device=dev)
        got = torch.full((B, L), float('nan'), device=dev)
        ws = torch.empty(int(lib.taco_workspace_bytes(h, B, 1, -(-T // hp.outputs_per_step))), dtype=torch.uint8, device=dev)
        assert lib.taco_griffin_lim(h, lin.data_ptr(), B, T, iters, want.data_ptr(), ws.data_ptr(), ws.numel(), s) == 0, _lib.last_error()
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info(dev)
        assert lib.taco_griffin_lim_noalloc(h, lin.data_ptr(), B, T, iters, got.data_ptr(), s) == 0, _lib.last_error()
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info(dev)
        assert free1 == free0                      # the call took no device memory
        assert bool(torch.isfinite(got).all()) and torch.equal(got, want), (B, T, iters)

    both(1, 1000, 60)        # one utterance at the default length, all 60 iterations (the size the limit is cut for)
    both(1, 37, 5)           # right after it: a smaller input on the same array, which still holds the first call's data
    both(3, 300, 4)          # several short utterances within the limit
    both(2, 512, 2)          # exactly the limit: B*T = 1024 frames
    # beyond the limit: refused, nothing launched, message names the limit
    lin = torch.zeros(1, 1025, hp.num_freq, device=dev)
    wav = torch.zeros(1, 1024 * hop + win, device=dev)
    assert lib.taco_griffin_lim_noalloc(h, lin.data_ptr(), 1, 1025, 1, wav.data_ptr(), s) < 0
    assert 'at most 1024' in _lib.last_error()
    assert lib.taco_griffin_lim_noalloc(h, None, 1, 10, 1, wav.data_ptr(), s) < 0
    # the Python face of it
    lin = rng.uniform(0.2, 1.0, size=(40, hp.num_freq)).astype(np.float32)
    assert torch.equal(engine.griffin_lim_noalloc(lin, iters=3), engine.griffin_lim(lin, iters=3))


def test_noalloc_griffin_lim_under_graph_capture(engine, hp):
    """The use the entry point is for: the call is made while a stream is being captured into a hipGraph (nothing may
    allocate, create streams or events, or synchronise there), the graph is replayed twice - the second time on new
    input written into the same buffer - and each replay equals taco_griffin_lim on that input, bit for bit."""
    import torch
    from tacotron_amd import _lib
    lib, h, dev = engine.lib, engine.handle, engine.device
    rng = np.random.default_rng(32)
    _, hop, win = hp.stft_params()
    B, T, iters = 2, 200, 6
    L = (T - 1) * hop + win
    lin = torch.empty(B, T, hp.num_freq, device=dev)
    got = torch.empty(B, L, device=dev)
    want = torch.empty(B, L, device=dev)
    ws = torch.empty(int(lib.taco_workspace_bytes(h, B, 1, -(-T // hp.outputs_per_step))), dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = lib.taco_griffin_lim_noalloc(h, lin.data_ptr(), B, T, iters, got.data_ptr(), s)
    assert rc == 0, _lib.last_error()
    for rep in range(2):
        lin.copy_(torch.from_numpy(rng.uniform(0.2, 1.0, size=(B, T, hp.num_freq)).astype(np.float32)))
        got.fill_(float('nan'))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        s0 = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        assert lib.taco_griffin_lim(h, lin.data_ptr(), B, T, iters, want.data_ptr(), ws.data_ptr(), ws.numel(), s0) == 0
        torch.cuda.synchronize()
        assert bool(torch.isfinite(got).all()) and torch.equal(got, want), rep
