#!/usr/bin/env python
"""bench.py - headline benchmark of the MI355X Tacotron synthesis path.

Metric (BASELINE.json): mel frames/sec, whole job. Workload at N=1 = BASELINE.json configs[1]:
batched inference B=32 per GPU, synthetic 150-character inputs, fp32, LJSpeech default hparams
(r=5, max_iters=200 -> 1000 mel frames per utterance), outputs mel + linear spectrogram.
A "step" is one pass of the hot path (taco_synthesize: encode -> 200-step decode -> post-net)
over one batch whose ids are already resident in HBM. With N>1 every rank runs its own shard of
32 utterances (weak scaling, no collective on the data path); value = all ranks' frames / max-rank time.

Extra objects on the JSON line:
  roofline     - the kernel family with the largest share of the step (the decode loop): its launches are timed
                 with HIP event pairs on each engine's stream over the timed region itself (taco_timing_begin/end);
                 algorithmic bytes per launch from SURVEY.md 8(d) / DESIGN.md. `kernels` holds every family of one
                 batch run alone in a separate instrumented pass.
  cpu_baseline - the fp32 numpy oracle ("port") timed on this host on a bounded sample of the same workload.
  kernels      - per-family launches / ms of one instrumented step (all families).
  rooflines    - one entry per family north_star asks evidence for: conv_gemm (fp32 MFMA TFLOP/s / 157.3),
                 bigru_scan, dec_persistent and gl_iter (algorithmic GB/s / 8000), one batch run alone.
  to_wav       - the same job through the 60-iteration Griffin-Lim (BASELINE.json configs[2], per GPU), timed with
                 the same barrier / max-over-ranks protocol (`mel_linear` instead when --to-wav makes that the headline).
  latency      - p50 wall time of one utterance (B=1) and of one batch alone, to mel+linear and to the waveform.
  long_form    - BASELINE.json configs[3] (T_in=400, max_iters=400: 2000 frames per utterance) at B=1 and B=32: frames/s,
                 us per decoder step, and the attention-bound roofline of its decode loop on (W_step + B*2*400*1024) B/step.
  chain        - the decoder's dependency-chain roofline: S steps x 11 dependent hand-offs x the all-to-all hand-off time
                 measured in this process by taco_debug_handoff_probe; the launch and the B=1 latency as fractions of it.
  gpu_max_hw_queues, value_at_inherited_hw_queues - the hardware-queue setting the figure was taken at, and the
                 headline leg repeated (child process) at the setting this process inherited.
"""
import argparse
import json
import os
import subprocess
import sys
import time

# More than three independent streams only overlap when the process may open more hardware queues than HIP's
# default of 4 (measured: 4 streams 5.0M -> 6.2M frames/s with 8). Must be set before the HIP runtime loads.
# The inherited value is kept when --keep-hw-queues is given; both are recorded on the JSON line, and at N=1 the
# headline leg is repeated in a child process at the inherited setting (`value_at_inherited_hw_queues`).
HWQ_INHERITED = os.environ.get('GPU_MAX_HW_QUEUES')


def _argv_int(flag, default):
    for i, a in enumerate(sys.argv):
        if a == flag and i + 1 < len(sys.argv):
            return int(sys.argv[i + 1])
        if a.startswith(flag + '='):
            return int(a.split('=', 1)[1])
    return default


# one hardware queue per stream in flight (at least 8, at most 32)
HWQ_WANTED = max(8, min(32, _argv_int('--streams', 6)))
if '--keep-hw-queues' not in sys.argv and int(HWQ_INHERITED or 0) < HWQ_WANTED:
    os.environ['GPU_MAX_HW_QUEUES'] = str(HWQ_WANTED)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TF = 157.3  # dense fp32 matrix peak (= vector peak)


def conv_gemm_flops(hp, B, T_in, S):
    """Exact MACs*2 of every conv_gemm launch of one step (encoder, keys, post-net, linear dense)."""
    r, M = hp.outputs_per_step, hp.num_mels
    T = S * r

    def cbhg(rows, cin, K, bank, p1, p2, hw_depth, n_hw, gru, width):
        f = 0
        for k in range(1, K + 1):
            f += 2 * rows * k * cin * bank
        f += 2 * rows * width * K * bank * p1
        f += 2 * rows * width * p1 * p2
        if p2 != hw_depth:
            f += 2 * rows * p2 * hw_depth
        f += n_hw * 2 * rows * hw_depth * 2 * hw_depth
        f += 2 * rows * hw_depth * 6 * gru
        return f

    enc = cbhg(B * T_in, hp.prenet_depths[1], hp.encoder_K, hp.encoder_bank_num_filters,
               hp.encoder_proj_num_filters[0], hp.encoder_proj_num_filters[1], hp.encoder_highway_depth,
               hp.encoder_num_highway_layers, hp.encoder_gru_num_cells, hp.encoder_proj_filter_width)
    keys = 2 * B * T_in * 2 * hp.encoder_gru_num_cells * hp.attention_depth
    post = cbhg(B * T, M, hp.decoder_K, hp.decoder_bank_num_filters, hp.decoder_proj_num_filters[0],
                hp.decoder_proj_num_filters[1], hp.decoder_highway_depth, hp.decoder_num_highway_layers,
                hp.decoder_gru_num_cells, hp.decoder_proj_filter_width)
    lin = 2 * B * T * 2 * hp.decoder_gru_num_cells * hp.num_freq
    return enc + keys + post + lin


def decoder_step_bytes(hp, B, T_in):
    """W_step (all decoder-loop parameters, fp32) + per-utterance keys+values."""
    from tacotron_amd.weights import variable_specs
    w = sum(int(np.prod(s)) for n, s, _ in variable_specs(hp) if n.startswith('decoder/')) * 4
    attn = B * 2 * T_in * 2 * hp.encoder_gru_num_cells * 4
    return w, attn


def family_work(hp, B, T_in, S):
    """Algorithmic work of ONE step per kernel family: (bound, amount) with amount in bytes or flops."""
    r = hp.outputs_per_step
    T = S * r
    w_step, attn = decoder_step_bytes(hp, B, T_in)
    v_bytes = hp.attention_depth * 4
    G = hp.encoder_gru_num_cells
    rows = B * T_in + B * T
    _, hop, win = hp.stft_params()
    L = (T - 1) * hop + win
    return {
        'conv_gemm': ('mfma', conv_gemm_flops(hp, B, T_in, S)),
        'dec_gemm': ('hbm', (w_step - v_bytes) * S),
        'dec_attention': ('hbm', (attn + v_bytes) * S),
        'decode_graph': ('hbm', (w_step + attn) * S),
        # roofline.achieved = (W_step + B*attn_bytes) * S / t_decode; one launch runs all S steps
        'dec_persistent': ('hbm', (w_step + attn) * S),
        'dec_tile': ('hbm', (w_step + attn) * S),
        # scan: reads the hoisted input projections, writes the outputs, loads its recurrent weights once per WG
        'bigru_scan': ('hbm', rows * (6 * G + 2 * G) * 4 + 2 * B * 2 * (3 * G * G) * 4),
        'highway_combine': ('hbm', 4 * rows * (2 * 128 + 128 + 128) * 4),
        'embed_gather': ('hbm', B * T_in * 128 * 4 * 2),
        'dec_finalize': ('hbm', 2 * (B * T * hp.num_mels + B * T_in * S) * 4),
        'gl_mag': ('hbm', 2 * B * T * hp.num_freq * 4),
        'gl_iter': ('hbm', (hp.griffin_lim_iters + 1) * B * (T * hp.num_freq + 2 * L) * 4),
        'gl_ola': ('hbm', B * (T * win + L) * 4),
    }


def cpu_baseline(hp, weights, T_in, S, budget_s=15.0):
    """fp32 numpy restatement (oracle/taco_oracle.py) on this host's cores: full hot path
    (encode -> decode -> post-net) on a bounded sample of the workload."""
    from oracle import taco_oracle as O
    from tacotron_amd.data.synthetic import synth_ids
    try:
        from threadpoolctl import threadpool_info
        thr = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        thr = os.cpu_count() or 1
    w32 = O.cast_weights(weights, np.float32)
    B = 8
    ids, lens = synth_ids(np.random.default_rng(0), B, T_in, ragged=False)
    run = lambda: O.postnet(O.decode(O.encode(ids, lens, w32, hp), w32, hp, max_iters=S)[0], w32, hp)
    run()  # warm-up (BLAS thread pools, page faults)
    n, t0 = 0, time.time()
    while True:
        run()
        n += 1
        el = time.time() - t0
        if el > budget_s:
            break
    frames = n * B * S * hp.outputs_per_step
    return {'value': frames / el, 'unit': 'mel frames/s', 'cores': int(thr), 'kind': 'port',
            'sample': '%d x (B=%d utterances, T_in=%d, %d decoder steps, mel+linear), fp32 numpy oracle, %.1f s'
                      % (n, B, T_in, S, el)}


def under_profiler():
    """rocprofv3 (and anything else that preloads a tool library) has initialised the GPU before main() runs: such a
    process must not start another program (fork + exec). The legs that need a child process are skipped then."""
    env = os.environ
    if any(k.startswith(('ROCPROFILER_', 'ROCP_', 'ROCPROF_')) for k in env):
        return True
    return any(t in env.get('LD_PRELOAD', '') for t in ('rocprof', 'roctracer', 'rocprofiler'))


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nproc-per-node N` as a
    CHILD process, before this process has touched the GPU, relay rank 0's JSON line and leave with the child's code."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    cp = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [l for l in cp.stdout.splitlines() if l.startswith('{')]
    for l in cp.stdout.splitlines():
        if not l.startswith('{'):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1])
    raise SystemExit(cp.returncode if cp.returncode else (0 if lines else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--batch', type=int, default=32, help='utterances per GPU')
    ap.add_argument('--t-in', type=int, default=150)
    ap.add_argument('--max-iters', type=int, default=200)
    ap.add_argument('--to-wav', action='store_true', help='include the 60-iteration Griffin-Lim (configs[2])')
    ap.add_argument('--streams', type=int, default=6,
                    help='independent batches in flight per GPU: step i runs on engine/stream i %% streams '
                         '(a serving loop keeps several batches in flight; 1 = strictly one batch at a time)')
    ap.add_argument('--decode-mode', type=int, default=2,
                    help='0 eager per-stage kernels, 1 hipGraph, 2 persistent kernel (32 workgroups per 16 utterances: '
                         'latency), 3 tile kernel (one workgroup per 16 utterances: throughput, wants --streams >= 12)')
    ap.add_argument('--bigru-mode', type=int, default=0,
                    help='CBHG BiGRU scan in the throughput legs: 0 = one utterance per workgroup, 1 / 2 = sixteen / four on MFMA')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the secondary legs (to-waveform rate, B=1 latencies, per-family rooflines, child run)')
    ap.add_argument('--keep-hw-queues', action='store_true', help='do not raise GPU_MAX_HW_QUEUES to 8')
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        if under_profiler():
            raise SystemExit('bench.py --gpus %d under a profiler: launch the ranks with torch.distributed.run yourself '
                             '(a process whose GPU the profiler has initialised must not start another)' % args.gpus)
        self_launch(args)
    if int(os.environ.get('WORLD_SIZE', 1)) != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%s: launch with --nproc-per-node %d'
                         % (args.gpus, os.environ.get('WORLD_SIZE', '1 (unset)'), args.gpus))

    inherited_leg = None
    profiled = under_profiler()
    if profiled and not args.no_extras and not args.keep_hw_queues:
        inherited_leg = {'skipped': 'profiler'}          # (no child process from a process the profiler has initialised)
    if (int(os.environ.get('RANK', 0)) == 0 and int(os.environ.get('WORLD_SIZE', 1)) == 1 and not args.no_extras and not args.keep_hw_queues and
            not profiled and os.environ.get('GPU_MAX_HW_QUEUES') != HWQ_INHERITED):
        # The same headline leg at the hardware-queue count this process inherited (the GPU box exports 4), in a child
        # process that runs to its end BEFORE this process touches the GPU: a process that has initialised the GPU
        # must not start (fork + exec) another program on this pool.
        env = dict(os.environ)
        if HWQ_INHERITED is None:
            env.pop('GPU_MAX_HW_QUEUES', None)
        else:
            env['GPU_MAX_HW_QUEUES'] = HWQ_INHERITED
        cmd = [sys.executable, os.path.abspath(__file__), '--keep-hw-queues', '--no-extras', '--no-roofline',
               '--no-cpu-baseline', '--steps', str(args.steps), '--warmup', str(args.warmup), '--batch', str(args.batch),
               '--t-in', str(args.t_in), '--max-iters', str(args.max_iters), '--streams', str(max(1, args.streams)),
               '--decode-mode', str(args.decode_mode)] + (['--to-wav'] if args.to_wav else [])
        cp = None
        try:
            cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
            line = [l for l in cp.stdout.splitlines() if l.startswith('{')][-1]
            child = json.loads(line)
            inherited_leg = {'value': child['value'], 'ms_per_step': child['ms_per_step'],
                             'gpu_max_hw_queues': HWQ_INHERITED}
        except Exception as ex:                    # reported, never fatal for the headline line
            inherited_leg = {'error': repr(ex)[:200], 'rc': cp.returncode if cp is not None else None,
                             'stdout_tail': (cp.stdout[-300:] if cp is not None else None),
                             'stderr_tail': (cp.stderr[-400:] if cp is not None else None)}

    import torch
    import torch.distributed as dist
    from tacotron_amd.engine import TacoEngine
    from tacotron_amd.hparams import HParams
    from tacotron_amd.weights import init_weights
    from tacotron_amd.data.synthetic import synth_ids

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world > 1:
        # one rank per GPU over RCCL. TACO_BENCH_BACKEND=gloo (rehearsal of the multi-rank path on a box with fewer
        # GPUs than ranks: the ranks then share devices round-robin) is for testing only.
        backend = os.environ.get('TACO_BENCH_BACKEND', 'nccl')
        ndev = torch.cuda.device_count()
        if backend == 'nccl' and local >= ndev:
            raise SystemExit('rank %d: LOCAL_RANK %d but only %d GPUs visible' % (rank, local, ndev))
        if world > max(ndev, 1):
            # rehearsal only: ranks share a GPU, and the persistent decoder's CU budget is per process - keep the
            # decodes in flight on one device within the chip (4 x 64 CUs) so that they cannot wait on each other
            per_gpu = -(-world // max(ndev, 1))
            args.streams = max(1, min(args.streams, 4 // per_gpu))
        local = local % max(ndev, 1)
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    hp = HParams()
    hp.max_iters = args.max_iters
    weights = init_weights(hp, seed=1234)          # "random-init weights" (BASELINE.json configs)
    engines = [TacoEngine(weights, hp, device=local) for _ in range(max(1, args.streams))]
    eng = engines[0]
    torch.cuda.synchronize()
    for e in engines:
        e.set_decode_graph(args.decode_mode)
        e.detached = True          # independent streams; the timed region synchronises the device at both ends
    B, T_in, S = args.batch, args.t_in, args.max_iters
    # this rank's shard of the synthetic job: utterances [rank*B, (rank+1)*B)
    ids_all, lens_all = synth_ids(np.random.default_rng(0), B * world, T_in, ragged=False)
    ids = torch.from_numpy(ids_all[rank * B:(rank + 1) * B]).to(eng.device)
    lens = torch.from_numpy(lens_all[rank * B:(rank + 1) * B]).to(eng.device)
    outs = [{} for _ in engines]
    out = outs[0]
    counter = [0]

    def step(to_wav):
        i = counter[0] % len(engines)
        counter[0] += 1
        engines[i].synthesize(ids, lens, max_iters=S, to_wav=to_wav, out=outs[i])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def p50_latency(e, n_ids, n_lens, to_wav, o, reps):
        """p50 wall time of one synthesis alone on the device (launch to synchronise), ms."""
        lat = []
        for _ in range(reps + 2):
            t0 = time.perf_counter()
            e.synthesize(n_ids, n_lens, max_iters=S, to_wav=to_wav, out=o)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        lat = sorted(lat[2:])                      # (two untimed: workspace growth, first-launch set-up)
        return 1e3 * lat[len(lat) // 2]

    def timed(n_steps, to_wav):
        """n_steps passes over the batch, `streams` in flight, bracketed by barrier + synchronise; max over ranks."""
        counter[0] = 0
        barrier()
        # decodes side by side: as many as fit the chip for mel + linear; one at a time when the vocoder's full-chip
        # launches of the other batches in flight run beside them (include/taco.h, taco_set_persist_budget)
        budget = 32 * ((B + 15) // 16) if to_wav and args.decode_mode == 2 else 0
        if os.environ.get('TACO_BENCH_DECODES'):             # (tuning) decodes allowed side by side, 0 = no limit
            budget = int(os.environ['TACO_BENCH_DECODES']) * 32 * ((B + 15) // 16)
        eng.set_persist_budget(budget)
        for e in engines:
            e.set_bigru_mode(args.bigru_mode)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(to_wav)
        barrier()
        el = time.perf_counter() - t0
        for e in engines:
            e.set_bigru_mode(0)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=eng.device if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(max(args.warmup, len(engines))):
        step(args.to_wav)
    barrier()
    # single-batch latency (one batch in flight), for the p50 figure
    lat_ms = p50_latency(eng, ids, lens, args.to_wav, out, 5)
    # HIP event pairs around every launch of the decode-loop kernel, on each engine's own stream, over the timed
    # region itself (two event records per batch): the roofline's launch duration is measured live, under the
    # same contention as `value`
    loop_fam = {2: 'dec_persistent', 1: 'decode_graph', 3: 'dec_tile'}.get(args.decode_mode)
    if loop_fam and not args.no_roofline:
        for e in engines:
            e.timing_begin(loop_fam)
    el = timed(args.steps, args.to_wav)
    live_n, live_ms = 0, 0.0
    if loop_fam and not args.no_roofline:
        for e in engines:
            n_, ms_ = e.timing_end()
            live_n += n_
            live_ms += ms_
    # a persistent decode whose bounded wait ran out would have produced garbage at full speed: fail loudly
    for e in engines:
        e.decode_status()
    steps_run = int(out['steps'].item())
    frames_per_step = world * B * steps_run * hp.outputs_per_step
    value = frames_per_step * args.steps / el

    def workload(to_wav):
        return ('BASELINE.json configs[%d]%s: batched inference B=%d per GPU, %d-char inputs, fp32, max_iters=%d r=%d, '
                'outputs mel+linear%s' % (2 if to_wav else 1, ' per GPU' if to_wav else '', B, T_in, S,
                                          hp.outputs_per_step, ' + %d-iteration Griffin-Lim waveform' %
                                          hp.griffin_lim_iters if to_wav else ''))

    res = {
        'metric': 'mel frames/sec (whole job)', 'value': value, 'unit': 'mel frames/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * el / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic (seeded character ids, random-init weights with TF default initialisers)',
        'config': {'workload': workload(args.to_wav),
                   'batch_per_gpu': B, 'global_batch': B * world, 'T_in': T_in, 'max_iters': S,
                   'frames_per_utterance': steps_run * hp.outputs_per_step, 'parallelism': 'utterance-sharded x%d' % world,
                   'streams_per_gpu': len(engines)},
        'p50_batch_latency_ms': lat_ms, 'batches_in_flight': len(engines),
        'gpu_max_hw_queues': {'effective': os.environ.get('GPU_MAX_HW_QUEUES'), 'inherited': HWQ_INHERITED},
        'decode_status': 'ok',
    }

    # ---- secondary legs (BASELINE.json's metric also names the to-waveform rate and B=1 / batch p50 latencies)
    if not args.no_extras:
        other = not args.to_wav
        n2 = max(12, min(args.steps, 60 if other else 200))
        for _ in range(len(engines)):
            step(other)                            # (workspace / output buffers of the other leg)
        el2 = timed(n2, other)
        for e in engines:
            e.decode_status()
        res['to_wav' if other else 'mel_linear'] = {
            'value': frames_per_step * n2 / el2, 'unit': 'mel frames/s', 'ms_per_step': 1e3 * el2 / n2, 'steps': n2,
            'workload': workload(other)}
        if rank == 0:
            o1 = {}
            res['latency'] = {
                'unit': 'ms, p50 of 9 (launch to synchronise, one synthesis alone on the GPU)',
                'b1_mel_linear': p50_latency(eng, ids[:1], lens[:1], False, o1, 9),
                'b1_to_wav': p50_latency(eng, ids[:1], lens[:1], True, o1, 9),
                'batch_mel_linear': lat_ms if not args.to_wav else p50_latency(eng, ids, lens, False, out, 5),
                'batch_to_wav': lat_ms if args.to_wav else p50_latency(eng, ids, lens, True, out, 5),
                'batch': B}
            # a batch that has the GPU to itself: its vocoder as two utterance groups on two streams (taco_set_gl_split)
            eng.set_gl_split(2)
            res['latency']['batch_to_wav_gl_split2'] = p50_latency(eng, ids, lens, True, out, 5)
            eng.set_gl_split(1)
            # BASELINE.json configs[4]: B=1 with the per-stage step chain captured into a hipGraph (decode mode 1)
            eng.set_decode_graph(1)
            res['latency']['b1_hipgraph_mel_linear'] = p50_latency(eng, ids[:1], lens[:1], False, o1, 9)
            res['latency']['b1_hipgraph_to_wav'] = p50_latency(eng, ids[:1], lens[:1], True, o1, 9)
            eng.set_decode_graph(args.decode_mode)

    if rank == 0 and not args.no_roofline:
        work = family_work(hp, B, T_in, S)
        kinds = eng.kernel_kinds()
        fam = {}
        inst_wav = args.to_wav or not args.no_extras      # the instrumented passes include the vocoder's families

        def one(k):
            # one batch alone on engine 0's stream, with HIP event pairs around every launch of family k
            eng.timing_begin(k)
            eng.synthesize(ids, lens, max_iters=S, to_wav=inst_wav, out=out)
            n, ms = eng.timing_end()
            if n:
                fam[k] = {'launches': n, 'ms': ms, 'avg_us': 1e3 * ms / n}

        # families of the decode mode that is NOT the one benchmarked would only run because asking for their
        # timing forces the per-stage path: leave them out
        skip = ({'dec_gemm', 'dec_attention', 'decode_graph', 'dec_tile'} if args.decode_mode == 2 else
                {'dec_gemm', 'dec_attention', 'decode_graph', 'dec_persistent'} if args.decode_mode == 3 else
                {'dec_persistent', 'dec_tile'})
        for k in kinds:
            if k not in skip and k != 'decode_graph':
                one(k)
        if 'decode_graph' not in skip and args.decode_mode == 1:
            one('decode_graph')
        res['kernels'] = fam
        gl_fams = {'gl_mag', 'gl_iter', 'gl_ola'}
        leaf = {k: v for k, v in fam.items() if k != 'decode_graph' and (args.to_wav or k not in gl_fams)}
        dom = max(leaf, key=lambda k: leaf[k]['ms'])

        def roof(k, sec_per_launch, launches):
            bound, amount = work[k]
            per_launch = amount / launches
            if bound == 'hbm':
                ach, peak, unit = per_launch / sec_per_launch / 1e9, HBM_PEAK_GBS, 'GB/s'
            else:
                ach, peak, unit = per_launch / sec_per_launch / 1e12, FP32_MFMA_PEAK_TF, 'TFLOP/s'
            return {'kernel': k, 'bound': bound, 'achieved': ach, 'peak': peak, 'unit': unit, 'frac': ach / peak,
                    'launches_per_step': launches, 'avg_launch_us': 1e6 * sec_per_launch,
                    'algorithmic_per_launch': per_launch}

        sec = leaf[dom]['ms'] * 1e-3 / leaf[dom]['launches']
        solo_us = 1e6 * sec
        if dom == loop_fam and live_n:
            sec = live_ms * 1e-3 / live_n          # average over the launches of the timed region
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json'))).get('B%d_Tin%d_S%d' % (B, T_in, S), {})
        except Exception:
            tj = {}
        traffic = tj.get(dom, {}).get('bytes_per_launch')
        r0 = roof(dom, sec, leaf[dom]['launches'])
        r0.update({'traffic': traffic, 'avg_launch_us_one_batch_in_flight': solo_us,
                   'launches_timed': live_n if dom == loop_fam and live_n else leaf[dom]['launches']})
        res['roofline'] = r0
        # one entry per kernel family north_star asks evidence for (MFMA utilisation of the CBHG / post-net GEMMs,
        # achieved GB/s of the decoder loop, the BiGRU scan and the vocoder): one batch alone, HIP events per launch
        res['rooflines'] = [dict(roof(k, fam[k]['ms'] * 1e-3 / fam[k]['launches'], fam[k]['launches']),
                                 traffic=tj.get(k, {}).get('bytes_per_launch'),
                                 measured='one batch alone, HIP event pair per launch')
                            for k in ('conv_gemm', 'bigru_scan', 'dec_persistent', 'dec_tile', 'decode_graph', 'gl_iter')
                            if k in fam and k in work]
        loop = next((k for k in ('dec_persistent', 'dec_tile', 'decode_graph') if k in fam), None)
        if loop:
            b, amt = work[loop]
            s2 = fam[loop]['ms'] * 1e-3
            res['decode_loop'] = {'kernel': loop, 'ms': fam[loop]['ms'], 'us_per_decoder_step': 1e6 * s2 / S,
                                  'us_per_decoder_step_live': (1e3 * live_ms / live_n / S) if live_n else None,
                                  'achieved_GBps': amt / s2 / 1e9, 'frac_of_hbm': amt / s2 / 1e9 / HBM_PEAK_GBS}
        r0['traffic_source'] = 'profiles/traffic.json (static: FETCH_SIZE / WRITE_SIZE from separate rocprofv3 --pmc passes)'
        try:
            busy = json.load(open(os.path.join(ROOT, 'profiles', 'mfma_busy.json')))
            for r in res['rooflines']:
                if r['kernel'] in busy:
                    r['counters'] = busy[r['kernel']]     # (static: SQ counter passes of profiles/r03, see profiles/README.md)
        except Exception:
            pass

    if rank == 0 and not args.no_roofline and not args.no_extras and args.decode_mode == 2:
        # ---- dependency-chain roofline of the decoder: a decoder step is 11 dependent all-to-all
        # hand-offs among a tile's 32 workgroups; the price of one such hand-off with nothing else in it is measured
        # here, in this process, on the idle chip (taco_debug_handoff_probe: 32 workgroups, 4000 dependent rounds)
        torch.cuda.synchronize()
        t_h = min(eng.handoff_probe(4000) for _ in range(3))
        EDGES = 11
        chain_ms = 1e-3 * S * EDGES * t_h

        def decode_ms(n_ids, n_lens, o, s_iters):
            eng.timing_begin('dec_persistent')
            eng.synthesize(n_ids, n_lens, max_iters=s_iters, to_wav=False, out=o)
            n, ms = eng.timing_end()
            return ms / max(n, 1)
        o1 = {}
        eng.synthesize(ids[:1], lens[:1], max_iters=S, to_wav=False, out=o1)
        b1_dec = min(decode_ms(ids[:1], lens[:1], o1, S) for _ in range(3))
        dl = res.get('decode_loop', {})
        res['chain'] = {
            'bound': 'dependency chain', 'handoff_us': t_h, 'edges_per_step': EDGES, 'steps': S,
            'chain_ms_per_launch': chain_ms, 'chain_us_per_step': EDGES * t_h,
            'measured': 'taco_debug_handoff_probe: 32 workgroups (one per CU), 4000 dependent all-to-all rounds of '
                        '32 x 1 KB {value, tag} granules, idle chip, best of 3',
            'launch_ms_alone': dl.get('ms'), 'frac_alone': (chain_ms / dl['ms']) if dl.get('ms') else None,
            'launch_ms_live': (live_ms / live_n) if live_n else None,
            'frac_live': (chain_ms * live_n / live_ms) if live_n else None,
            'b1_decode_ms': b1_dec, 'b1_frac': chain_ms / b1_dec if b1_dec else None,
            'b1_latency_mel_linear_ms': res.get('latency', {}).get('b1_mel_linear')}

        # ---- BASELINE.json configs[3]: long-form, T_in = 400 characters, max_iters = 400 (2000 frames per utterance)
        T2, S2 = 400, 400
        ids2, lens2 = synth_ids(np.random.default_rng(3), B, T2, ragged=False)
        ids2 = torch.from_numpy(ids2).to(eng.device)
        lens2 = torch.from_numpy(lens2).to(eng.device)
        w_step, attn2 = decoder_step_bytes(hp, B, T2)
        _, attn2_1 = decoder_step_bytes(hp, 1, T2)
        lf = {'workload': 'BASELINE.json configs[3]: long-form, T_in=%d, max_iters=%d (%d frames per utterance), fp32, '
                          'mel+linear' % (T2, S2, S2 * hp.outputs_per_step),
              'attention_bytes_per_step': {'b1': w_step + attn2_1, 'batch': w_step + attn2}}
        outs2 = [{} for _ in engines]
        for i, e in enumerate(engines):
            e.synthesize(ids2, lens2, max_iters=S2, to_wav=False, out=outs2[i])      # workspace / buffers
        torch.cuda.synchronize()
        eng.set_persist_budget(0)
        n3 = 4 * len(engines)
        t0 = time.perf_counter()
        for i in range(n3):
            engines[i % len(engines)].synthesize(ids2, lens2, max_iters=S2, to_wav=False, out=outs2[i % len(engines)])
        torch.cuda.synchronize()
        el3 = time.perf_counter() - t0
        for e in engines:
            e.decode_status()
        o2 = {}
        eng.synthesize(ids2[:1], lens2[:1], max_iters=S2, to_wav=False, out=o2)
        d1 = min(decode_ms(ids2[:1], lens2[:1], o2, S2) for _ in range(3))
        dB = min(decode_ms(ids2, lens2, outs2[0], S2) for _ in range(2))

        def lroof(bytes_per_step, ms):
            ach = bytes_per_step * S2 / (ms * 1e-3) / 1e9
            return {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                    'us_per_decoder_step': 1e3 * ms / S2, 'decode_ms': ms,
                    'chain_frac': 1e-3 * S2 * EDGES * t_h / ms}
        lf['batch'] = dict(B=B, value=B * S2 * hp.outputs_per_step * n3 / el3, unit='mel frames/s',
                           ms_per_step=1e3 * el3 / n3, steps=n3, batches_in_flight=len(engines),
                           p50_batch_latency_ms=p50_latency(eng, ids2, lens2, False, outs2[0], 3),
                           decode_roofline=lroof(w_step + attn2, dB))
        lat1 = p50_latency(eng, ids2[:1], lens2[:1], False, o2, 5)
        lf['b1'] = dict(B=1, p50_latency_ms=lat1, value=S2 * hp.outputs_per_step / (lat1 * 1e-3), unit='mel frames/s',
                        p50_latency_to_wav_ms=p50_latency(eng, ids2[:1], lens2[:1], True, o2, 3),
                        decode_roofline=lroof(w_step + attn2_1, d1))
        res['long_form'] = lf

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(hp, weights, T_in, S)
        if not args.no_extras:
            # BASELINE.json configs[0] (synthesizer.py:37-45): one utterance through the 60-iteration Griffin-Lim on the CPU
            from oracle import taco_oracle as O
            w32 = O.cast_weights(weights, np.float32)
            ids1, lens1 = synth_ids(np.random.default_rng(0), 1, T_in, ragged=False)
            t0 = time.time()
            f = O.tacotron_forward(ids1, lens1, w32, hp, dtype=np.float32, max_iters=S)
            O.spectrogram_tensorflow_inv(f['linear'][0], hp, dtype=np.float32)
            res['cpu_baseline']['b1_to_wav_ms'] = 1e3 * (time.time() - t0)
            res['cpu_baseline']['b1_sample'] = ('1 run: B=1, T_in=%d, %d decoder steps, mel+linear, %d Griffin-Lim '
                                                'iterations (BASELINE.json configs[0]), fp32 numpy oracle'
                                                % (T_in, S, hp.griffin_lim_iters))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for e in engines:
        e.close()
    if rank == 0 and inherited_leg is not None:
        res['value_at_inherited_hw_queues'] = inherited_leg
    if rank == 0:
        print(json.dumps(res))


if __name__ == '__main__':
    main()
