set -o pipefail
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "decode_modes_agree or parity_per_mode or batch_invariance" > $O/tile_tests.log 2>&1; rc=$?
tail -5 $O/tile_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/time_tile.py 32 12 || exit 1
timeout -k 10 300 python bench.py --decode-mode 3 --streams 12 --no-cpu-baseline --no-extras --steps 240 > $O/bench_tile.json 2> $O/bench_tile.err || { tail -5 $O/bench_tile.err; exit 1; }
python3 -c "
import json
d = json.loads([l for l in open('$O/bench_tile.json') if l.startswith('{')][-1])
print('frames/s', d['value'], 'ms/step', d['ms_per_step'], 'lat', d['p50_batch_latency_ms']); print(d.get('roofline')); print(d.get('kernels'))"
