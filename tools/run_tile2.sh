set -o pipefail
O=gpurun_out/r03; mkdir -p $O
export GPU_MAX_HW_QUEUES=32
timeout -k 10 200 python tools/time_tile.py 32 12 || exit 1
timeout -k 10 200 python tools/time_tile.py 32 30 || exit 1
timeout -k 10 300 python bench.py --decode-mode 3 --streams 30 --no-cpu-baseline --no-extras --no-roofline --steps 240 --warmup 30 > $O/bench_tile.json 2> $O/bench_tile.err || { tail -5 $O/bench_tile.err; exit 1; }
python3 -c "
import json
d = json.loads([l for l in open('$O/bench_tile.json') if l.startswith('{')][-1])
print('frames/s', d['value'], 'ms/step', d['ms_per_step'], 'lat', d['p50_batch_latency_ms'])"
