# What a round ends with, on the GPU box (through gpurun): the GPU tests, the smoke entry, a short bench line.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100 > gpurun_out/bench_short.json 2> gpurun_out/bench_short.err || exit 1
python3 -c "
import json
d = json.loads([l for l in open('gpurun_out/bench_short.json') if l.startswith('{')][-1])
print('frames/s', d['value'], 'ms/step', d['ms_per_step'], 'to_wav', d['to_wav']['value'], d['to_wav']['ms_per_step'])
print('latency', d['latency']); print('decode', d['decode_loop']); print('hwq4', d.get('value_at_inherited_hw_queues'))
for r in d['rooflines']: print(r['kernel'], r['achieved'], r['unit'], r['frac'])"
