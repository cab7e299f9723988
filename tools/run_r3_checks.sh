# Round-3 check set (GPU box, through gpurun): GPU tests, smoke, the full default bench line, the product sharded call.
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests.log 2>&1; rc=$?
tail -3 $O/gpu_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
timeout -k 10 600 python bench.py > $O/bench_r3.json 2> $O/bench_r3.err || { tail -20 $O/bench_r3.err; exit 1; }
python3 -c "
import json
d = json.loads([l for l in open('$O/bench_r3.json') if l.startswith('{')][-1])
print('frames/s', d['value'], 'ms/step', d['ms_per_step'], 'to_wav', d['to_wav']['value'], d['to_wav']['ms_per_step'])
print('latency', d['latency']); print('decode', d['decode_loop']); print('hwq4', d.get('value_at_inherited_hw_queues'))
print('chain', d.get('chain')); print('long_form', d.get('long_form')); print('cpu', d.get('cpu_baseline'))
for r in d['rooflines']: print(r['kernel'], r['achieved'], r['unit'], r['frac'])"
timeout -k 10 300 python tools/time_sharded.py > $O/sharded.json 2> $O/sharded.err || { tail -20 $O/sharded.err; exit 1; }
cat $O/sharded.json
