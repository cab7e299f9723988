set -o pipefail
O=gpurun_out/r03; mkdir -p $O
for st in 8 12 16; do
timeout -k 10 300 python bench.py --to-wav --decode-mode 3 --streams $st --no-cpu-baseline --no-extras --no-roofline --steps 96 --warmup $st > $O/bench_tile_wav$st.json 2> $O/bench_tile_wav$st.err || { tail -5 $O/bench_tile_wav$st.err; exit 1; }
python3 -c "
import json
d = json.loads([l for l in open('$O/bench_tile_wav$st.json') if l.startswith('{')][-1])
print('mode3 to_wav streams $st: frames/s', d['value'], 'ms/step', d['ms_per_step'], 'lat', d['p50_batch_latency_ms'])"
done
timeout -k 10 300 python bench.py --to-wav --decode-mode 2 --streams 6 --no-cpu-baseline --no-extras --no-roofline --steps 96 > $O/bench_m2_wav.json 2> $O/bench_m2_wav.err || exit 1
python3 -c "
import json
d = json.loads([l for l in open('$O/bench_m2_wav.json') if l.startswith('{')][-1])
print('mode2 to_wav streams 6: frames/s', d['value'], 'ms/step', d['ms_per_step'], 'lat', d['p50_batch_latency_ms'])"
