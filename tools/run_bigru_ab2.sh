set -o pipefail
O=gpurun_out/r03; mkdir -p $O
for st in 8 7; do
for m in 0 1; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --bigru-mode $m --streams $st > $O/bigru_ab.json 2> $O/bigru_ab.err || { tail -5 $O/bigru_ab.err; exit 1; }
  python3 -c "
import json
d = json.loads([l for l in open('$O/bigru_ab.json') if l.startswith('{')][-1])
print('streams=$st bigru_mode=$m frames/s %.4g ms/step %.3f p50 batch latency %.2f' % (d['value'], d['ms_per_step'], d['p50_batch_latency_ms']))"
done; done
TACO_BIGRU_MODE=2 timeout -k 10 200 python tools/time_families.py 2>&1 | tail -8
