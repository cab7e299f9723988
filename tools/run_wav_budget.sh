# to-waveform throughput against the number of persistent decodes allowed side by side (one box)
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
for d in 1 2 1 2; do
  TACO_BENCH_DECODES=$d timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --to-wav --steps 60 > $O/wavb.json 2> $O/wavb.err || { tail -5 $O/wavb.err; exit 1; }
  python3 -c "
import json
d = json.loads([l for l in open('$O/wavb.json') if l.startswith('{')][-1])
print('to_wav decodes=$d frames/s %.4g ms/step %.3f' % (d['value'], d['ms_per_step']))"
done
