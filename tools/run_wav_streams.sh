# to-waveform throughput against the number of batches in flight (one box)
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
for st in 3 4 5 6 7; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --to-wav --steps 60 --streams $st > $O/wavst.json 2> $O/wavst.err || { tail -5 $O/wavst.err; exit 1; }
  python3 -c "
import json
d = json.loads([l for l in open('$O/wavst.json') if l.startswith('{')][-1])
print('to_wav streams=$st frames/s %.4g ms/step %.3f' % (d['value'], d['ms_per_step']))"
done
