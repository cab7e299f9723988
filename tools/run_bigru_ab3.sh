set -o pipefail
O=gpurun_out/r03; mkdir -p $O
for m in 0 2 1 0 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --bigru-mode $m > $O/bigru_ab.json 2> $O/bigru_ab.err || { tail -5 $O/bigru_ab.err; exit 1; }
  python3 -c "
import json
d = json.loads([l for l in open('$O/bigru_ab.json') if l.startswith('{')][-1])
print('bigru_mode=$m frames/s %.4g ms/step %.3f' % (d['value'], d['ms_per_step']))"
done
