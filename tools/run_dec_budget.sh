set -o pipefail
O=gpurun_out/r03; mkdir -p $O
for cfg in "4 0 6" "3 0 6" "3 1 6" "3 0 7" "2 0 6"; do
  set -- $cfg
  TACO_BENCH_DECODES=$1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --bigru-mode $2 --streams $3 > $O/decb.json 2> $O/decb.err || { tail -5 $O/decb.err; exit 1; }
  python3 -c "
import json
d = json.loads([l for l in open('$O/decb.json') if l.startswith('{')][-1])
print('decodes=$1 bigru_mode=$2 streams=$3 frames/s %.4g ms/step %.3f' % (d['value'], d['ms_per_step']))"
done
