set -o pipefail
O=gpurun_out/r03; mkdir -p $O
for cfg in "0 6" "3 6" "3 8" "0 8" "2 8" "3 10"; do
set -- $cfg
TACO_BENCH_DECODES=$1 timeout -k 10 300 python bench.py --streams $2 --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/sweep.json 2> $O/sweep.err || { tail -5 $O/sweep.err; exit 1; }
python3 -c "
import json
d = json.loads([l for l in open('$O/sweep.json') if l.startswith('{')][-1])
print('decodes=$1 streams=$2: frames/s %.3fM ms/step %.3f' % (d['value']/1e6, d['ms_per_step']))"
done
