set -o pipefail
O=gpurun_out/r03; mkdir -p $O
for lib in "" tacotron_amd/libtaco_v_nobigru.so; do
TACO_LIB=$lib timeout -k 10 300 python tools/bench_with_lib.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bound.json 2> $O/bound.err || { tail -5 $O/bound.err; exit 1; }
python3 -c "
import json
d = json.loads([l for l in open('$O/bound.json') if l.startswith('{')][-1])
print('lib=[$lib] frames/s', d['value'], 'ms/step', d['ms_per_step'])"
done
