#!/bin/bash
# usage: tools/sweep.sh "<modes>" "<streams>" [extra bench args]
mkdir -p gpurun_out
for m in $1; do for n in $2; do
  timeout -k 10 200 python -u bench.py --decode-mode $m --streams $n --steps 24 --no-cpu-baseline --no-roofline $3 > gpurun_out/b.log 2>&1 || { tail -5 gpurun_out/b.log; exit 1; }
  tail -1 gpurun_out/b.log | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('mode $m streams', r['config']['streams_per_gpu'], int(r['value']), 'frames/s', round(r['ms_per_step'],2), 'ms/step  latency', round(r['p50_batch_latency_ms'],2))"
done; done
