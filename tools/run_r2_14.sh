set -o pipefail
python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_t14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t14.log; tail -5 gpurun_out/r2_t14.log
grep -q "rc=0" gpurun_out/r2_t14.log || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --keep-hw-queues > gpurun_out/bench_budget.json 2>/dev/null || exit 1
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/bench_budget.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['to_wav'], d['latency'])"
