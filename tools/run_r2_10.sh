set -o pipefail
python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_t10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t10.log; tail -3 gpurun_out/r2_t10.log
grep -q "rc=0" gpurun_out/r2_t10.log || exit 1
for t in 128x128 default; do
if [ $t = default ]; then unset TACO_GEMM_TILE; else export TACO_GEMM_TILE=$t; fi
timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['p50_batch_latency_ms'], d['kernels']['conv_gemm'])" || exit 1
done
