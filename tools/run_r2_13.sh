set -o pipefail
python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_t13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t13.log; tail -3 gpurun_out/r2_t13.log
grep -q "rc=0" gpurun_out/r2_t13.log || exit 1
timeout -k 10 120 python tools/gemm_bench.py 20 || exit 1
timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['p50_batch_latency_ms'], d['kernels']['conv_gemm'])" || exit 1
