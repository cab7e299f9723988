set -o pipefail
python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t7.log; tail -5 gpurun_out/r2_t7.log
grep -q "rc=0" gpurun_out/r2_t7.log || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace2 -- python3 tools/one_step.py 3 > gpurun_out/trace2.log 2>&1; f=$(find gpurun_out/trace2 -name "*kernel_trace.csv" | head -1); python3 tools/trace_list.py $f "conv_gemm\|cbhg" 99 | tail -3; python3 tools/trace_list.py $f cbhg_tail 2
TACO_NO_CBHG_TAIL=1 timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('unfused', d['value'], d['p50_batch_latency_ms'], d['kernels']['conv_gemm'])"
timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused', d['value'], d['p50_batch_latency_ms'], d['kernels']['conv_gemm'])"
