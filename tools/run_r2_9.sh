set -o pipefail
python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "griffin or gl or wav or no_alloc" > gpurun_out/r2_t9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t9.log; tail -3 gpurun_out/r2_t9.log
grep -q "rc=0" gpurun_out/r2_t9.log || exit 1
for b in 1 2 4 6 8 12 16 32; do timeout -k 10 120 python tools/gl_bench.py $b 60 || exit 1; done
