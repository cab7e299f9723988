set -o pipefail
python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "griffin or gl or wav or no_alloc" > gpurun_out/r2_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t8.log; tail -5 gpurun_out/r2_t8.log
grep -q "rc=0" gpurun_out/r2_t8.log || exit 1
echo "--- two-frame kernel"; timeout -k 10 120 python tools/gl_bench.py 32 60 || exit 1
echo "--- B=1"; timeout -k 10 120 python tools/gl_bench.py 1 60 || exit 1
echo "--- B=8"; timeout -k 10 120 python tools/gl_bench.py 8 60 || exit 1
if [ -f tacotron_amd/libtaco_b_stamps.so ]; then TACO_LIB=tacotron_amd/libtaco_b_stamps.so timeout -k 10 120 python tools/gl_stamps.py 32; fi
