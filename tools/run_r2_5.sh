set -o pipefail
python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t5.log; tail -5 gpurun_out/r2_t5.log
echo "--- one-frame kernel"; TACO_GL_ONE_FRAME=1 timeout -k 10 120 python tools/gl_bench.py 32 60 || exit 1
echo "--- two-frame kernel"; timeout -k 10 120 python tools/gl_bench.py 32 60 || exit 1
echo "--- B=1"; TACO_GL_ONE_FRAME=1 timeout -k 10 120 python tools/gl_bench.py 1 60; timeout -k 10 120 python tools/gl_bench.py 1 60
echo "--- B=8"; TACO_GL_ONE_FRAME=1 timeout -k 10 120 python tools/gl_bench.py 8 60; timeout -k 10 120 python tools/gl_bench.py 8 60
