set -o pipefail
python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "griffin or gl or cfg3 or no_alloc or golden or synth" > gpurun_out/r2_t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t4.log; tail -8 gpurun_out/r2_t4.log
echo "--- old one-frame kernel"; TACO_GL_ONE_FRAME=1 timeout -k 10 120 python tools/gl_bench.py 32 60 || exit 1
echo "--- v0 (T2+T3 regs, spills)"; timeout -k 10 120 python tools/gl_bench.py 32 60 || exit 1
echo "--- v1 (T2 regs)"; TACO_LIB=tacotron_amd/libtaco_b_v1.so timeout -k 10 120 python tools/gl_bench.py 32 60 || exit 1
echo "--- v3 (LDS twiddles)"; TACO_LIB=tacotron_amd/libtaco_b_v3.so timeout -k 10 120 python tools/gl_bench.py 32 60 || exit 1
echo "--- B=1"; TACO_GL_ONE_FRAME=1 timeout -k 10 120 python tools/gl_bench.py 1 60; timeout -k 10 120 python tools/gl_bench.py 1 60; TACO_LIB=tacotron_amd/libtaco_b_v3.so timeout -k 10 120 python tools/gl_bench.py 1 60
