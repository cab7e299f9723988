# Griffin-Lim tests (bit identity of the state variants, parity) + timing of builds side by side
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread -k "griffin or cfg3 or no_alloc or noalloc" > $O/gl_tests.log 2>&1 || { tail -30 $O/gl_tests.log; exit 1; }
tail -3 $O/gl_tests.log
bash tools/run_gl_ab.sh "$@"
