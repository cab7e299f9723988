# Griffin-Lim timing of several builds of the library in one call on one box: tools/run_gl_ab.sh "" nopass nogather ...
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
for v in "$@"; do
  lib=""; [ -n "$v" ] && [ "$v" != head ] && lib=tacotron_amd/libtaco_v_$v.so
  echo "== build [$v]"
  TACO_LIB=$lib timeout -k 10 120 python tools/gl_bench.py 32 60 || exit 1
  TACO_LIB=$lib timeout -k 10 120 python tools/gl_bench.py 32 60 || exit 1
done
