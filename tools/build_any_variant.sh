#!/bin/bash
# A variant build of the library: ONE source compiled with extra flags, the other objects taken from the regular build
# (run `python -m tacotron_amd.build` first).   tools/build_any_variant.sh <name> <source.hip> [flags...]
#   tools/build_any_variant.sh nobigru api.hip -DTACO_EXP_NO_BIGRU   ->  tacotron_amd/libtaco_v_nobigru.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift; shift
o=tacotron_amd/csrc/build
base=${src%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" \
    -c tacotron_amd/csrc/${VSRC:-$src} -o $o/${base}_$name.o
objs=""
for f in api gemm rnn decoder dec_persistent dec_tile griffinlim; do
  if [ "$f" = "$base" ]; then objs="$objs $o/${base}_$name.o"; else objs="$objs $o/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tacotron_amd/libtaco_v_$name.so $objs
echo tacotron_amd/libtaco_v_$name.so
