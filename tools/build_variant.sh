#!/bin/bash
# A variant build of the library with extra compiler flags for griffinlim.hip (diagnostic switches), e.g.
#   tools/build_variant.sh stamps -DGL_STAMPS -DGL_STAMP_WG=1016   ->  tacotron_amd/libtaco_v_stamps.so
# The other objects are the ones of the regular build (run `python -m tacotron_amd.build` first).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
o=tacotron_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" \
    -c tacotron_amd/csrc/${GLSRC:-griffinlim.hip} -o $o/griffinlim_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tacotron_amd/libtaco_v_$name.so \
    $o/api.o $o/gemm.o $o/rnn.o $o/decoder.o $o/dec_persistent.o $o/dec_tile.o $o/griffinlim_$name.o
echo tacotron_amd/libtaco_v_$name.so
