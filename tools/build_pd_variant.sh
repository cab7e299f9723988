#!/bin/bash
# A variant build of the library with extra compiler flags for dec_persistent.hip, e.g.
#   tools/build_pd_variant.sh l2 -DPD_AUX_LD=1 -DPD_AUX_ST=0   ->  tacotron_amd/libtaco_v_l2.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
o=tacotron_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" \
    -c tacotron_amd/csrc/dec_persistent.hip -o $o/dec_persistent_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tacotron_amd/libtaco_v_$name.so \
    $o/api.o $o/gemm.o $o/rnn.o $o/decoder.o $o/dec_persistent_$name.o $o/dec_tile.o $o/griffinlim.o
echo tacotron_amd/libtaco_v_$name.so
