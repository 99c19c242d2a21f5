#!/bin/bash
# builds tools/conv_bench from the in-tree objects (run `python quickvc-official_amd/build.py` first)
#   EXTRA="-DQVC_ABLATE=32" TAG=ab32 ./tools/build_conv_bench.sh   -> tools/conv_bench_ab32 (kernel TUs rebuilt with EXTRA)
#   EXTRA="-DQVC_STAMP" TAG=stamp ...                              -> phase stamps of the WaveNet stack kernels
set -e
cd "$(dirname "$0")/.."
O=quickvc-official_amd/csrc/_obj
C=quickvc-official_amd/csrc
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC"
TAG=${TAG:-}
if [ -n "$PF_CONV$PF_WN$EXTRA" ]; then
  X="-DQVC_PF_CONV=${PF_CONV:-3} -DQVC_PF_WN=${PF_WN:-3} $EXTRA"
  # every translation unit that sees the kernel argument structs gets the same defines
  $H $X -c $C/qvc_conv_f16.hip -o $O/qvc_conv_f16_$TAG.o &
  $H $X -c $C/qvc_wn2.hip -o $O/qvc_wn2_$TAG.o &
  $H $X -c $C/qvc_chain.hip -o $O/qvc_chain_$TAG.o &
  $H $X -c $C/qvc_small.hip -o $O/qvc_small_$TAG.o &
  $H $X -c tools/conv_bench.hip -o $O/conv_bench_$TAG.o &
  wait
  $H $O/conv_bench_$TAG.o $O/qvc_conv_f16_$TAG.o $O/qvc_wn2_$TAG.o $O/qvc_chain_$TAG.o $O/qvc_conv_bf16.o $O/qvc_small_$TAG.o $O/qvc_pack.o -o tools/conv_bench_$TAG
  exit 0
fi
$H -c tools/conv_bench.hip -o $O/conv_bench.o
$H $O/conv_bench.o $O/qvc_conv_f16.o $O/qvc_wn2.o $O/qvc_chain.o $O/qvc_conv_bf16.o $O/qvc_small.o $O/qvc_pack.o -o tools/conv_bench
