#!/bin/bash
# builds tools/conv_bench from the in-tree objects (run `python quickvc-official_amd/build.py` first)
set -e
cd "$(dirname "$0")/.."
O=quickvc-official_amd/csrc/_obj
# optional: PF_CONV / PF_WN env vars rebuild the f16 kernels with another prefetch depth into tools/conv_bench_$TAG
TAG=${TAG:-}
if [ -n "$PF_CONV$PF_WN$EXTRA" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DQVC_PF_CONV=${PF_CONV:-3} -DQVC_PF_WN=${PF_WN:-3} $EXTRA -c quickvc-official_amd/csrc/qvc_conv_f16.hip -o $O/qvc_conv_f16_$TAG.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/conv_bench.hip -o $O/conv_bench.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $O/conv_bench.o $O/qvc_conv_f16_$TAG.o $O/qvc_conv_bf16.o $O/qvc_small.o $O/qvc_pack.o -o tools/conv_bench_$TAG
  exit 0
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/conv_bench.hip -o $O/conv_bench.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $O/conv_bench.o $O/qvc_conv_f16.o $O/qvc_conv_bf16.o $O/qvc_small.o $O/qvc_pack.o -o tools/conv_bench
