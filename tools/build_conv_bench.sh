#!/bin/bash
# builds tools/conv_bench from the in-tree objects (run `python quickvc-official_amd/build.py` first)
set -e
cd "$(dirname "$0")/.."
O=quickvc-official_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/conv_bench.hip -o $O/conv_bench.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $O/conv_bench.o $O/qvc_conv_f16.o $O/qvc_conv_bf16.o $O/qvc_small.o $O/qvc_pack.o -o tools/conv_bench
