#!/bin/bash
# build a kernel-experiment variant of librb3gpu.so: tools/build_variant.sh NAME -DFLAG...   -> ropebwt3_amd/prof/NAME.so
set -e
cd $(dirname $0)/..
N=$1; shift
mkdir -p ropebwt3_amd/prof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -Iinclude -Iropebwt3_amd/csrc -c ropebwt3_amd/csrc/rb3gpu.hip -o /tmp/rb3gpu_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ropebwt3_amd/prof/$N.so /tmp/rb3gpu_$N.o ropebwt3_amd/build/rb3gpu_sort.o ropebwt3_amd/build/rb3gpu_fmdenc.o ropebwt3_amd/build/rb3gpu_comm.o
