#!/bin/bash
# the tracing variant of the library (cr_kernels.hpp stamps, -DAGGMG_CR_TRACE) beside the product build: build_trace/
set -e
cd "$(dirname "$0")/../agglomerationmultigrid1d_amd/csrc"
make
OUT=../../build_trace
mkdir -p $OUT
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -pthread"
for f in aggmg_hip cgt dist; do /opt/rocm/bin/hipcc $FLAGS -DAGGMG_CR_TRACE ${CR_TRACE_EXTRA} -c -o $OUT/$f.o $f.hip & done
wait
/opt/rocm/bin/hipcc $FLAGS -shared -o $OUT/libaggmg_hip_trace.so $OUT/aggmg_hip.o $OUT/cgt.o $OUT/dist.o setup.o spops.o -ldl
