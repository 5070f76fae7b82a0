#!/bin/bash
# Build variants of the two-level kernels' tile shape (slabs per thread on the finer / coarser level of a pair):
# AGGMG_HIP_LIB selects the library a run loads.  Measurement aid.
set -e
cd "$(dirname "$0")/../agglomerationmultigrid1d_amd/csrc"
mkdir -p ../../build_variants
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -pthread"
for v in "2 1" "4 2" "3 2"; do
  set -- $v
  out=../../build_variants/libaggmg_pair_$1$2.so
  /opt/rocm/bin/hipcc $FLAGS -DAGGMG_PAIR_NSA=$1 -DAGGMG_PAIR_NSB=$2 -c -o /tmp/aggmg_pair_$1$2.o aggmg_hip.hip
  /opt/rocm/bin/hipcc $FLAGS -shared -o $out /tmp/aggmg_pair_$1$2.o cgt.o dist.o setup.o spops.o -ldl
  echo built $out
done
