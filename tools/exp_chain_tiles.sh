#!/bin/bash
# Build variants of the chain kernel's tile shape (slabs per thread for M = 1, 2, 4) and time the config-5
# hierarchy with each: AGGMG_HIP_LIB selects the library a run loads.  Measurement aid.
set -e
cd "$(dirname "$0")/../agglomerationmultigrid1d_amd/csrc"
mkdir -p ../../build_variants
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -pthread"
for v in "1 1 1" "2 2 2" "4 4 2" "4 2 2" "2 4 2" "4 4 3" "2 2 3" "3 3 2"; do
  set -- $v
  out=../../build_variants/libaggmg_cgt_$1$2$3.so
  /opt/rocm/bin/hipcc $FLAGS -DAGGMG_CGT_NS1=$1 -DAGGMG_CGT_NS2=$2 -DAGGMG_CGT_NS4=$3 -c -o /tmp/cgt_$1$2$3.o cgt.hip
  /opt/rocm/bin/hipcc $FLAGS -shared -o $out aggmg_hip.o /tmp/cgt_$1$2$3.o dist.o setup.o spops.o -ldl
  echo built $out
done
