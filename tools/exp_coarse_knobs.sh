#!/bin/bash
# chunk size / workgroup size of the cyclic-reduction stage kernels, caches flushed before every solve
for cfg in "256 12 256" "256 11 512" "256 10 512" "128 10 512" "64 9 512" "256 9 512" "64 8 512"; do
  set -- $cfg
  echo "== threads $1 fill $2 minwg $3"
  AGGMG_CR_THREADS=$1 AGGMG_CR_FILL=$2 AGGMG_CR_MINWG=$3 timeout -k 10 120 python tools/exp_coarse.py --cold --steps 20 --cases "24:1,20:2,17:2" || exit 1
done
