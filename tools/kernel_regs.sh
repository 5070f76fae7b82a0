#!/bin/bash
# VGPR / scratch use of the kernels in one .hip file (device-only compile):
#   tools/kernel_regs.sh file.hip [grep-filter]     (run from the file's directory)
set -e
f=$1; pat=${2:-.}
tmp=$(mktemp -d /tmp/kregs.XXXXXX)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -c -o $tmp/k.bundle $f
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/k.bundle \
  --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/k.co | grep -E "\.name:|vgpr_count|private_segment_fixed_size" \
  | paste - - - | sed 's/ \+/ /g' | grep -E "$pat" | c++filt | cut -c1-200
rm -rf $tmp
