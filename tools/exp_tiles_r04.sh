mkdir -p gpurun_out/r4k
for v in base cgt_112 cgt_222 cgt_122 cgt_223 cgt_224 base; do
  if [ $v = base ]; then L=$PWD/agglomerationmultigrid1d_amd/libaggmg_hip.so; else L=$PWD/build_variants/libaggmg_$v.so; fi
  AGGMG_HIP_LIB=$L python tools/exp_cg_chain.py --log2-elems 24 >> gpurun_out/r4k/cg_$v.log 2>&1
done
for v in base ns4_3 ns4_4 base; do
  if [ $v = base ]; then L=$PWD/agglomerationmultigrid1d_amd/libaggmg_hip.so; else L=$PWD/build_variants/libaggmg_$v.so; fi
  AGGMG_HIP_LIB=$L python bench.py --no-cpu-baseline --no-smoother-bench --cg-log2-elems 0 --ragged-log2-elems 0 --also-log2-elems 0 >> gpurun_out/r4k/dg_$v.log 2>&1
done
echo done
