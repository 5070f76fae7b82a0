# r04 tile-shape sweeps after the DPP change (build the variants first: see the loop heads; AGGMG_HIP_LIB picks one)
mkdir -p gpurun_out/r4k
for v in base cgtnt512 base cgtnt512; do
  if [ $v = base ]; then L=$PWD/agglomerationmultigrid1d_amd/libaggmg_hip.so; else L=$PWD/build_variants/libaggmg_$v.so; fi
  AGGMG_HIP_LIB=$L python tools/exp_cg_chain.py --log2-elems 24 >> gpurun_out/r4k/cg2_$v.log 2>&1
done
for v in base nt4_512_1 nt4_512_2 nt4_128_4 base nt4_512_1 nt4_512_2; do
  if [ $v = base ]; then L=$PWD/agglomerationmultigrid1d_amd/libaggmg_hip.so; else L=$PWD/build_variants/libaggmg_$v.so; fi
  AGGMG_HIP_LIB=$L python bench.py --no-cpu-baseline --no-smoother-bench --cg-log2-elems 0 --ragged-log2-elems 0 --also-log2-elems 0 >> gpurun_out/r4k/dg2_$v.log 2>&1
done
echo done
