mkdir -p gpurun_out/nt6
for v in base nt6 base nt6 base nt6; do
  if [ $v = base ]; then L=$PWD/agglomerationmultigrid1d_amd/libaggmg_hip.so; else L=$PWD/build_variants/libaggmg_$v.so; fi
  AGGMG_HIP_LIB=$L python bench.py --no-cpu-baseline --no-smoother-bench --cg-log2-elems 0 --ragged-log2-elems 0 --also-log2-elems 0 > gpurun_out/nt6/$v.json 2>> gpurun_out/nt6/err.log
  python - $v <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/nt6/{sys.argv[1]}.json'))
print(sys.argv[1], round(d['ms_per_step'],4), {k:round(v['ms_per_launch'],4) for k,v in d['kernels'].items()}, round(d['vcycles_loop']['ms_per_cycle'],4))
PY
done
