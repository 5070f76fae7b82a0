# -DAGGMG_NT=3 (non-temporal loads at the AGGMG_LD sites: the chain kernel's operator rows and transfer rows, the
# block-tridiagonal kernel's non-symmetric operator arrays) against the default (stores only): config 5's shape and config 3
# at 2^24 (build_variants/libaggmg_nt3.so: aggmg_hip.hip and cgt.hip compiled with the flag)
mkdir -p gpurun_out/nt3
for v in base nt3 base nt3; do
  if [ $v = base ]; then L=$PWD/agglomerationmultigrid1d_amd/libaggmg_hip.so; else L=$PWD/build_variants/libaggmg_$v.so; fi
  AGGMG_HIP_LIB=$L python bench.py --no-cpu-baseline --no-smoother-bench --ragged-log2-elems 0 --also-log2-elems 0 > gpurun_out/nt3/$v.json 2>> gpurun_out/nt3/err.log
  python - $v <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/nt3/{sys.argv[1]}.json'))
c=d['config5_2p24_1gpu']
print(sys.argv[1], 'config3', round(d['ms_per_step'],4), 'config5', round(c['ms_per_step'],4), {k:round(v['ms_per_launch'],4) for k,v in c['kernels'].items()})
PY
done
