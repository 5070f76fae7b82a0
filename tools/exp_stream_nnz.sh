# generic CSR residual: the stream kernel with row blocks of 768 / 1024 / 2048 / 4096 entries (build_variants/libaggmg_snnz*.so,
# -DAGGMG_STREAM_NNZ=...) against the row-thread kernels and rocSPARSE's adaptive csrmv (tools/exp_rocsparse_calib.py)
mkdir -p gpurun_out/snnz
for v in ${VARIANTS:-base s2048 s2048r s2048rn s1536rn s2048}; do
  if [ $v = base ]; then L=$PWD/agglomerationmultigrid1d_amd/libaggmg_hip.so; else L=$PWD/build_variants/libaggmg_$v.so; fi
  for rt in ${RTS:-0}; do
    echo "$v rowthread=$rt" >> gpurun_out/snnz/all.log
    AGGMG_HIP_LIB=$L AGGMG_CSR_ROWTHREAD=$rt python tools/exp_rocsparse_calib.py >> gpurun_out/snnz/all.log 2>&1
  done
done
python - <<'PY'
import json
name=None
for ln in open('gpurun_out/snnz/all.log'):
    if ln.startswith('{'):
        d=json.loads(ln)
        print(name, {k:(round(v['libaggmg_generic_residual']['us'],1), round(v['rocsparse_csrmv_adaptive']['us'],1)) for k,v in d.items()})
    else: name=ln.strip()
PY
