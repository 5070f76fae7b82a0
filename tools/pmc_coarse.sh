#!/bin/bash
# instruction / wait counters of the coarsest-solve kernels (caches flushed before every solve)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_coarse
SETS=${AGGMG_PMC_SETS:-"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD|SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE|SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"}
IFS="|" read -ra ALL <<< "$SETS"
for set in "${ALL[@]}"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_coarse/$tag -o run -- python3 $R/tools/exp_coarse.py --cold --cases "${1:-20:2}" --steps 6 > $R/gpurun_out/pmc_coarse_$tag.log 2>&1 || echo "set failed: $set"
done
python3 - $R/gpurun_out/pmc_coarse <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void aggmg::", "")
        if k.startswith("cr_stage") or k.startswith("cr_tail"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {sum(v) / len(v):16.0f}   (n={len(v)})")
PY
