#!/bin/bash
# kernel trace of the coarsest solve alone, caches flushed before every solve:
#   bash tools/prof_coarse.sh "20:2" [threads fill minwg]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "$2" ] && export AGGMG_CR_THREADS=$2 AGGMG_CR_FILL=$3 AGGMG_CR_MINWG=$4
rm -rf $R/gpurun_out/prof_coarse
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_coarse -o run -- python3 $R/tools/exp_coarse.py --cold --cases "${1:-24:1,20:2}" --steps 20 > $R/gpurun_out/prof_coarse.log 2>&1
f=$(ls $R/gpurun_out/prof_coarse/*kernel_stats.csv $R/gpurun_out/prof_coarse/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "cr_stage" in r["Name"] or "cr_tail" in r["Name"]:
        print(f'{r["Name"].split("(")[0].replace("void aggmg::", ""):45s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f} us')
PY
else echo "no kernel_stats.csv under gpurun_out/prof_coarse"; fi
grep ms_per $R/gpurun_out/prof_coarse.log | cut -c1-80
