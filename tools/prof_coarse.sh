#!/bin/bash
# kernel trace of the coarsest solve alone: bash tools/prof_coarse.sh "24:1,20:2"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_coarse
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_coarse -o run -- python3 $R/tools/exp_coarse.py --cases "${1:-24:1,20:2}" --steps 20 > $R/gpurun_out/prof_coarse.log 2>&1
f=$(ls $R/gpurun_out/prof_coarse/*kernel_stats.csv $R/gpurun_out/prof_coarse/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then grep -E "cr_stage|cr_tail|Name" "$f" | cut -c1-220; else echo "no kernel_stats.csv under gpurun_out/prof_coarse"; fi
