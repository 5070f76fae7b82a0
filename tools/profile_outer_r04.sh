# rocprofv3 kernel trace of the device-resident outer loops at a test after every cycle / sweep (tools/exp_outer_loop.py):
# config 3's hierarchy and config 5's shape at 2^24 elements.  Outputs under gpurun_out/r4prof_outer_*.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for kind in dg cg; do
  D=$R/gpurun_out/r4prof_outer_$kind
  rm -rf $D && mkdir -p $D
  if [ $kind = cg ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/tools/exp_outer_loop.py --cg > $D/kt.log 2>&1
  else
    rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/tools/exp_outer_loop.py > $D/kt.log 2>&1
  fi
  find $D -name "*agent_info.csv" -delete
  find $D -name "*kernel_trace.csv" -delete
  echo "profiled outer loops $kind"
done
du -sh $R/gpurun_out/r4prof_outer_*
