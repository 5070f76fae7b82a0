# round-4 profile passes (run on the GPU box from the repo root; each rocprofv3 pass is its own run, counters never
# combined with a trace): config 2 smoother phases with 3 copies and 1 copy, the V-cycle roles of the two 2^24
# hierarchies, one rank's share of an 8-rank job, SQ counters of the coarsest solve.  Outputs under gpurun_out/r4prof_*;
# summaries via tools/summarize_smoother_profile.py / tools/summarize_profiles.py / tools/summarize_dist_trace.py.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for K in 3 1; do
  D=$R/gpurun_out/r4prof_smoother$K
  rm -rf $D && mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/tools/profile_smoother.py --copies $K > $D/kt.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $R/tools/profile_smoother.py --copies $K > $D/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $R/tools/profile_smoother.py --copies $K > $D/write.log 2>&1
  find $D -name "*agent_info.csv" -delete
  echo "profiled smoother copies=$K"
done
for spec in "dg 24" "cg 24"; do
  set -- $spec
  kind=$1; E=$2
  D=$R/gpurun_out/r4prof_$kind
  rm -rf $D && mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/tools/profile_vcycle.py --kind $kind --log2-elems $E --steps 6 > $D/kt.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $R/tools/profile_vcycle.py --kind $kind --log2-elems $E --steps 6 > $D/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $R/tools/profile_vcycle.py --kind $kind --log2-elems $E --steps 6 > $D/write.log 2>&1
  find $D -name "*agent_info.csv" -delete
  echo "profiled $kind"
done
D=$R/gpurun_out/r4prof_dist
rm -rf $D && mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/tools/exp_dist_rank.py --log2-elems 24 --world 8 --rank 3 --steps 20 > $D/kt.log 2>&1
find $D -name "*agent_info.csv" -delete
echo "profiled one rank of eight"
du -sh $R/gpurun_out/r4prof_*
