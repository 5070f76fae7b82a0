# three rocprofv3 passes (kernel trace + stats, FETCH_SIZE, WRITE_SIZE) of default V-cycles of one hierarchy kind:
#   tools/profile_kind_r03.sh dg|cg [log2_elems]   ->  gpurun_out/r3prof_<kind>/{kt,fetch,write}
# then here: python tools/summarize_profiles.py gpurun_out/r3prof_<kind> r03 <kind> <log2_elems>
set -e
kind=$1; E=${2:-24}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/r3prof_$kind
rm -rf $D && mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/tools/profile_vcycle.py --kind $kind --log2-elems $E --steps 6 > $D/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $R/tools/profile_vcycle.py --kind $kind --log2-elems $E --steps 6 > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $R/tools/profile_vcycle.py --kind $kind --log2-elems $E --steps 6 > $D/write.log 2>&1
find $D -name "*agent_info.csv" -delete
echo profiled $kind
