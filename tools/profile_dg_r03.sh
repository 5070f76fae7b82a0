set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/r3prof_dg
rm -rf $D && mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/tools/profile_vcycle.py --kind dg --log2-elems 24 --steps 6 > $D/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $R/tools/profile_vcycle.py --kind dg --log2-elems 24 --steps 6 > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $R/tools/profile_vcycle.py --kind dg --log2-elems 24 --steps 6 > $D/write.log 2>&1
find $D -name "*agent_info.csv" -delete
echo profiled dg
