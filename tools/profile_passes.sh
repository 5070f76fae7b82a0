set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "dg 24" "cg 24" "cggeneric 22"; do
  set -- $spec
  kind=$1; E=$2
  D=gpurun_out/r2prof_$kind
  mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 tools/profile_vcycle.py --kind $kind --log2-elems $E --steps 6 > $D/kt.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 tools/profile_vcycle.py --kind $kind --log2-elems $E --steps 6 > $D/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 tools/profile_vcycle.py --kind $kind --log2-elems $E --steps 6 > $D/write.log 2>&1
  echo "profiled $kind"
  # keep the merged-back output small: only the csv files we summarise
  find $D -name "*agent_info.csv" -delete
done
du -sh gpurun_out/r2prof_dg gpurun_out/r2prof_cg gpurun_out/r2prof_cggeneric
