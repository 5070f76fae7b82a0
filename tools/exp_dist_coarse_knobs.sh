for cfg in "12 256" "11 256" "10 256" "10 128" "9 64" "9 128" "8 64"; do
  set -- $cfg
  echo "== max_q $1 threads $2"
  AGGMG_CR_MAX_Q=$1 AGGMG_CR_THREADS=$2 timeout -k 10 120 python tools/exp_dist_rank.py --log2-elems 24 --world 8 --rank 3 --profile 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_cycle'],4), round(d['ms_issue_per_cycle'],4), d['chunked'], round(d['kernels_ms']['coarse_L0'],4), round(d['kernels_sum_ms'],4))"
done
