#!/bin/bash
# A/B of one environment knob of the generation path on the generation bench, alternating processes on one box:
#   tools/gen_knob_ab.sh KMB_GEN_FOLD_EMBED [reps] [rounds]     (default vs KNOB=0)
K=${1:?knob}; REPS=${2:-40}; N=${3:-5}
for i in $(seq $N); do
  echo -n "default   "; python tools/gen_bench.py --reps $REPS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_generate'])"
  echo -n "$K=0  "; env $K=0 python tools/gen_bench.py --reps $REPS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_generate'])"
done
