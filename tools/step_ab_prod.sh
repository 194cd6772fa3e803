#!/bin/bash
# Same-box A/B of the whole training step under two environment settings with the PRODUCT library (knobs it reads itself:
# KMB_WGRAD_GROUP, KMB_SMALL_SPLIT, KMB_NO_SIDE_STREAM ...), alternating processes:
#   tools/step_ab_prod.sh "KMB_WGRAD_GROUP=0" "KMB_WGRAD_GROUP=1" [batch=64] [rounds=3]
A=$1; B=$2; BATCH=${3:-64}; R=${4:-3}
for i in $(seq 1 $R); do
  for S in "$A" "$B"; do
    printf "%s b=%s round %s: " "$S" "$BATCH" "$i"
    env $S python bench.py --batch $BATCH --no-extras --no-cpu-baseline --no-roofline --no-pcie --steps 20 --warmup 8 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step_windows'], 'clock', d['clock_mhz'], 'loss', d['final_loss'])"
  done
done
