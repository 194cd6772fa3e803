#!/bin/bash
# Same-box A/B of the whole training step under two environment settings, alternating processes (diagnostic library: the A/B
# knobs of csrc/diag.h exist only there):
#   tools/step_ab.sh "KMB_SIDE_PRIORITY=high" "KMB_SIDE_PRIORITY=default" [batch=1024] [rounds=3]
export KMB_LIB_PATH=km-bart_amd/lib/libkmbart_hip_diag.so
A=$1; B=$2; BATCH=${3:-1024}; R=${4:-3}
for i in $(seq 1 $R); do
  for S in "$A" "$B"; do
    printf "%s b=%s round %s: " "$S" "$BATCH" "$i"
    env $S python bench.py --batch $BATCH --no-extras --no-cpu-baseline --no-roofline --no-pcie --steps 20 --warmup 8 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step_windows'], 'clock', d['clock_mhz'])"
  done
done
