#!/bin/bash
# whole-step throughput under a SEQUENCE of environment settings on one box ("-" = no setting):
#   tools/step_ab_seq.sh <batch> "ENV_A" "-" "ENV_A" "-" ...
BATCH=$1; shift
for E in "$@"; do
  if [ "$E" = "-" ]; then X="KMB_NOP=1"; else X="$E"; fi
  echo -n "[$E] "; env $X python bench.py --batch $BATCH --steps ${STEPS:-20} --warmup ${WARM:-6} --no-cpu-baseline --no-roofline --no-pcie --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
