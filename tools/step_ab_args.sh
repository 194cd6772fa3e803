#!/bin/bash
# whole-step throughput for a SEQUENCE of extra bench.py argument strings on one box ("-" = none):
#   tools/step_ab_args.sh <batch> "-" "--serial" "-" "--serial"
BATCH=$1; shift
for E in "$@"; do
  if [ "$E" = "-" ]; then X=""; else X="$E"; fi
  echo -n "[$E] "; python bench.py --batch $BATCH --steps 20 --warmup 6 --no-cpu-baseline --no-roofline --no-pcie --no-extras $X 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
