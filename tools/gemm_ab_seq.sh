#!/bin/bash
# in-step GEMM total under a SEQUENCE of environment settings on one box (order effects visible):
#   tools/gemm_ab_seq.sh <batch> "ENV_A" "ENV_B" "ENV_B" "ENV_A" ...      ("-" = no setting)
BATCH=$1; shift
for E in "$@"; do
  if [ "$E" = "-" ]; then X="KMB_NOP=1"; else X="$E"; fi
  echo -n "[$E] "; env $X python tools/gemm_shape_table.py $BATCH 2>/dev/null | tail -1
done
