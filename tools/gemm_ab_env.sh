#!/bin/bash
# Same-box A/B of the in-step GEMM time under two environment settings:
#   tools/gemm_ab_env.sh "KMB_GEMM_FORCE_ORDER=1" "KMB_GEMM_FORCE_ORDER=0" [batch] [rounds]
export KMB_USE_DIAG=1   # the A/B knobs live in the diagnostic build (csrc/diag.h)
A=$1; B=$2; BATCH=${3:-1024}; R=${4:-2}
for i in $(seq 1 $R); do
  echo "--- round $i: $A"; env $A python tools/gemm_shape_table.py $BATCH 2>/dev/null | tail -1
  echo "--- round $i: $B"; env $B python tools/gemm_shape_table.py $BATCH 2>/dev/null | tail -1
done
