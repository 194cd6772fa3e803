#!/bin/bash
# Same-box A/B of the in-step GEMM time with / without some kernel variants in the tuner:
#   tools/gemm_ab_instep.sh "14,15" [batch] [rounds]
# alternates `KMB_GEMM_EXCLUDE=<list>` and no exclusion, `rounds` times each, and prints "total GEMM ms" of every run
# (tools/gemm_shape_table.py: every GEMM launch of one training step timed with HIP events on its own stream).
EX=${1:-14,15}; B=${2:-1024}; R=${3:-2}
for i in $(seq 1 $R); do
  echo "--- round $i: excluded $EX"
  KMB_GEMM_EXCLUDE=$EX python tools/gemm_shape_table.py $B 2>/dev/null | tail -1
  echo "--- round $i: all variants"
  python tools/gemm_shape_table.py $B 2>/dev/null | tail -1
done
