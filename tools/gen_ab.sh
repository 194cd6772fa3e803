#!/bin/bash
# A/B of two builds of the library on the generation bench, alternating processes on one box:
#   tools/gen_ab.sh <variant-lib-name> [reps]      e.g. tools/gen_ab.sh vu20
V=${1:?variant}; REPS=${2:-20}
for i in 1 2 3; do
  echo -n "product  "; python tools/gen_bench.py --reps $REPS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_generate'])"
  echo -n "$V  "; KMB_LIB_PATH=km-bart_amd/lib/libkmbart_hip_$V.so python tools/gen_bench.py --reps $REPS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_generate'])"
done
