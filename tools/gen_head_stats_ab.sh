for i in 1 2 3; do
  echo -n "stats    "; python tools/gen_bench.py --reps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_generate'])"
  echo -n "nostats  "; KMB_GEN_HEAD_STATS=0 python tools/gen_bench.py --reps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_generate'])"
done
