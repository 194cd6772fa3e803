#!/bin/bash
# Kernel times of the fused decode blocks against the number of rows (batch x beams): is a block bound by what ONE CU can
# pull in (time independent of the rows while workgroups <= CUs) or by what the L2s / the fabric deliver to all of them
# (time grows with the rows)?   bash tools/gen_rows_sweep.sh   (GPU box; writes gpurun_out/gen_rows/)
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
O=gpurun_out/gen_rows; mkdir -p $O
for B in 8 16 32 64; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$B -o g -- python3 tools/gen_bench.py --batch $B --reps 3 > $O/b$B.log 2>&1
  echo "== batch $B (rows $((B*5)))" >> $O/summary.md
  grep '^{' $O/b$B.log | tail -1 >> $O/summary.md
  python3 tools/summarize_rocprof.py $O/p$B/g_kernel_stats.csv 4 | grep -E "decode_|allrows|topk_part|gather_rows|all kernels" >> $O/summary.md
done
python3 tools/decode_stamps.py > $O/stamps.txt 2>&1
cat $O/summary.md; cat $O/stamps.txt | tail -40
