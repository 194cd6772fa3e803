# L1 -> L2 read latency, L2 hit rate and tag stalls per kernel of a training step (diagnostic: what the GEMM K loops wait for).
# Three rocprofv3 --pmc passes (each with --kernel-trace only) over one profiled step; run through gpurun from the repo root.
#   BATCH=1024 bash tools/pmc_l2_latency.sh   -> gpurun_out/pmc_l2/summary.md
BATCH=${BATCH:-1024}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_l2
rm -rf $O; mkdir -p $O
export KMB_GEMM_TUNE_FILE=$GRAFT_REPO_ROOT/$O/gemm_tune.txt
python3 bench.py --batch $BATCH --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-pcie --no-extras --serial > $O/warm.log 2>&1
P="python3 bench.py --batch $BATCH --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-pcie --no-extras --serial"
timeout 500 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d $O/lat -o l -- $P > $O/lat.log 2>&1
timeout 500 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/hit -o h -- $P > $O/hit.log 2>&1
timeout 500 rocprofv3 --pmc TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/stall -o s -- $P > $O/stall.log 2>&1
python3 tools/pmc_l2_summary.py $O > $O/summary.md 2>&1
rm -rf $O/lat/*trace* $O/hit/*trace* $O/stall/*trace*
