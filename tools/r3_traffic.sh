# per-shape GEMM traffic of one step (run through gpurun from the repo root): BATCH=1024 tools/r3_traffic.sh
BATCH=${BATCH:-1024}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/traffic_b$BATCH
rm -rf $O; mkdir -p $O
export KMB_GEMM_TUNE_FILE=$GRAFT_REPO_ROOT/$O/gemm_tune.txt
python3 tools/one_step_gemm_trace.py $BATCH $O/launches0.txt > $O/warm.log 2>&1     # fills the tuning file
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 tools/one_step_gemm_trace.py $BATCH $O/launches_f.txt > $O/f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 tools/one_step_gemm_trace.py $BATCH $O/launches_w.txt > $O/w.log 2>&1
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/gemm_traffic_by_shape.py $O/launches_f.txt $F $W | tee $O/traffic_by_shape.txt
rm -rf $O/pmc_fetch $O/pmc_write    # tens of MB of CSV: only the table is kept
