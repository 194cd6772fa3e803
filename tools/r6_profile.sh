# Evidence run for profiles/ (run through gpurun from the repo root; R = round tag, default r06).  Round 6 = round 5's script with
#   * the fingerprint of the GEMM sources the passes ran on (gemm_sources.sha -> profiles/r06_b1024_pmc_traffic.json; bench.py quotes the file's
#     traffic only when the fingerprint is its own tree's),
#   * kernel-stats tables whose per-step columns take the step count from the run (tools/summarize_rocprof.py auto),
#   * an ordered kernel trace of the generation bench (what sits between the decode blocks),
#   * without the resident decoder-layers kernel (tools/experiments/ since this round).
#   1. default bench (JSON line incl. roofline, cpu_baseline, fine_tune / batch-sweep / generation legs); writes the GEMM tuning choices to a file
#   2. rocprofv3 --kernel-trace --stats of the product configuration (tuning preloaded: no tuning launches), overlapped and --serial
#   3. PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA-busy; --pmc with --kernel-trace only)
#   4. per-shape GEMM table, library yardstick, per-shape traffic join, generation kernel stats, pre-training step, attention / top-k timings
R=${R:-r06}
BATCH=${BATCH:-1024}     # per-GPU batch of the training legs (bench.py's default)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
python3 -c "import bench; print(bench.gemm_sources_sha16())" 2>/dev/null | tail -1 > $O/gemm_sources.sha
export KMB_GEMM_TUNE_FILE=$GRAFT_REPO_ROOT/$O/gemm_tune.txt
rm -f $KMB_GEMM_TUNE_FILE
timeout 900 python bench.py --batch $BATCH > $O/bench_default.log 2> $O/bench_default.err
B="python3 bench.py --batch $BATCH --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-pcie --no-extras"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- $B > $O/prof.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profs -o s -- $B --serial > $O/profs.log 2>&1
P="python3 bench.py --batch $BATCH --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-pcie --no-extras --serial"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- $P > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- $P > $O/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o m -- $P > $O/pmc_mfma.log 2>&1
timeout 300 python tools/gemm_shape_table.py $BATCH 2>&1 | grep -v amdgpu > $O/gemm_shapes.txt
timeout 400 python tools/gemm_yardstick.py $O/gemm_shapes.txt 2>/dev/null > $O/yardstick.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gen -o g -- python3 tools/gen_bench.py --reps 5 > $O/gen_bench.log 2>&1
python3 tools/gen_trace_neighbours.py $(find $O/prof_gen -name "*kernel_trace.csv" | head -1) > $O/gen_trace_neighbours.txt 2>&1
timeout 200 python tools/pretrain_bench.py --batch 384 2>&1 | grep -v amdgpu | tail -3 > $O/pretrain.log
timeout 200 python tools/attn_bwd_time.py 2>&1 | grep -v amdgpu > $O/attn_bwd.txt
# per-shape HBM-side traffic of one step's GEMM launches (joins the launch list with per-dispatch FETCH_SIZE / WRITE_SIZE)
timeout 300 python3 tools/one_step_gemm_trace.py $BATCH $O/launches0.txt > $O/trace_warm.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/tpmc_fetch -o f -- python3 tools/one_step_gemm_trace.py $BATCH $O/launches_f.txt > $O/tf.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/tpmc_write -o w -- python3 tools/one_step_gemm_trace.py $BATCH $O/launches_w.txt > $O/tw.log 2>&1
python3 tools/gemm_traffic_by_shape.py $O/launches_f.txt $(find $O/tpmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/tpmc_write -name "*counter_collection.csv" | head -1) > $O/traffic_by_shape.txt 2>&1
rm -rf $O/tpmc_fetch $O/tpmc_write
timeout 100 python tools/topk_time.py 2>&1 | grep -v amdgpu > $O/topk_time.txt
timeout 100 python tools/allrows_stamps.py 2>&1 | grep -v amdgpu > $O/allrows_stamps.txt   # needs lib/libkmbart_hip_stamp.so (tools/gemm_stamps.py --build)
timeout 100 python tools/allrows_time.py 2>&1 | grep -v amdgpu > $O/allrows_time.txt
timeout 200 python tools/decode_stamps.py 2>&1 | grep -v amdgpu > $O/decode_stamps.txt     # needs lib/libkmbart_hip_dstamp.so (tools/decode_stamps.py --build)
timeout 200 python tools/gen_host_wait.py 64 2>&1 | grep -v amdgpu | tail -1 > $O/gen_host_wait.txt
tail -1 $O/bench_default.log | cut -c1-400
find $O -name "*.csv" | head -20
find $O -name "*kernel_trace.csv" -size +8M -delete   # (the merged-back budget is 64 MiB)
