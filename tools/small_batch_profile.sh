# Per-kernel profile of a small-batch step (b = 64, the reference default): bench line overlapped and serial, rocprofv3 kernel stats of the serial
# step with the GEMM tuning preloaded, per-launch GEMM times of one step.  Run through gpurun from the repo root; writes gpurun_out/b64/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/b64
mkdir -p $O
export KMB_GEMM_TUNE_FILE=$GRAFT_REPO_ROOT/$O/gemm_tune.txt
rm -f $KMB_GEMM_TUNE_FILE
B="python3 bench.py --batch 64 --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-pcie --no-extras"
$B 2>/dev/null | tail -1 | cut -c1-200
$B --serial 2>/dev/null | tail -1 | cut -c1-200
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profs -o s -- $B --serial > $O/profs.log 2>&1
python tools/summarize_rocprof.py $O/profs/s_kernel_stats.csv auto > $O/serial.md
python3 tools/one_step_gemm_trace.py 64 $O/launches.txt > $O/trace.log 2>&1
rm -rf $O/prof/*trace* $O/profs/*trace*
