# Evidence run for profiles/ (run through gpurun from the repo root):
#   1. default bench (JSON line incl. roofline + cpu_baseline); writes the GEMM tuning choices to a file
#   2. rocprofv3 --kernel-trace --stats of the product configuration (tuning preloaded: no tuning launches)
#   3. the same with --serial (weight-gradient GEMMs not overlapped): stand-alone kernel durations
#   4. two PMC passes (FETCH_SIZE, WRITE_SIZE; --pmc with --kernel-trace only) for HBM traffic per launch
#   5. per-shape GEMM table, generation benchmark
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export KMB_GEMM_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/gemm_tune.txt
rm -f $KMB_GEMM_TUNE_FILE
timeout 600 python bench.py > gpurun_out/r3_bench_default.log 2> gpurun_out/r3_bench_default.err
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-pcie"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3 -o r3 -- $B > gpurun_out/r3_prof.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3s -o r3s -- $B --serial > gpurun_out/r3s_prof.log 2>&1
P="python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-pcie --serial"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- $P > gpurun_out/r3_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- $P > gpurun_out/r3_pmc_write.log 2>&1
timeout 300 python tools/gemm_shape_table.py 512 2>&1 | grep -v amdgpu > gpurun_out/r3_gemm_shapes_b512.txt
timeout 300 python tools/gen_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r3_gen_bench.log
tail -1 gpurun_out/r3_bench_default.log | cut -c1-300
ls gpurun_out/pmc_fetch gpurun_out/pmc_write
#   6. in-kernel GEMM timelines (diagnostic library, built beforehand with `python tools/gemm_stamps.py --build`),
#      library yardstick on the measured shape table
( for v in 7 8 11; do echo "== KMB_GEMM_VARIANT=$v"; for shp in "16384 3072 768" "4096 4096 4096"; do KMB_GEMM_VARIANT=$v timeout 100 python tools/gemm_stamps.py $shp 2>&1 | grep -v amdgpu; done; done ) > gpurun_out/r3_gemm_stamps.txt
timeout 300 python tools/gemm_yardstick.py gpurun_out/r3_gemm_shapes_b512.txt 2>/dev/null > gpurun_out/r3_yardstick.txt
timeout 200 python tools/pretrain_bench.py --batch 384 2>&1 | grep -v amdgpu | tail -3 > gpurun_out/r3_pretrain.log
