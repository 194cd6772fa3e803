# Evidence run for profiles/: default bench JSON, rocprofv3 kernel stats of the product configuration, and a --serial
# trace (weight-gradient GEMMs not overlapped) for stand-alone kernel durations.  Run through gpurun from the repo root.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 500 python bench.py > gpurun_out/r2_bench_default.log 2> gpurun_out/r2_bench_default.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2 -o r2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-pcie > gpurun_out/r2_prof.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2s -o r2s -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-pcie --serial > gpurun_out/r2s_prof.log 2>&1
tail -2 gpurun_out/r2_bench_default.log
