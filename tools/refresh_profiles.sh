# Turn the outputs of tools/r3_profile.sh (r2_profile.sh for round 2) (gpurun_out/$R/, merged back by gpurun) into the committed files under profiles/.
# (kernel-stats tables: the step count comes from the run itself -- summarize_rocprof.py auto -- not from a constant)
# Run from the repo root on the build container:  bash tools/refresh_profiles.sh [r02] [1024] [r02]
#   $1 = the R tag the evidence run wrote under (gpurun_out/$1), $2 = its per-GPU batch, $3 = prefix of the committed files
I=${1:-r02}
B=${2:-1024}
R=${3:-$I}
O=gpurun_out/$I
set -e
tail -1 $O/bench_default.log > profiles/${R}_bench_b${B}.json
cp $O/prof/p_kernel_stats.csv profiles/${R}_kernel_stats_b${B}.csv
cp $O/profs/s_kernel_stats.csv profiles/${R}_kernel_stats_b${B}_serial.csv
python tools/summarize_rocprof.py $O/prof/p_kernel_stats.csv auto > profiles/${R}_kernel_stats_b${B}.md
python tools/summarize_rocprof.py $O/profs/s_kernel_stats.csv auto > profiles/${R}_kernel_stats_b${B}_serial.md
python tools/summarize_pmc.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv \
    --json profiles/${R}_b${B}_pmc_traffic.json --batch ${B} --gemm-sha $(cat $O/gemm_sources.sha) > profiles/${R}_pmc_hbm_traffic_b${B}.md
python tools/summarize_pmc_mfma.py $O/pmc_mfma/m_counter_collection.csv > profiles/${R}_pmc_mfma_util_b${B}.md
cp $O/gemm_shapes.txt profiles/${R}_gemm_shape_table_b${B}.txt
cp $O/yardstick.txt profiles/${R}_gemm_library_yardstick_b${B}.txt
cp $O/attn_bwd.txt profiles/${R}_attn_bwd_time.txt
[ -f $O/decode_stamps.txt ] && cp $O/decode_stamps.txt profiles/${R}_decode_stamps.txt
[ -f $O/gen_host_wait.txt ] && cp $O/gen_host_wait.txt profiles/${R}_generation_host_wait.txt
[ -f $O/traffic_by_shape.txt ] && cp $O/traffic_by_shape.txt profiles/${R}_gemm_traffic_by_shape_b${B}.txt
[ -f $O/gen_trace_neighbours.txt ] && cp $O/gen_trace_neighbours.txt profiles/${R}_generation_trace_neighbours.txt
cp $O/topk_time.txt profiles/${R}_topk_time.txt
[ -f $O/allrows_stamps.txt ] && cp $O/allrows_stamps.txt profiles/${R}_allrows_stamps_run.txt && cat $O/allrows_time.txt >> profiles/${R}_allrows_stamps_run.txt
grep '^{' $O/gen_bench.log | tail -1 > profiles/${R}_generation_bench.json
python tools/summarize_rocprof.py $O/prof_gen/g_kernel_stats.csv 6 > profiles/${R}_generation_kernel_stats.md
grep '^{' $O/pretrain.log | tail -1 > profiles/${R}_pretrain_bench.json
echo "profiles/${R}_* refreshed from $O"
