# round 6, call 3: the API / decode tests this round touched, the default bench (all legs) on the new library, the b = 64 profile
O=gpurun_out/r6_s2
mkdir -p $O
python -m pytest tests/test_api_gpu.py tests/test_decode_fused_gpu.py tests/test_bench_two_ranks_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/tests.log 2>&1
echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
bash tools/small_batch_profile.sh > $O/b64.log 2>&1
cat $O/b64.log | tail -5
