O=gpurun_out/r6_s3
mkdir -p $O
python -m pytest tests/test_data_pipeline_gpu.py tests/test_model_gpu.py tests/test_api_gpu.py -x -q -m gpu > $O/tests.log 2>&1
echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
python tools/gen_bench.py --reps 10 2>/dev/null | tail -1 > $O/gen.json; cat $O/gen.json
KMB_LIB_PATH=km-bart_amd/lib/libkmbart_hip_prev.so python tools/gen_bench.py --reps 10 2>/dev/null | tail -1 > $O/gen_prev.json; cat $O/gen_prev.json
python tools/gen_bench.py --reps 10 2>/dev/null | tail -1 > $O/gen2.json; cat $O/gen2.json
