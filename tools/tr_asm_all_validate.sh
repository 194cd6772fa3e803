# Round-6 starter: validate the inline-asm transposing reads (a) as shipped at the end of round 5 (v7 / v8 / 256-wide v11: the full suite never ran on
# them -- the round's GPU budget ended) and (b) for the kernels still on the intrinsic (-DKMB_TR_ASM_ALL: variants 5, 12, 13, 15 and gemm_lean.hip = 6).
# On the build container first (no GPU; the libraries travel with the snapshot):
#     python km-bart_amd/build.py
#     python km-bart_amd/build.py --variant diagtrb KMB_DIAG KMB_TR_BUILTIN       # the intrinsic everywhere
#     python km-bart_amd/build.py --variant diag    KMB_DIAG                       # as shipped
#     python km-bart_amd/build.py --variant diagall KMB_DIAG KMB_TR_ASM_ALL        # the candidate
# then ONE call:   gpurun --timeout 2400 -- 'bash tools/tr_asm_all_validate.sh'
# Reading the result: every md5 of a shape must be the same in the three libraries and "(stable)"; the suite must pass; then make KMB_TR_ASM_ALL the
# default (csrc/gemm.hip KMB_TR_ALL), re-run tools/gemm_tr_asm_hazards.py (the CPU test does) and the suite once more.
O=gpurun_out/tr_asm
mkdir -p $O
L=km-bart_amd/lib
python -m pytest tests/test_gemm_variants_gpu.py tests/test_gemm_persistent_gpu.py tests/test_gemm_group_gpu.py tests/test_contention_gpu.py -x -q -m gpu > $O/gemm_tests.log 2>&1
echo "rc=$?" >> $O/gemm_tests.log
tail -3 $O/gemm_tests.log
for v in 7 8 11 14 5 12 13 15 6; do
  for lib in diagtrb diag diagall; do
    [ -f $L/libkmbart_hip_$lib.so ] && KMB_LIB_PATH=$L/libkmbart_hip_$lib.so timeout 120 python tools/gemm_tr_asm_ab.py $v 2>&1 | grep -v amdgpu
  done
done > $O/ab.txt 2>&1
grep -c CHANGED $O/ab.txt
python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1
echo "rc=$?" >> $O/suite.log
tail -3 $O/suite.log
python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
