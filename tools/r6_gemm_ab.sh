# Round 6: ONE gpurun call that checks a GEMM-side change against the round's reference bits and times it inside a step on the SAME box.
#   before (build container):  cp km-bart_amd/lib/libkmbart_hip.so km-bart_amd/lib/libkmbart_hip_prev.so   (the library to compare with)
#                              python km-bart_amd/build.py && python km-bart_amd/build.py --variant diag KMB_DIAG
#   gpurun --timeout 2400 -- 'bash tools/r6_gemm_ab.sh [tag]'
# 1. md5 of seven shapes x every launch variant (diag build of the tree) against profiles/r06_gemm_transposing_reads_all_variants.txt (the
#    intrinsic build's bits, identical in all three libraries of the round's first call);  2. the GEMM bitwise / contention tests;
# 3. bench.py (training legs only) alternating prev / new library, two rounds, with the held clock in every line.
T=${1:-ab}
O=gpurun_out/r6_$T
mkdir -p $O
L=km-bart_amd/lib
for v in 7 8 11 14 5 12 13 15 6 9; do
  KMB_LIB_PATH=$L/libkmbart_hip_diag.so timeout 150 python tools/gemm_tr_asm_ab.py $v 2>&1 | grep -v amdgpu
done > $O/md5.txt 2>&1
python - <<PY > $O/md5_check.txt 2>&1
import re
ref = {}
for l in open("profiles/r06_gemm_transposing_reads_all_variants.txt"):
    m = re.match(r"\s*(\d+)\s+(\d+)\s+(\d+) akc=(\d) bkc=(\d).*md5 (\w+)", l)
    if m: ref[m.group(1, 2, 3, 4, 5)] = m.group(6)
bad = n = 0
for l in open("$O/md5.txt"):
    m = re.match(r"\s*(\d+)\s+(\d+)\s+(\d+) akc=(\d) bkc=(\d).*md5 (\w+) (\(\w+)", l)
    if m:
        n += 1
        if ref.get(m.group(1, 2, 3, 4, 5)) != m.group(6) or m.group(7) != "(stable": bad += 1; print("MISMATCH", l.strip())
print("checked", n, "lines, mismatches", bad)
PY
tail -2 $O/md5_check.txt
python -m pytest tests/test_gemm_variants_gpu.py tests/test_gemm_persistent_gpu.py tests/test_gemm_group_gpu.py tests/test_contention_gpu.py -x -q -m gpu > $O/gemm_tests.log 2>&1
echo "rc=$?" >> $O/gemm_tests.log
tail -3 $O/gemm_tests.log
B="python bench.py --no-cpu-baseline --no-pcie --no-extras"
for i in 1 2; do
  KMB_LIB_PATH=$L/libkmbart_hip_prev.so timeout 300 $B > $O/bench_prev_$i.json 2> $O/bench_prev_$i.err
  timeout 300 $B > $O/bench_new_$i.json 2> $O/bench_new_$i.err
done
python - <<PY | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f.split("/")[-1], d["ms_per_step"], "ms", d["value"], "tok/s", d.get("clock_mhz"), "MHz  gemm", r.get("gemm_ms_per_step"), "ms frac", r.get("frac"),
              {k: v["tflops"] for k, v in (r.get("by_variant") or {}).items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
