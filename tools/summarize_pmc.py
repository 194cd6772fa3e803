"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), gfx950 corrections applied
(MI355X_MICROARCH.md section HBM: counters are in KiB; FETCH_SIZE under-reports wide coalesced reads by 2x).

    python tools/summarize_pmc.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv
"""
import collections
import csv
import sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = agg[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
print("| kernel | launches | fetch MB/launch (x2 corrected) | write MB/launch | HBM MB/launch |")
print("|---|---|---|---|---|")
rows = []
for k in f:
    n = f[k][0]
    fe = f[k][1] * 1024 * 2 / n / 1e6
    wr = w.get(k, [1, 0.0])[1] * 1024 / max(w.get(k, [1, 0.0])[0], 1) / 1e6
    rows.append((n * (fe + wr), k, n, fe, wr))
gemm_bytes, gemm_n = 0.0, 0
for tot, k, n, fe, wr in sorted(rows, reverse=True)[:16]:
    print(f"| `{k[:60]}` | {n} | {fe:.1f} | {wr:.1f} | {fe + wr:.1f} |")
    if k.startswith("gemm_kernel"):
        gemm_bytes += tot
        gemm_n += n
print(f"\nGEMM kernels: {gemm_n} launches, {gemm_bytes / max(gemm_n, 1):.1f} MB of HBM traffic per launch on average")
