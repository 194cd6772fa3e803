"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), gfx950 corrections applied
(MI355X_MICROARCH.md section HBM: counters are in KiB; FETCH_SIZE under-reports wide coalesced reads by 2x).

    python tools/summarize_pmc.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv \
        [--json profiles/r01_pmc_traffic.json --batch 512]
"""
import collections
import csv
import json
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

if "--json" in sys.argv:
    out = sys.argv[sys.argv.index("--json") + 1]
    batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else None
    # fingerprint of the GEMM sources the passes ran on: written by the evidence run ON THE GPU BOX (tools/r6_profile.sh -> gemm_sources.sha);
    # bench.py quotes this file's traffic only when it matches the tree it runs from
    sha = sys.argv[sys.argv.index("--gemm-sha") + 1] if "--gemm-sha" in sys.argv else None
    json.dump({"per_gpu_batch": batch, "gemm_sources_sha16": sha, "gemm_launches": gemm_n,
               "gemm_hbm_bytes_per_launch": round(gemm_bytes / max(gemm_n, 1) * 1e6),
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (each with --kernel-trace "
                         "only) over `bench.py --steps 2 --warmup 2 --serial` with the GEMM tuning preloaded; counters "
                         "in KiB; FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md "
                         "section HBM); calibration: adamw_kernel measures 4231 MB = 141.04 M params x 30 B exactly",
               "kernels": {k: {"launches": n, "fetch_mb": round(fe, 1), "write_mb": round(wr, 1)}
                           for _, k, n, fe, wr in sorted(rows, reverse=True)[:16]}},
              open(out, "w"), indent=1)
