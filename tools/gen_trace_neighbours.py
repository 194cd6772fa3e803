"""What runs between the decode blocks: from an ORDERED rocprofv3 kernel trace (`--kernel-trace`, the *_kernel_trace.csv) of tools/gen_bench.py,
the kernels that are neither decode blocks nor the vocabulary projection / top-k, each with the kernel dispatched before and after it and the gap
to them -- the ~70 `__amd_rocclr_copyBuffer` launches per generate of profiles/r05_generation_kernel_stats.md were unexplained.

    python tools/gen_trace_neighbours.py gpurun_out/r06/prof_gen/.../g_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]


pat = collections.Counter()
tot = collections.Counter()
for i, r in enumerate(rows):
    n = short(r["Kernel_Name"])
    if n.startswith("decode_") or n.startswith("gemm_kernel") or n.startswith("topk") or n.startswith("beam_"):
        continue
    prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
    nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
    pat[(prev, n, nxt)] += 1
    tot[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("dispatches", len(rows), " span ms", (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6)
print("| before | kernel | after | count |")
print("|---|---|---|---|")
for (p, n, x), c in pat.most_common(40):
    print("| %s | %s | %s | %d |" % (p, n, x, c))
print()
for n, t in tot.most_common(12):
    print("%-50s %.3f ms in all" % (n, t / 1e6))
