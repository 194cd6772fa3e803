"""Turns `rocprofv3 --kernel-trace --stats --output-format csv` output into the table committed under profiles/.

    python tools/summarize_rocprof.py gpurun_out/prof_X/X_kernel_stats.csv STEPS > profiles/rNN_kernel_stats.md
"""
import csv
import sys

path, steps = sys.argv[1], int(sys.argv[2])
rows = [r for r in csv.DictReader(open(path))]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("| kernel | calls | calls/step | total ms | avg us | min us | max us | share |")
print("|---|---|---|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0] if not name.startswith("__amd") else name
    t = float(r["TotalDurationNs"])
    if t / tot < 0.0005:
        continue
    print(f"| `{name[:70]}` | {r['Calls']} | {int(r['Calls']) / steps:.1f} | {t / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | "
          f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {100 * t / tot:.1f}% |")
print(f"\nall kernels: {tot / 1e6:.3f} ms over {steps} steps = {tot / 1e6 / steps:.3f} ms/step of GPU kernel time")
