"""Turns `rocprofv3 --kernel-trace --stats --output-format csv` output into the table committed under profiles/.

    python tools/summarize_rocprof.py gpurun_out/prof_X/X_kernel_stats.csv STEPS > profiles/rNN_kernel_stats.md

STEPS: a number, or `auto` (training runs): taken FROM THE RUN -- `embed_bwd_kernel` is launched exactly twice per training step (encoder and decoder
embedding), so steps = its calls / 2 (`auto:<kernel substring>:<launches per step>` for another anchor).  Round 5's committed tables divided a
65-step run (bench.py times three windows: warm-up + 3 x --steps) by a hard-coded 25 and were wrong per step by 2.6 x (Weak 4).
"""
import csv
import sys

path = sys.argv[1]
rows = [r for r in csv.DictReader(open(path))]
if sys.argv[2].startswith("auto"):
    parts = sys.argv[2].split(":")
    anchor, per_step = (parts[1], int(parts[2])) if len(parts) == 3 else ("embed_bwd_kernel", 2)
    calls = sum(int(r["Calls"]) for r in rows if anchor in r["Name"])
    if calls == 0 or calls % per_step:
        raise SystemExit("cannot derive the step count: %d launches of %s, %d expected per step" % (calls, anchor, per_step))
    steps = calls // per_step
else:
    steps = int(sys.argv[2])
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
