"""MFMA-busy fraction and held clock per kernel from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES, with --kernel-trace only).

    python tools/summarize_pmc_mfma.py gpurun_out/r02/pmc_mfma/m_counter_collection.csv > profiles/r02_pmc_mfma_util_b512.md

MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / kernel
duration (MI355X_MICROARCH.md 'DVFS give-back': reads a few % high on sub-ms dispatches)."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    key = (name, r["Dispatch_Id"])
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[key] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
per = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for (name, _), c in agg.items():
    p = per[name]
    p[0] += 1
    p[1] += disp[(name, _)]
    p[2] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    p[3] += c.get("GRBM_GUI_ACTIVE", 0.0)
print(__doc__.split("\n\n")[2].strip() + "\n")
print("| kernel | launches | avg us | clock GHz | MFMA busy |")
print("|---|---|---|---|---|")
tot_busy = tot_cyc = 0.0
for name, (n, us, busy, gui) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    if not name.startswith(("gemm_kernel", "attn_")) or us <= 0 or gui <= 0:
        continue
    cyc = gui / 8.0
    print(f"| `{name[:60]}` | {n} | {us / n:.1f} | {cyc / (us * 1e3):.2f} | {100.0 * busy / (cyc * 1024):.1f} % |")
    if name.startswith("gemm_kernel"):
        tot_busy += busy
        tot_cyc += cyc * 1024
print(f"\nAll GEMM launches together: matrix pipe busy {100.0 * tot_busy / tot_cyc:.1f} % of the SIMD cycles.")
