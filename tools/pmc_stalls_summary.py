"""Aggregate the PMC passes of tools/pmc_stalls.sh by kernel name: every counter as a share of the kernel's wave cycles
(SQ_WAVE_CYCLES: cycles x resident waves, summed over the chip) or per instruction."""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]


def load(sub):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (root, sub), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    seen = set()
    if not f:
        return agg, n
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        k = re.sub(r"\((?:Kmb|const|unsigned|float|int|long|at::).*", "", k)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key)
            n[k] += 1
    return agg, n


wait, n1 = load("wait")
vmem, _ = load("vmem")
lds, _ = load("lds")
valu, _ = load("valu")
names = [k for k in sorted(wait, key=lambda k: -wait[k].get("SQ_WAVE_CYCLES", 0)) if "gemm" in k][:14]


def sh(a, b):
    return "%.1f %%" % (100.0 * a / b) if b else "-"


print("Shares of SQ_WAVE_CYCLES (wave-resident cycles) unless noted; raw counters are summed over the chip and the launches of a step.\n")
print("| kernel | launches | wait: any instruction | wait: LDS | wait any | issuing (ACTIVE_INST_ANY) |")
print("|---|---|---|---|---|---|")
for k in names:
    w = wait[k]
    wc = w.get("SQ_WAVE_CYCLES", 0.0)
    print("| `%s` | %d | %s | %s | %s | %s |" % (k[:58], n1[k], sh(w.get("SQ_WAIT_INST_ANY", 0), wc), sh(w.get("SQ_WAIT_INST_LDS", 0), wc),
                                            sh(w.get("SQ_WAIT_ANY", 0), wc), sh(w.get("SQ_ACTIVE_INST_ANY", 0), wc)))
print("\n| kernel | VMEM instructions / launch | cycles per VMEM instruction (INST_CYCLES_VMEM / INSTS_VMEM) | INST_CYCLES_VMEM share | TA address FIFO full share | TA command FIFO full share |")
print("|---|---|---|---|---|---|")
for k in names:
    v = vmem[k]
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    ni = v.get("SQ_INSTS_VMEM", 0.0)
    print("| `%s` | %.3g | %s | %s | %s | %s |" % (k[:58], ni / max(n1[k], 1), "%.0f" % (v.get("SQ_INST_CYCLES_VMEM", 0) / ni) if ni else "-",
                                              sh(v.get("SQ_INST_CYCLES_VMEM", 0), wc), sh(v.get("SQ_VMEM_TA_ADDR_FIFO_FULL", 0), wc),
                                              sh(v.get("SQ_VMEM_TA_CMD_FIFO_FULL", 0), wc)))
print("\n| kernel | LDS instructions / launch | LDS_IDX_ACTIVE / launch | bank conflict cycles / IDX_ACTIVE | LDS cmd FIFO full / IDX_ACTIVE | LDS data FIFO full / IDX_ACTIVE |")
print("|---|---|---|---|---|---|")
for k in names:
    l = lds[k]
    ia = l.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print("| `%s` | %.3g | %.3g | %s | %s | %s |" % (k[:58], l.get("SQ_INSTS_LDS", 0) / max(n1[k], 1), ia / max(n1[k], 1), sh(l.get("SQ_LDS_BANK_CONFLICT", 0), ia),
                                                sh(l.get("SQ_LDS_CMD_FIFO_FULL", 0), ia), sh(l.get("SQ_LDS_DATA_FIFO_FULL", 0), ia)))
print("\n| kernel | MFMA busy cycles / GUI_ACTIVE x 8 / 1024 SIMDs | MFMA instructions / launch | VALU instructions / launch | INST_CYCLES_VALU / INSTS_VALU |")
print("|---|---|---|---|---|")
for k in names:
    a = valu[k]
    gui = a.get("GRBM_GUI_ACTIVE", 0.0)
    nv = a.get("SQ_INSTS_VALU", 0.0)
    print("| `%s` | %s | %.3g | %.3g | %s |" % (k[:58], sh(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), gui / 8 * 1024), a.get("SQ_INSTS_MFMA", 0) / max(n1[k], 1),
                                           nv / max(n1[k], 1), "%.1f" % (a.get("SQ_INST_CYCLES_VALU", 0) / nv) if nv else "-"))
