"""Aggregate the three PMC passes of tools/pmc_l2_latency.sh by kernel name."""
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


lat, n1 = load("lat")
hit, n2 = load("hit")
st, n3 = load("stall")
print("| kernel | launches | L1->L2 read requests / launch | mean latency (cycles) | L2 hit rate | tag stall cycles / busy cycles |")
print("|---|---|---|---|---|---|")
names = sorted(lat, key=lambda k: -lat[k].get("TCP_TCC_READ_REQ_LATENCY_sum", 0))
for k in names[:24]:
    req = lat[k].get("TCP_TCC_READ_REQ_sum", 0.0)
    la = lat[k].get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0)
    h, m = hit[k].get("TCC_HIT_sum", 0.0), hit[k].get("TCC_MISS_sum", 0.0)
    ts, tb = st[k].get("TCC_TAG_STALL_sum", 0.0), st[k].get("TCC_BUSY_sum", 0.0)
    print("| `%s` | %d | %.3g | %s | %s | %s |" % (k[:60], n1[k], req / max(n1[k], 1), "%.0f" % (la / req) if req else "-",
                                             "%.2f" % (h / (h + m)) if h + m else "-", "%.2f" % (ts / tb) if tb else "-"))
