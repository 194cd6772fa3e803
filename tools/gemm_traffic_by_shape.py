"""Per-shape HBM-side traffic of the GEMM launches of one training step: joins the launch list of
tools/one_step_gemm_trace.py (launch order) with the per-dispatch FETCH_SIZE / WRITE_SIZE of two rocprofv3 --pmc passes
over that same script (the last len(list) gemm_kernel dispatches of each pass are the listed step, in order).
gfx950 corrections as in tools/summarize_pmc.py (KiB units; FETCH_SIZE x 2).

    python tools/gemm_traffic_by_shape.py launches.txt fetch_counter_collection.csv write_counter_collection.csv
"""
import collections
import csv
import sys


def gemm_dispatches(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "gemm_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


launches = [l.split() for l in open(sys.argv[1])]
f = gemm_dispatches(sys.argv[2], "FETCH_SIZE")[-len(launches):]
w = gemm_dispatches(sys.argv[3], "WRITE_SIZE")[-len(launches):]
assert len(f) == len(launches) == len(w), (len(f), len(w), len(launches))
agg = collections.OrderedDict()
for (v, M, N, K, sp, act, us, *rest), rf, rw in zip(launches, f, w):
    key = (int(v), int(M), int(N), int(K), int(sp), int(act) + (16 if rest and int(rest[0]) else 0))   # + 16: a residual operand is read
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0, rf["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "")])
    a[0] += 1
    a[1] += float(rf["Counter_Value"]) * 1024 * 2 / 1e6
    a[2] += float(rw["Counter_Value"]) * 1024 / 1e6
    a[3] += float(us)
print("layout(3=fwd,2=dgrad,0=wgrad) M N K split act (r: + residual operand) | n | fetch MB | write MB | algorithmic read / write MB | ratio | us | kernel")
tot_meas = tot_alg = 0.0
for (v, M, N, K, sp, act), (n, fe, wr, us, kern) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    out_b = 4 if v == 0 else 2
    res, act = act >= 16, act % 16
    alg_r = 2.0 * (M * K + N * K) / 1e6 + (M * N * 2 / 1e6 if act == 2 else 0) + (M * N * 2 / 1e6 if res else 0)
    alg_w = out_b * M * N / 1e6 + (M * N * 2 / 1e6 if act == 1 else 0)
    if sp > 1:
        alg_w = sp * M * N * 4 / 1e6   # split-K writes fp32 slabs (the reduce pass is another kernel)
    tot_meas += fe + wr
    tot_alg += n * (alg_r + alg_w)
    print(f"{v} {M:6d} {N:6d} {K:6d} s{sp:<2d} a{act}{'r' if res else ' '} | {n:2d} | {fe / n:8.1f} | {wr / n:8.1f} | {alg_r:7.1f} / {alg_w:7.1f} | "
          f"{(fe + wr) / n / (alg_r + alg_w):5.2f} | {us / n:7.1f} | {kern[:44]}")
print(f"total measured {tot_meas / 1e3:.2f} GB, algorithmic {tot_alg / 1e3:.2f} GB, ratio {tot_meas / tot_alg:.2f}, per launch "
      f"{tot_meas / len(launches):.1f} MB")
