"""Same-box, per-shape A/B of the in-step GEMM times under two environment settings (each configuration runs
tools/gemm_shape_table.py in its own process, alternating, `rounds` times; the table shows the mean per shape, the chosen
kernel variant | tile order of each side, and the difference).

    python tools/gemm_ab_table.py "KMB_GEMM_EXCLUDE=14,15" "" [batch] [rounds]
"""
import collections
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: E402
_diag.use_diag_lib()   # the A/B knobs live in the diagnostic build only (csrc/diag.h)
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A, B = sys.argv[1], sys.argv[2]
batch = sys.argv[3] if len(sys.argv) > 3 else "1024"
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2


def run(envs, tag):
    env = dict(os.environ)
    for kv in envs.split():
        k, v = kv.split("=", 1)
        env[k] = v
    tune = os.path.join(tempfile.gettempdir(), "kmb_ab_tune_%s.txt" % tag)
    if os.path.exists(tune):
        os.remove(tune)
    env["KMB_GEMM_TUNE_FILE"] = tune
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_shape_table.py"), batch], env=env, capture_output=True,
                       text=True)
    rows = {}
    for line in r.stdout.splitlines():
        if "|" in line and line[0].isdigit():
            left, right = line.split("|")
            v, M, N, K, sp, act = left.split()
            n, us, tf, share = right.split()
            rows[(int(v), int(M), int(N), int(K), int(sp[1:]), int(act[1:]))] = (int(n), float(us))
    picks = {}
    if os.path.exists(tune):
        for line in open(tune):
            akc, bkc, M, N, K, sp, act, best = (int(x) for x in line.split())
            picks[(akc * 2 + bkc, M, N, K, sp, act)] = "v%d|o%d" % (best & 15, best >> 4)
    return rows, picks


acc = {"A": collections.defaultdict(list), "B": collections.defaultdict(list)}
picks = {}
for i in range(rounds):
    for tag, envs in (("A", A), ("B", B)):
        rows, pk = run(envs, tag)
        picks[tag] = pk
        for k, (n, us) in rows.items():
            acc[tag][k].append((n, us))
print("A = [%s]   B = [%s]   batch %s, %d rounds" % (A, B, batch, rounds))
print("layout M N K split act | n | A us (pick) | B us (pick) | B/A")
ta = tb = 0.0
for k in sorted(acc["A"], key=lambda k: -acc["A"][k][0][0] * acc["A"][k][0][1]):
    n = acc["A"][k][0][0]
    a = sum(u for _, u in acc["A"][k]) / len(acc["A"][k])
    b = sum(u for _, u in acc["B"][k]) / len(acc["B"][k]) if k in acc["B"] else float("nan")
    ta += n * a
    tb += n * b
    print("%d %6d %6d %6d s%-2d a%d | %2d | %8.1f %-7s | %8.1f %-7s | %5.3f" % (*k, n, a, picks["A"].get(k, "-"), b, picks["B"].get(k, "-"), b / a))
print("total GEMM ms: A %.3f  B %.3f  (B/A %.4f)" % (ta / 1e3, tb / 1e3, tb / ta))
