"""Is a persistent GEMM's epilogue bound inside the CU or by all 256 workgroups storing at once?  (diagnostic build only)

The 256 workgroups of a persistent launch start together and run tiles of equal length, so they reach their epilogues
together: at 65536 x 3072 x 768 with the GeLU + GeLU' epilogue every round ends in a 64 MB burst of stores after a K loop
that leaves the memory system nearly idle.  Two experiments on that launch (and on the fc2 data-gradient class):

  1. fewer persistent workgroups (KMB_GEMM_GRID = 256 .. 32; the tiles are handed out dynamically, so a smaller grid just
     takes more rounds): time x grid / tiles = time per tile; with the epilogue and with it skipped (KMB_GEMM_ABLATE).  An
     epilogue that is CU-side work costs the same per tile at every grid; a burst-bound one shrinks with the grid.
  2. the full grid started in four groups KMB_GEMM_STAGGER half-microseconds apart, so that a quarter of the chip is in
     its epilogue at any time.

    python tools/epilogue_burst.py [M]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: E402
_diag.use_diag_lib()
os.environ["KMB_GEMM_ABLATE_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, bf, gemm  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
torch.manual_seed(0)


def case(name):
    if name == "fc1 forward + GeLU + GeLU'":
        N, K = 3072, 768
        kw = dict(bias=torch.randn(N, device=DEV), act=1, preact=torch.empty((M, N), dtype=torch.bfloat16, device=DEV))
    elif name == "fc2 data gradient x GeLU' + column sums":
        N, K = 3072, 768
        kw = dict(act=2, aux=bf(torch.randn(M, N, device=DEV)), colsum=torch.zeros(((M + 63) // 64, N), device=DEV))
    elif name == "out-projection + bias + residual + dropout":
        N, K = 768, 768
        kw = dict(bias=torch.randn(N, device=DEV), residual=bf(torch.randn(M, N, device=DEV)), drop_p=0.1, drop_seed=7)
    else:
        N, K = 2304, 768
        kw = dict(bias=torch.randn(N, device=DEV))
    A = bf(torch.randn(M, K, device=DEV))
    B = bf(torch.randn(N, K, device=DEV) * 0.05)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    return N, K, lambda: gemm(A, B, out_bf16=out, **kw)


def time_us(run, reps=7):
    for _ in range(2):
        run()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for e0, e1 in evs:
        e0.record()
        run()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
    return ts[len(ts) // 2]


def setenv(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


print("# tools/epilogue_burst.py  M = %d, the tuner's pick per shape (a persistent 256x256 variant on all four), back-to-back launches, median of 7" % M)
for name in ("fc1 forward + GeLU + GeLU'", "fc2 data gradient x GeLU' + column sums", "out-projection + bias + residual + dropout",
             "qkv forward + bias"):
    N, K, run = case(name)
    setenv(KMB_GEMM_GRID=None, KMB_GEMM_ABLATE=None, KMB_GEMM_STAGGER=None)
    run()                       # the tuner ranks its candidates on this first launch, before any knob is set
    torch.cuda.synchronize()
    tiles = (M // 256) * (N // 256)
    print("\n%s: %d x %d x %d, %d tiles" % (name, M, N, K, tiles))
    print("  grid | launch us | without epilogue | per tile us: with / without / epilogue")
    for g in (256, 192, 128, 64, 32):
        setenv(KMB_GEMM_GRID=g, KMB_GEMM_ABLATE=None, KMB_GEMM_STAGGER=None)
        t1 = time_us(run)
        setenv(KMB_GEMM_ABLATE=1)
        t0 = time_us(run)
        rounds = -(-tiles // g)
        print("  %4d | %9.1f | %9.1f | %6.2f / %6.2f / %6.2f   (%d rounds)" % (g, t1, t0, t1 / rounds, t0 / rounds, (t1 - t0) / rounds, rounds))
    setenv(KMB_GEMM_GRID=None, KMB_GEMM_ABLATE=None)
    base = time_us(run)
    row = []
    for s in (4, 8, 12, 16, 20, 28):
        setenv(KMB_GEMM_STAGGER=s)
        row.append("%d: %.1f" % (s, time_us(run)))
    setenv(KMB_GEMM_STAGGER=None)
    again = time_us(run)
    print("  full grid, start stagger (half-us between the four groups): none %.1f us | %s | none again %.1f" % (base, " | ".join(row), again))
