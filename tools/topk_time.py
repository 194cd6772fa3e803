"""Time of the log-softmax top-k at the decode shape (320 rows x 50320) for several k: one workgroup per row
(kmb_logsoftmax_topk) and the split form the decode loop uses (kmb_logsoftmax_topk_ws)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import DEV, ptr, stream, check
from kmbart import _lib
lib = _lib.load()
rows, V, ld = 320, 50320, 50432
logits = torch.randn(rows, ld, device=DEV) * 4
add = torch.randn(rows, device=DEV)
scr = torch.empty(int(lib.kmb_logsoftmax_topk_scratch(rows)), device=DEV)
for k, ft in ((1, -1), (2, -1), (5, -1), (10, -1)):
    val = torch.empty((rows, k), device=DEV); idx = torch.empty((rows, k), dtype=torch.int32, device=DEV)
    for name, call in (("one workgroup per row", lambda: lib.kmb_logsoftmax_topk(ptr(logits), ld, V, rows, ptr(add), ft, -1, k, ptr(val), ptr(idx), stream())),
                       ("split + combine", lambda: lib.kmb_logsoftmax_topk_ws(ptr(logits), ld, V, rows, ptr(add), ft, -1, k, ptr(val), ptr(idx), ptr(scr), scr.numel(), stream()))):
        for _ in range(3): check(call())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): check(call())
        e1.record(); torch.cuda.synchronize()
        print(f"k {k:2d}  {name:22s} {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us")

# the decode loop's beam step (64 items x 5 beams, k = 10): two launches over the logits (kmb_beam_step) against one launch that selects
# from the per-block statistics the all-rows vocabulary projection leaves (kmb_beam_step_stats; the statistics come from a real projection)
import ctypes as C  # noqa: E402
from gpu_util import bf  # noqa: E402
from kmbart._lib import KmbGemm  # noqa: E402
B, nb, k = 64, 5, 10
g = torch.Generator(device=DEV).manual_seed(1)
A = bf(torch.randn(rows, 768, device=DEV, generator=g) * 0.5)
W = bf(torch.randn(ld, 768, device=DEV, generator=g) * 0.05)
bias = torch.randn(V, device=DEV, generator=g)
p = KmbGemm()
p.A, p.B, p.lda, p.ldb, p.a_kc, p.b_kc = ptr(A), ptr(W), 768, 768, 1, 1
p.M, p.N, p.K, p.bias, p.col_scale, p.drop_scale = rows, V, 768, ptr(bias), 1.0, 1.0
p.out_f32, p.ld_out_f32 = ptr(logits), ld
stats = torch.empty(int(lib.kmb_op_gemm_allrows_stats_floats(V)), device=DEV)
check(lib.kmb_op_gemm_allrows_stats(C.byref(p), ptr(stats), stream()))
cand = torch.empty((B, k, 2), dtype=torch.int32, device=DEV)
ns = torch.empty(rows, device=DEV); nt = torch.empty(rows, dtype=torch.int64, device=DEV); ni = torch.empty(rows, dtype=torch.int32, device=DEV)
for name, call in (("beam step, two launches over the logits", lambda: lib.kmb_beam_step(ptr(logits), ld, V, B, nb, ptr(add), -1, -1, k, ptr(cand), 2, ptr(ns), ptr(nt), ptr(ni), ptr(scr), scr.numel(), stream())),
                   ("beam step from the block statistics", lambda: lib.kmb_beam_step_stats(ptr(logits), ld, V, B, nb, ptr(add), -1, -1, k, ptr(cand), 2, ptr(ns), ptr(nt), ptr(ni), ptr(stats), (V + 255) // 256, stream()))):
    for _ in range(3): check(call())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): check(call())
    e1.record(); torch.cuda.synchronize()
    print(f"k {k:2d}  {name:42s} {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us")
