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
