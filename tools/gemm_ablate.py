"""Epilogue cost of a short-K GEMM: plain, GeLU + pre-activation store, and with the stores dropped (tile_order bit 256).
The per-phase timeline comes from tools/gemm_stamps.py."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: E402
_diag.use_diag_lib()   # the A/B knobs live in the diagnostic build only (csrc/diag.h)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C  # noqa: E402
import torch  # noqa: E402
from kmbart import _lib  # noqa: E402
from kmbart._lib import KmbGemm, check, ptr  # noqa: E402
from gpu_util import DEV, bf, stream  # noqa: E402

os.environ.setdefault("KMB_GEMM_VARIANT", "7")
lib = _lib.load()
for (M, N, K) in [(16384, 3072, 768), (16384, 768, 768), (16384, 768, 3072), (16384, 2304, 768)]:
    A = bf(torch.randn((M, K), device=DEV))
    B = bf(torch.randn((N, K), device=DEV) * 0.05)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    pre = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    bias = torch.zeros(N, device=DEV)
    for label, order, act in (("full", 0, 0), ("full+gelu+preact", 0, 1), ("no-stores", 256, 0)):
        g = KmbGemm()
        g.A, g.B, g.lda, g.ldb, g.a_kc, g.b_kc = ptr(A), ptr(B), K, K, 1, 1
        g.M, g.N, g.K = M, N, K
        g.bias = ptr(bias)
        g.col_scale, g.drop_scale = 1.0, 1.0
        g.act = act
        if act == 1:
            g.preact, g.ld_preact = ptr(pre), N
        g.out_bf16, g.ld_out_bf16 = ptr(out), N
        g.tile_order = order
        for _ in range(3):
            check(lib.kmb_op_gemm(C.byref(g), stream()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            check(lib.kmb_op_gemm(C.byref(g), stream()))
        e1.record()
        torch.cuda.synchronize()
        print(f"{M} {N} {K} {label:24s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
