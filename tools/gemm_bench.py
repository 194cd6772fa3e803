"""GEMM microbenchmark for one kernel variant (KMB_GEMM_VARIANT=1, 7 or 8): correctness vs torch + TFLOP/s per shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, bf, gemm, rel_err  # noqa: E402

torch.manual_seed(0)
SHAPES = [  # (M, N, K, a_kc, b_kc)
    (16384, 3072, 768, True, True), (16384, 768, 3072, True, True), (16384, 2304, 768, True, True),
    (16384, 768, 768, True, True), (8192, 768, 768, True, True), (8192, 768, 50432, True, False),
    (16384, 768, 3072, True, False), (16384, 3072, 768, True, False), (3072, 768, 16384, False, False),
    (768, 768, 16384, False, False), (50320, 768, 8192, False, False), (4096, 4096, 4096, True, True),
]
print("variant", os.environ.get("KMB_GEMM_VARIANT", "auto"))
for (M, N, K, akc, bkc) in SHAPES:
    A = bf(torch.randn((M, K) if akc else (K, M), device=DEV))
    B = bf(torch.randn((N, K) if bkc else (K, N), device=DEV) * 0.05)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    gemm(A, B, a_kc=akc, b_kc=bkc, out_bf16=out)
    Af = A.float() if akc else A.float().t()
    Bf = B.float() if bkc else B.float().t()
    ref = (Af[:512] @ Bf.t())
    err = rel_err(out[:512], ref)
    for _ in range(3):
        gemm(A, B, a_kc=akc, b_kc=bkc, out_bf16=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        gemm(A, B, a_kc=akc, b_kc=bkc, out_bf16=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{M:6d} {N:6d} {K:6d} akc={int(akc)} bkc={int(bkc)}  {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TF  err {err:.1e}", flush=True)
