"""Fixed cost per launch vs K-proportional cost: time (M, N, K) for several K (no in-kernel instrumentation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import DEV, bf, gemm
M, N = int(sys.argv[1]), int(sys.argv[2])
print("variant", os.environ.get("KMB_GEMM_VARIANT", "auto"), "M", M, "N", N)
for K in (64, 128, 256, 384, 768, 1536, 3072):
    A = bf(torch.randn((M, K), device=DEV)); B = bf(torch.randn((N, K), device=DEV) * 0.05)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    for _ in range(3): gemm(A, B, out_bf16=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): gemm(A, B, out_bf16=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"  K={K:5d}  {us:8.1f} us   {2.0*M*N*K/us/1e6:7.1f} TF")
