import os, sys
sys.path.insert(0, "tools"); import _diag; _diag.use_diag_lib()
sys.path.insert(0, "km-bart_amd"); sys.path.insert(0, "tests")
import torch
from gpu_util import DEV, bf, gemm
M, N = 65536, 3072
res = {}
for K in (768, 1536, 3072):
    A = bf(torch.randn(M, K, device=DEV)) * 0.5
    B = bf(torch.randn(N, K, device=DEV)) * 0.5
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    bias = torch.randn(N, device=DEV)
    for _ in range(3):
        gemm(A, B, out_bf16=out, bias=bias, tile_order=3 | 512)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gemm(A, B, out_bf16=out, bias=bias, tile_order=3 | 512); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    res[K] = sorted(ts)[2]
    print("v%s K=%d: %.1f us = %.2f us per tile (12 tiles per workgroup), %.0f TF" % (os.environ["KMB_GEMM_VARIANT"], K, res[K], res[K] / 12, 2.0 * M * N * K / res[K] * 1e-6))
step = (res[3072] - res[768]) / 12 / 36
print("per 64-deep step %.2f us; per-tile overhead beyond its steps %.2f us" % (step, res[768] / 12 - 12 * step))
