"""K-loop rate of a forced GEMM variant on the benchmark-batch forward shapes: back-to-back launches with the epilogue skipped
(diagnostic build: KMB_GEMM_ABLATE), optionally from a timing-only library (KMB_LIB_PATH, e.g. `build.py --variant v11nodma KMB_DIAG
KMB_V11_NODMA`: the loop without its LDS-DMA).  Forced variants bypass launch_config, so the tile order / prefetch bits are given here.

    KMB_GEMM_VARIANT=11 python tools/kloop_time.py
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: E402
_diag.use_diag_lib()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, bf, gemm  # noqa: E402

SHAPES = [("qkv_fwd", 65536, 2304, 768), ("out_fwd", 65536, 768, 768), ("fc1", 65536, 3072, 768), ("fc2_fwd", 65536, 768, 3072),
          ("dec_out_fwd", 32768, 768, 768)]
ORDER = 1 | 2   # per-XCD ranges + L2 prefetch, as launch_config sets them
for name, M, N, K in SHAPES:
    A = bf(torch.randn(M, K, device=DEV)) * 0.5
    B = bf(torch.randn(N, K, device=DEV)) * 0.5
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    bias = torch.randn(N, device=DEV)
    res = []
    for ablate in (0, 512):
        for _ in range(3):
            gemm(A, B, out_bf16=out, bias=bias, tile_order=ORDER | ablate)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gemm(A, B, out_bf16=out, bias=bias, tile_order=ORDER | ablate)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res.append(ts[len(ts) // 2])
    fl = 2.0 * M * N * K
    print("%-12s %6dx%5dx%5d | v%s with bias epilogue %7.1f us %5.0f TF | K loops only %7.1f us %5.0f TF" %
          (name, M, N, K, os.environ.get("KMB_GEMM_VARIANT", "?"), res[0], fl / res[0] * 1e-6, res[1], fl / res[1] * 1e-6))
