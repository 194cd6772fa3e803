"""Sustained time per launch (1 s per shape, at the board's power cap) of ONE forced launch variant (KMB_GEMM_VARIANT, argv[1]) on the step's main
forward / data-gradient shapes, plain epilogue -- against which the tuner's three-launch bursts can be compared: does the burst ranking pick the
variant that is fastest when the chip is power-limited?   for v in 7 8 11 12 13 14 15 6 9; do python tools/gemm_sustained_variants.py $v; done"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
v = sys.argv[1] if len(sys.argv) > 1 else "0"
if v != "0":
    os.environ["KMB_GEMM_VARIANT"] = v
import torch  # noqa: E402
from gpu_util import DEV, bf, gemm  # noqa: E402

SHAPES = [(65536, 3072, 768, 1), (65536, 768, 3072, 1), (65536, 2304, 768, 1), (65536, 768, 768, 1), (32768, 768, 768, 1), (32768, 3072, 768, 1),
          (65536, 768, 3072, 0), (65536, 3072, 768, 0), (65536, 768, 2304, 0), (32768, 768, 768, 0)]
torch.manual_seed(0)
out_line = ["v%-2s" % v]
for (M, N, K, bkc) in SHAPES:
    A = bf(torch.randn((M, K), device=DEV))
    B = bf(torch.randn((N, K) if bkc else (K, N), device=DEV) * 0.05)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    for _ in range(5):
        gemm(A, B, a_kc=True, b_kc=bool(bkc), out_bf16=out, tile_order=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < 1.0:
        for _ in range(50):
            gemm(A, B, a_kc=True, b_kc=bool(bkc), out_bf16=out, tile_order=1)
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    out_line.append("%7.1f" % (e0.elapsed_time(e1) / n * 1e3))
    del A, B, out
print(" ".join(out_line), flush=True)
