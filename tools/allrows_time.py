"""Vocabulary projection of a decode step (R x 50320 x 768, fp32 logits): the all-rows kernel against the tuner's pick."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, bf, gemm  # noqa: E402

for R in (320, 160, 192):
    g = torch.Generator(device=DEV).manual_seed(1)
    A = bf(torch.randn(R, 768, device=DEV, generator=g) * 0.5)
    B = bf(torch.randn(50432, 768, device=DEV, generator=g) * 0.05)
    bias = torch.randn(50320, device=DEV, generator=g)
    out = torch.empty((R, 50432), dtype=torch.float32, device=DEV)
    for allrows in (False, True):
        for _ in range(3):
            gemm(A, B, N=50320, bias=bias, out_f32=out, allrows=allrows)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            gemm(A, B, N=50320, bias=bias, out_f32=out, allrows=allrows)
        e1.record()
        torch.cuda.synchronize()
        print(f"R={R} {'all-rows' if allrows else 'tuner   '}: {e0.elapsed_time(e1) * 50.0:7.1f} us")
