"""Cold vs warm operand timing of one GEMM shape: back-to-back launches re-read A / B from the Infinity Cache; inside a
training step the A operand was just streamed out by the previous kernel and comes from HBM.  A 600 MB fill between
launches evicts L2 and the Infinity Cache.  python tools/gemm_cold_warm.py M N K [a_kc b_kc [split_k]]
(split_k > 1: raw fp32 slabs, the weight-gradient form, e.g. 3072 768 32768 0 0 7)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, bf, gemm  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
akc = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
bkc = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
split = int(sys.argv[6]) if len(sys.argv) > 6 else 0
torch.manual_seed(0)
A = bf(torch.randn((M, K) if akc else (K, M), device=DEV))
B = bf(torch.randn((N, K) if bkc else (K, N), device=DEV))
out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
SK = dict(split_k=split, slab=torch.empty(split * M * N, dtype=torch.float32, device=DEV)) if split > 1 else dict(out_bf16=out)
junk = torch.empty(600 << 20, dtype=torch.uint8, device=DEV)
for _ in range(3):
    gemm(A, B, a_kc=akc, b_kc=bkc, **SK)
torch.cuda.synchronize()


def timed(cold):
    ts = []
    for _ in range(10):
        if cold:
            junk.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gemm(A, B, a_kc=akc, b_kc=bkc, **SK)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def timed_after_write():
    """A is (re)written by an ordinary streaming kernel right before the launch, after a cache flush: is a freshly
    WRITTEN operand served from the Infinity Cache?"""
    A2 = A.clone()
    ts = []
    for _ in range(10):
        junk.fill_(1)
        A.copy_(A2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gemm(A, B, a_kc=akc, b_kc=bkc, **SK)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


w, c = timed(False), timed(True)
print(f"  A just written by a plain-store kernel after the flush: {timed_after_write():.1f} us")
fl = 2.0 * M * N * K
print(f"{M}x{N}x{K} akc={int(akc)} bkc={int(bkc)}: warm {w:.1f} us ({fl / w * 1e-6:.0f} TF)  cold {c:.1f} us ({fl / c * 1e-6:.0f} TF)")
