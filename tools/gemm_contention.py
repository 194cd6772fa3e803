"""What a GEMM launch costs while another kernel holds part of the chip (the situation of data-parallel training: RCCL's
all-reduce kernel runs on the communication stream during backward).  A helper kernel parks `--hog` one-wave workgroups
on as many CUs for 2 ms on a second stream; 100 us later the GEMM is timed on the main stream.

    hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libcuhog.so tools/cu_hog.hip
    KMB_GEMM_VARIANT=11 python tools/gemm_contention.py [--shared]      # persistent, every tile dynamic with --shared
    KMB_GEMM_VARIANT=8  python tools/gemm_contention.py                 # one workgroup per tile
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, bf, gemm  # noqa: E402
from kmbart import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hog", type=int, default=32)
ap.add_argument("--shared", action="store_true")
ap.add_argument("--lib", default="/tmp/libcuhog.so")
a = ap.parse_args()
hog = C.CDLL(a.lib)
hog.cu_hog.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p]
_lib.load().kmb_gemm_shared_device(int(a.shared))
M, N, K = 16384, 3072, 768
A = bf(torch.randn(M, K, device=DEV))
B = bf(torch.randn(N, K, device=DEV) * 0.05)
out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
for _ in range(3):
    gemm(A, B, out_bf16=out)
torch.cuda.synchronize()
side = torch.cuda.Stream()


def timed(with_hog):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        if with_hog:
            hog.cu_hog(a.hog, 200000, C.c_void_p(side.cuda_stream))      # 2 ms
            hog.cu_hog(1, 10000, C.c_void_p(torch.cuda.current_stream().cuda_stream))   # 100 us head start
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gemm(A, B, out_bf16=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


print(f"variant {os.environ.get('KMB_GEMM_VARIANT', 'auto')} shared_device={int(a.shared)}: alone {timed(False):7.1f} us, "
      f"beside {a.hog} parked workgroups {timed(True):7.1f} us")
