"""A/B for the transposing LDS reads as inline asm (csrc/gemm.hip kmb_tr_read_asm: no compiler-made vmcnt(0) behind the LDS-DMA issue) against the
intrinsic (-DKMB_TR_BUILTIN): time and md5 of the outputs of one launch variant (argv[1], default 8 = the 256 x 256 kernel) on weight-gradient /
data-gradient layouts.  Run once per library and compare:
    python km-bart_amd/build.py --variant diag KMB_DIAG;  python km-bart_amd/build.py --variant diagtrb KMB_DIAG KMB_TR_BUILTIN      (no GPU needed)
    KMB_LIB_PATH=km-bart_amd/lib/libkmbart_hip_diagtrb.so python tools/gemm_tr_asm_ab.py 8
    KMB_LIB_PATH=km-bart_amd/lib/libkmbart_hip_diag.so    python tools/gemm_tr_asm_ab.py 8"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["KMB_GEMM_VARIANT"] = sys.argv[1] if len(sys.argv) > 1 else "8"
import torch  # noqa: E402
from gpu_util import DEV, bf, gemm  # noqa: E402

torch.manual_seed(0)
SHAPES = [  # (M, N, K, a_kc, b_kc)
    (4096, 4096, 8192, False, False), (3072, 768, 65536, False, False), (768, 768, 32768, False, False),
    (4096, 4096, 8192, True, False), (65536, 768, 3072, True, False), (65536, 3072, 768, True, False), (4096, 4096, 8192, True, True),
]
if "--short" in sys.argv:
    SHAPES = [SHAPES[0], SHAPES[1], SHAPES[4], SHAPES[5]]
print("library", os.environ.get("KMB_LIB_PATH", "product"), "variant", os.environ["KMB_GEMM_VARIANT"])
for (M, N, K, akc, bkc) in SHAPES:
    A = bf(torch.randn((M, K) if akc else (K, M), device=DEV))
    B = bf(torch.randn((N, K) if bkc else (K, N), device=DEV) * 0.05)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    for _ in range(3):
        gemm(A, B, a_kc=akc, b_kc=bkc, out_bf16=out)
    torch.cuda.synchronize()
    h = hashlib.md5(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        gemm(A, B, a_kc=akc, b_kc=bkc, out_bf16=out)
    e1.record()
    torch.cuda.synchronize()
    h2 = hashlib.md5(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{M:6d} {N:6d} {K:6d} akc={int(akc)} bkc={int(bkc)}  {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s  md5 {h} {'(stable)' if h == h2 else '(CHANGED between runs: ' + h2 + ')'}", flush=True)
