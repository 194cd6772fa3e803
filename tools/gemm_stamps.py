"""In-kernel timeline of the GEMM (diagnostic build, never the product library).

`python tools/gemm_stamps.py --build` (no GPU needed) compiles csrc/gemm.hip with -DKMB_GEMM_STAMP and links it with the
regular objects into km-bart_amd/lib/libkmbart_hip_stamp.so.  On the GPU box `python tools/gemm_stamps.py M N K [akc bkc]`
runs the v7 kernel from that library and prints, from per-workgroup s_memrealtime stamps (10 ns ticks):
  prologue (entry -> first stage landed), K loop, epilogue phase 1 (accumulators -> LDS), phase 2 (math + stores),
  and per CU slot the gap between one workgroup's last stamp and the next workgroup's entry.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "km-bart_amd")
STAMP_LIB = os.environ.get("KMB_STAMP_LIB", os.path.join(PKG, "lib", "libkmbart_hip_stamp.so"))


def build():
    sys.path.insert(0, PKG)
    import build as b
    b.build()
    objdir = os.path.join(PKG, "lib", "obj")
    o = os.path.join(objdir, "gemm_stamp.o")
    subprocess.check_call(["hipcc", "-x", "hip"] + b.FLAGS + ["-DKMB_GEMM_STAMP", "-c", os.path.join(b.CSRC, "gemm.hip"), "-o", o])
    objs = [o] + [os.path.join(objdir, s + ".o") for s in b.SOURCES if s != "gemm.hip"]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", STAMP_LIB] + objs + b.RCCL_LINK)
    print("built", STAMP_LIB)


def main():
    if "--build" in sys.argv:
        build()
        return
    os.environ["KMB_GEMM_VARIANT"] = os.environ.get("KMB_GEMM_VARIANT", "7")
    sys.path.insert(0, PKG)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import numpy as np
    import torch
    from kmbart import _lib
    _lib.LIB_PATH = STAMP_LIB
    from gpu_util import DEV, bf, gemm
    lib = _lib.load()
    lib.kmb_debug_set_stamps.restype = C.c_int
    lib.kmb_debug_set_stamps.argtypes = [C.c_void_p]
    args = [int(a) for a in sys.argv[1:] if not a.startswith("-")]
    M, N, K = args[:3]
    akc, bkc = (bool(args[3]), bool(args[4])) if len(args) >= 5 else (True, True)
    torch.manual_seed(0)
    A = bf(torch.randn((M, K) if akc else (K, M), device=DEV))
    B = bf(torch.randn((N, K) if bkc else (K, N), device=DEV) * 0.05)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    # KMB_STAMP_EPI = a1 (bias + GeLU + stored GeLU': the fc1 forward class) | a2 (x stored GeLU' + column sums: the fc2 data-
    # gradient class) | res (bias + residual + dropout): the epilogue class of the stamped launches
    epi = os.environ.get("KMB_STAMP_EPI", "")
    EK = {}
    if epi == "a1":
        EK = dict(bias=torch.randn(N, device=DEV), act=1, preact=torch.empty((M, N), dtype=torch.bfloat16, device=DEV))
    elif epi == "a2":
        EK = dict(act=2, aux=bf(torch.randn(M, N, device=DEV)), colsum=torch.zeros(((M + 63) // 64, N), device=DEV))
    elif epi == "bias":
        EK = dict(bias=torch.randn(N, device=DEV))
    elif epi == "biasres":
        EK = dict(bias=torch.randn(N, device=DEV), residual=bf(torch.randn(M, N, device=DEV)))
    elif epi == "res":
        EK = dict(bias=torch.randn(N, device=DEV), residual=bf(torch.randn(M, N, device=DEV)), drop_p=0.1, drop_seed=7)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    stamps = torch.zeros((tiles, 8), dtype=torch.int64, device=DEV)
    for _ in range(5):
        gemm(A, B, a_kc=akc, b_kc=bkc, out_bf16=out, **EK)
    torch.cuda.synchronize()
    assert lib.kmb_debug_set_stamps(C.c_void_p(stamps.data_ptr())) == 0
    if os.environ.get("KMB_COLD") == "1":   # operands from HBM, as inside a training step (tools/gemm_cold_warm.py)
        junk = torch.empty(600 << 20, dtype=torch.uint8, device=DEV)
        junk.fill_(1)
        torch.cuda.synchronize()
    gemm(A, B, a_kc=akc, b_kc=bkc, out_bf16=out, **EK)
    torch.cuda.synchronize()
    lib.kmb_debug_set_stamps(None)
    s = stamps.cpu().numpy().astype(np.int64)
    s = s[s[:, 0] != 0]  # the 256x256 variants launch a quarter of the workgroups
    if os.environ["KMB_GEMM_VARIANT"] in ("11", "12", "13"):   # persistent kernels: totals per workgroup over its tiles
        tick = 0.01
        bn = {"11": 256, "12": 128, "13": 192}[os.environ["KMB_GEMM_VARIANT"]]
        ntile = ((M + 255) // 256) * ((N + bn - 1) // bn)
        print(f"v{os.environ['KMB_GEMM_VARIANT']} M={M} N={N} K={K} workgroups={len(s)} tiles={ntile} ({ntile / len(s):.2f} per workgroup) "
              f"kernel span {(s[:, 4].max() - s[:, 0].min()) * tick:.1f} us")
        for name, v in (("prologue", s[:, 1] - s[:, 0]), ("k_loops (sum)", s[:, 2]), ("  of which wait+barrier", s[:, 5]),
                        ("epilogues (sum)", s[:, 3]), ("store drain (sum)", s[:, 6]), ("whole workgroup", s[:, 4] - s[:, 0])):
            v = v * tick
            print(f"  {name:24s} median {np.median(v):7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f} us")
        return
    tiles = len(s)
    t0 = s[:, 0].min()
    tick = 0.01  # us
    ph = {
        "prologue": (s[:, 1] - s[:, 0]) * tick, "k_loop": (s[:, 2] - s[:, 1]) * tick,
        "epi_phase1": (s[:, 3] - s[:, 2]) * tick, "epi_phase2": (s[:, 4] - s[:, 3]) * tick,
        "whole_wg": (s[:, 4] - s[:, 0]) * tick,
    }
    print(f"M={M} N={N} K={K} akc={int(akc)} bkc={int(bkc)} tiles={tiles} kernel span {(s[:, 4].max() - t0) * tick:.1f} us")
    for k, v in ph.items():
        print(f"  {k:11s} median {np.median(v):6.2f}  p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f} us")
    if s[:, 5].any():
        w = s[:, 5] * tick
        print(f"  wave 0 parked at the K-loop wait+barrier: median {np.median(w):6.2f}  p10 {np.percentile(w, 10):6.2f}  p90 {np.percentile(w, 90):6.2f} us")
    hw = s[:, 7] & 0xFFFFFFFF
    xcc = (s[:, 7] >> 32) & 0xF
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    sh = (hw >> 12) & 0x1
    key = xcc * 1000 + se * 100 + sh * 50 + cu
    gaps, per_cu = [], []
    for kcu in np.unique(key):
        idx = np.where(key == kcu)[0]
        per_cu.append(len(idx))
        order = idx[np.argsort(s[idx, 0])]
        # two workgroups share a CU: pair every entry with the latest exit before it
        ends = np.sort(s[order, 4])
        for i in order:
            prev = ends[ends <= s[i, 0]]
            if len(prev):
                gaps.append((s[i, 0] - prev.max()) * tick)
    print(f"  CUs seen {len(np.unique(key))}, workgroups per CU min {min(per_cu)} max {max(per_cu)}")
    if gaps:
        gaps = np.array(gaps)
        print(f"  exit->next entry gap on a CU: median {np.median(gaps):.2f}  p90 {np.percentile(gaps, 90):.2f} us")
    first = (s[:, 0] - t0) * tick
    print(f"  first-wave entries (tiles entering within 2 us of the first): {(first < 2.0).sum()}, last entry at {first.max():.1f} us")


if __name__ == "__main__":
    main()
