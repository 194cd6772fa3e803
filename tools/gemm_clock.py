"""The clock the chip holds inside the persistent GEMM kernels (MI355X_MICROARCH.md 'DVFS give-back' item 6): shader-clock ticks
(s_memtime) over real-time ticks (s_memrealtime, 100 MHz) stamped around each persistent workgroup's whole life, after >= 2 s of
back-to-back launches of the same shape on random data.  Diagnostic build (-DKMB_GEMM_STAMP -DKMB_STAMP_SLOTS=12 -DKMB_V11_PREFETCH=0), never the
product library.  (The L2 touch of the activation panel is compiled out: when this tool was written the touch's asm load kept one register
"reserved" only by a "+v" constraint chain, and with the stamp build's extra live values the allocator split that web -- memory fault.
Since then the touch writes v255 and the kernels are compiled not to allocate it (KMB_L2_TOUCH, csrc/gemm.hip); the flag stays so that
the committed clock figures remain reproducible.)
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "km-bart_amd")
LIB = os.path.join(PKG, "lib", "libkmbart_hip_clock.so")


def build():
    sys.path.insert(0, PKG)
    import build as b
    b.build()
    objdir = os.path.join(PKG, "lib", "obj")
    o = os.path.join(objdir, "gemm_clock.o")
    subprocess.check_call(["hipcc", "-x", "hip"] + b.FLAGS + ["-DKMB_GEMM_STAMP", "-DKMB_STAMP_SLOTS=12", "-DKMB_V11_PREFETCH=0", "-c",
                                                             os.path.join(b.CSRC, "gemm.hip"), "-o", o])
    objs = [o] + [os.path.join(objdir, s + ".o") for s in b.SOURCES if s != "gemm.hip"]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + b.RCCL_LINK)
    print("built", LIB)


def main():
    if "--build" in sys.argv:
        build()
        return
    os.environ.setdefault("KMB_GEMM_VARIANT", "11")
    sys.path.insert(0, PKG)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import numpy as np
    import torch
    from kmbart import _lib
    _lib.LIB_PATH = LIB
    os.environ["KMB_LIB_PATH"] = LIB
    from gpu_util import DEV, bf, gemm
    lib = _lib.load()
    lib.kmb_debug_set_stamps.restype = C.c_int
    lib.kmb_debug_set_stamps.argtypes = [C.c_void_p]
    zeros = "zeros" in sys.argv
    print("# in-kernel clock of the persistent GEMM (variant %s), %s operands, 2 s of back-to-back launches before the stamped one"
          % (os.environ["KMB_GEMM_VARIANT"], "ALL-ZERO" if zeros else "random"))
    for name, M, N, K, kw in (("fc1 forward + GeLU", 65536, 3072, 768, dict(act=1)), ("qkv forward", 65536, 2304, 768, {}),
                              ("fc2 forward K=3072", 65536, 768, 3072, {})):
        A = bf(torch.zeros(M, K, device=DEV) if zeros else torch.randn(M, K, device=DEV) * 0.5)
        B = bf(torch.zeros(N, K, device=DEV) if zeros else torch.randn(N, K, device=DEV) * 0.5)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ekw = dict(bias=torch.randn(N, device=DEV), tile_order=3, **kw)
        if kw.get("act") == 1:
            ekw["preact"] = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        stamps = torch.zeros((256, 12), dtype=torch.int64, device=DEV)
        t0 = time.time()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        while time.time() - t0 < 2.0:
            for _ in range(50):
                gemm(A, B, out_bf16=out, **ekw)
            torch.cuda.synchronize()
            n += 50
        e0.record()
        for _ in range(20):
            gemm(A, B, out_bf16=out, **ekw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        assert lib.kmb_debug_set_stamps(C.c_void_p(stamps.data_ptr())) == 0
        gemm(A, B, out_bf16=out, **ekw)
        torch.cuda.synchronize()
        lib.kmb_debug_set_stamps(None)
        s = stamps.cpu().numpy().astype(np.int64)
        ok = s[:, 9] > 0
        clk = s[ok, 8] / s[ok, 9] * 0.1   # GHz
        print("%-22s %6dx%5dx%5d | %7.1f us %5.0f TFLOP/s | in-kernel clock median %.2f GHz (p10 %.2f, p90 %.2f) over %d workgroups | "
              "peak at that clock %.0f TFLOP/s" % (name, M, N, K, us, 2.0 * M * N * K / us * 1e-6, float(np.median(clk)),
                                                  float(np.percentile(clk, 10)), float(np.percentile(clk, 90)), int(ok.sum()),
                                                  256 * 4 * 1024 * float(np.median(clk)) * 1e9 * 1e-12))


if __name__ == "__main__":
    main()
