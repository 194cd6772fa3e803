"""In-kernel timeline of the all-rows vocabulary projection (diagnostic build: `python tools/gemm_stamps.py --build` first).
`python tools/allrows_stamps.py [R]`: per workgroup entry -> first stage landed -> K step 6 -> K loop done -> end (10 ns ticks)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "km-bart_amd")
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kmbart import _lib  # noqa: E402
_lib.LIB_PATH = os.environ.get("KMB_STAMP_LIB", os.path.join(PKG, "lib", "libkmbart_hip_stamp.so"))
from gpu_util import DEV, bf, stream  # noqa: E402
from kmbart._lib import KmbGemm, check, ptr  # noqa: E402

lib = _lib.load()
lib.kmb_debug_set_stamps.restype = C.c_int
lib.kmb_debug_set_stamps.argtypes = [C.c_void_p]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 320
V, ld = 50320, 50432
g = torch.Generator(device=DEV).manual_seed(1)
A = bf(torch.randn(R, 768, device=DEV, generator=g) * 0.5)
W = bf(torch.randn(ld, 768, device=DEV, generator=g) * 0.05)
bias = torch.randn(V, device=DEV, generator=g)
out = torch.empty((R, ld), dtype=torch.float32, device=DEV)
stats = torch.empty(int(lib.kmb_op_gemm_allrows_stats_floats(V)), device=DEV)
p = KmbGemm()
p.A, p.B, p.lda, p.ldb, p.a_kc, p.b_kc = ptr(A), ptr(W), 768, 768, 1, 1
p.M, p.N, p.K, p.bias, p.col_scale, p.drop_scale = R, V, 768, ptr(bias), 1.0, 1.0
p.out_f32, p.ld_out_f32 = ptr(out), ld
nblk = (V + 255) // 256
SL = 8
trash = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
for with_stats in (False, True):
    for cold in (False, True):
        st = torch.zeros((nblk, SL), dtype=torch.int64, device=DEV)
        for _ in range(3):
            check(lib.kmb_op_gemm_allrows_stats(C.byref(p), ptr(stats), stream()) if with_stats else lib.kmb_op_gemm_allrows(C.byref(p), stream()))
        if cold:
            trash.fill_(1)   # evict the tied matrix from L2 / MALL, as the decoder layers before the projection do
        torch.cuda.synchronize()
        assert lib.kmb_debug_set_stamps(st.data_ptr()) == 0
        check(lib.kmb_op_gemm_allrows_stats(C.byref(p), ptr(stats), stream()) if with_stats else lib.kmb_op_gemm_allrows(C.byref(p), stream()))
        torch.cuda.synchronize()
        assert lib.kmb_debug_set_stamps(None) == 0
        s = st.cpu().numpy().astype(np.int64)
        t0 = s[:, 0].min()
        f = lambda x: "median %6.2f  p90 %6.2f us" % (np.median(x) / 100.0, np.percentile(x, 90) / 100.0)   # noqa: E731
        print("stats=%d cold=%d: %d workgroups, span %.2f us, entries spread over %.2f us" %
              (with_stats, cold, nblk, (s[:, 4].max() - t0) / 100.0, (s[:, 0].max() - t0) / 100.0))
        print("   entry -> first stage landed   ", f(s[:, 1] - s[:, 0]))
        print("   K steps 0..5                  ", f(s[:, 3] - s[:, 1]))
        print("   K steps 6..11                 ", f(s[:, 2] - s[:, 3]))
        print("   epilogue                      ", f(s[:, 4] - s[:, 2]))
        print("   whole workgroup               ", f(s[:, 4] - s[:, 0]))
        print("   shader clock over the K loop   median %.0f MHz" % float(np.median(s[:, 5] / np.maximum(s[:, 6], 1)) * 100.0))
