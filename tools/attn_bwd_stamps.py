"""In-kernel timeline of attn_bwd_small_kernel (diagnostic build -DKMB_ATTN_STAMP, never the product library): thread 0 of every
workgroup stamps s_memrealtime (10 ns) at the phase boundaries of every item it works through; printed: median microseconds per phase
over the steady-state items (the first and last item of a workgroup left out) for the encoder / decoder / cross shapes at batch B.

    python tools/attn_bwd_stamps.py --build      (no GPU needed: km-bart_amd/lib/libkmbart_hip_astamp.so)
    B=1024 python tools/attn_bwd_stamps.py
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
if "--build" in sys.argv:
    import build as b
    print(b.build_variant("astamp", ["KMB_ATTN_STAMP"], sources=("attention.hip",)))
    sys.exit(0)
os.environ["KMB_LIB_PATH"] = os.path.join(ROOT, "km-bart_amd", "lib", "libkmbart_hip_astamp.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gpu_util import DEV, attn_struct, stream  # noqa: E402
from kmbart import _lib  # noqa: E402
from kmbart._lib import check, ptr  # noqa: E402

lib = _lib.load()
lib.kmb_debug_set_attn_stamps.restype = C.c_int
lib.kmb_debug_set_attn_stamps.argtypes = [C.c_void_p]
NAMES = ["wait: previous item's LDS free (barrier; + previous item's column-sum write)", "registers -> LDS", "barrier", "next item's loads issued",
         "S, dP MFMAs + softmax gradient (+ delta) + P / dS to LDS", "barrier", "dQ, dK, dV MFMAs", "column sums (DPP)",
         "dQ / dK / dV: staging + stores"]
B, H, d = int(os.environ.get("B", "1024")), 12, 768
for name, Tq, Tk, causal in (("enc self", 64, 64, 0), ("dec self", 32, 32, 1), ("cross", 32, 64, 0)):
    g = torch.Generator(device=DEV).manual_seed(1)
    qkv = (torch.randn(B * Tq, 3 * d, device=DEV, generator=g) * 0.5).bfloat16()
    kv = (torch.randn(B * Tk, 3 * d, device=DEV, generator=g) * 0.5).bfloat16() if Tk != Tq else qkv
    O = torch.empty(B * Tq, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B * H * Tq, dtype=torch.float32, device=DEV)
    mask = torch.ones(B, Tk, dtype=torch.int64, device=DEV)
    a = attn_struct(qkv[:, :d], kv[:, d:2 * d], kv[:, 2 * d:], B, H, Tq, Tk, mask, causal, O, lse)
    check(lib.kmb_op_attn_fwd(C.byref(a), stream()))
    dO = (torch.randn(B * Tq, d, device=DEV, generator=g) * 0.1).bfloat16()
    dqkv = torch.empty(B * Tq, 3 * d, dtype=torch.bfloat16, device=DEV)
    dkv = torch.empty(B * Tk, 3 * d, dtype=torch.bfloat16, device=DEV) if Tk != Tq else dqkv
    cs = torch.empty(B, 3 * d, dtype=torch.float32, device=DEV)
    a.dO, a.lddo = ptr(dO), d
    a.dQ, a.dK, a.dV = ptr(dqkv[:, :d]), ptr(dkv[:, d:2 * d]), ptr(dkv[:, 2 * d:])
    a.lddq, a.lddk, a.lddv = 3 * d, 3 * d, 3 * d
    a.dq_scale = 0.125
    a.dq_colsum, a.dk_colsum, a.dv_colsum, a.ld_colsum = ptr(cs[:, :d]), ptr(cs[:, d:2 * d]), ptr(cs[:, 2 * d:]), 3 * d
    for _ in range(3):
        check(lib.kmb_op_attn_bwd(C.byref(a), stream()))
    stamps = torch.zeros((4096, 32, 16), dtype=torch.int64, device=DEV)
    assert lib.kmb_debug_set_attn_stamps(C.c_void_p(stamps.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.kmb_op_attn_bwd(C.byref(a), stream()))
    e1.record()
    torch.cuda.synchronize()
    lib.kmb_debug_set_attn_stamps(None)
    s = stamps.cpu().numpy()
    used = s[:, :, 0] > 0
    nwg = int(used[:, 0].sum())
    per_wg = used.sum(axis=1)
    rows = []
    for w in range(4096):
        n = int(per_wg[w])
        for it in range(1, n - 1):   # steady state
            rows.append(s[w, it, :10])
    rows = np.array(rows, dtype=np.float64)
    dt = (rows[:, 1:] - rows[:, :-1]) * 0.01
    item = np.array([s[w, it + 1, 0] - s[w, it, 0] for w in range(4096) for it in range(1, int(per_wg[w]) - 1)], dtype=np.float64) * 0.01
    print(f"{name}: B={B} Tq={Tq} Tk={Tk}: launch {e0.elapsed_time(e1) * 1e3:.1f} us, {nwg} workgroups, {int(per_wg.max())} items each; "
          f"item to item median {np.median(item):.2f} us")
    for i, nm in enumerate(NAMES):
        print(f"   {nm:52s} median {np.median(dt[:, i]):6.2f}  p90 {np.percentile(dt[:, i], 90):6.2f} us")
