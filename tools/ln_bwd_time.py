"""Time the LayerNorm forward and backward (csrc/norm.hip ln_fwd_stream_kernel, ln_bwd_kernel) at the training shapes: rows = batch x 64 / 32 tokens, d = 768, with and without the second
(dropout-masked) output; prints microseconds and TB/s of the bytes it has to move (dy, z in; dz [, out2] out).  B=1024 python tools/ln_bwd_time.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, stream  # noqa: E402
from kmbart import _lib  # noqa: E402
from kmbart._lib import KmbDrop, check, ptr  # noqa: E402

lib = _lib.load()
B, D = int(os.environ.get("B", "1024")), 768
thr = round(0.1 * 65536)
sc = 1.0 / (1.0 - thr / 65536.0)
for rows in (B * 64, B * 32):
    g = torch.Generator(device=DEV).manual_seed(1)
    z = torch.randn(rows, D, device=DEV, generator=g).bfloat16()
    dy = torch.randn(rows, D, device=DEV, generator=g).bfloat16()
    gamma = torch.ones(D, device=DEV)
    mean, rstd = torch.zeros(rows, device=DEV), torch.ones(rows, device=DEV)
    dz, out2 = torch.empty_like(z), torch.empty_like(z)
    dg, db = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    scratch = torch.empty(int(lib.kmb_op_ln_bwd_scratch(rows, D)), device=DEV)
    d1, d2 = KmbDrop(thr, 77, sc), KmbDrop(thr, 99, sc)
    beta = torch.zeros(D, device=DEV)
    zs = [z] + [z.clone() for _ in range(3)]   # four row sets in turn: 800 MB at 65536 rows, so the rows come from HBM and not from the 256 MB Infinity Cache
    ys = [torch.empty_like(z) for _ in range(4)]
    for k in range(4):
        check(lib.kmb_op_ln_fwd(ptr(zs[k]), ptr(gamma), ptr(beta), ptr(ys[k]), ptr(mean), ptr(rstd), rows, D, 1e-5, stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(20):
        check(lib.kmb_op_ln_fwd(ptr(zs[k & 3]), ptr(gamma), ptr(beta), ptr(ys[k & 3]), ptr(mean), ptr(rstd), rows, D, 1e-5, stream()))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50.0
    print(f"rows {rows:6d}  {'forward (four row sets in turn)':32s} {us:7.1f} us  {rows * D * 4 / us * 1e-6:.2f} TB/s")
    del zs, ys
    for name, o2, dr1, dr2 in (("dz only, no dropout", None, None, None), ("dz + out2, both dropout masks", out2, d1, d2), ("dz + out2, out2 mask only", out2, None, d2)):
        def run():
            check(lib.kmb_op_ln_bwd(ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(dz), ptr(o2), C.byref(dr1) if dr1 else None,
                                    C.byref(dr2) if dr2 else None, ptr(dg), ptr(db), ptr(scratch), rows, D, stream()))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50.0
        byts = rows * D * 2 * (3 + (1 if o2 is not None else 0))
        print(f"rows {rows:6d}  {name:32s} {us:7.1f} us  {byts / us * 1e-6:.2f} TB/s (includes the parameter-gradient reducer launch)")
