"""Time attention backward at the training shapes (b = 512: encoder self T = 64, decoder self T = 32 causal, cross
Tq = 32 / Tk = 64), interleaved layout [row, 3d] as the engine uses it.  KMB_ATTN_BWD_SMALL=0 selects the general kernel."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: E402
_diag.use_diag_lib()   # the A/B knobs live in the diagnostic build only (csrc/diag.h)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, attn_struct, stream  # noqa: E402
from kmbart import _lib  # noqa: E402
from kmbart._lib import check, ptr  # noqa: E402

lib = _lib.load()
B, H, d = int(os.environ.get("B", "512")), 12, 768
for name, Tq, Tk, causal in (("enc self", 64, 64, 0), ("dec self", 32, 32, 1), ("cross", 32, 64, 0)):
    g = torch.Generator(device=DEV).manual_seed(1)
    qkv = (torch.randn(B * Tq, 3 * d, device=DEV, generator=g) * 0.5).bfloat16()
    kv = (torch.randn(B * Tk, 3 * d, device=DEV, generator=g) * 0.5).bfloat16() if Tk != Tq else qkv
    O = torch.empty(B * Tq, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B * H * Tq, dtype=torch.float32, device=DEV)
    mask = torch.ones(B, Tk, dtype=torch.int64, device=DEV)   # the training step always passes a padding mask
    a = attn_struct(qkv[:, :d], kv[:, d:2 * d], kv[:, 2 * d:], B, H, Tq, Tk, None if os.environ.get("NOMASK") else mask, causal, O, lse)
    check(lib.kmb_op_attn_fwd(C.byref(a), stream()))
    dO = (torch.randn(B * Tq, d, device=DEV, generator=g) * 0.1).bfloat16()
    dqkv = torch.empty(B * Tq, 3 * d, dtype=torch.bfloat16, device=DEV)
    dkv = torch.empty(B * Tk, 3 * d, dtype=torch.bfloat16, device=DEV) if Tk != Tq else dqkv
    cs = torch.empty(B, 3 * d, dtype=torch.float32, device=DEV)
    a.dO, a.lddo = ptr(dO), d
    a.dQ, a.dK, a.dV = ptr(dqkv[:, :d]), ptr(dkv[:, d:2 * d]), ptr(dkv[:, 2 * d:])
    a.lddq, a.lddk, a.lddv = 3 * d, 3 * d, 3 * d
    a.dq_scale = 0.125
    if not os.environ.get("NOCOLSUM"):   # ablation: without the bias-gradient column sums (48 DPP chains + a barrier per item)
        a.dq_colsum, a.dk_colsum, a.dv_colsum, a.ld_colsum = ptr(cs[:, :d]), ptr(cs[:, d:2 * d]), ptr(cs[:, 2 * d:]), 3 * d
    for _ in range(3):
        check(lib.kmb_op_attn_bwd(C.byref(a), stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        check(lib.kmb_op_attn_bwd(C.byref(a), stream()))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50.0
    byts = 2 * d * (B * Tq * 4 + B * Tk * 4)   # Q, dO, O in + dQ out; K, V in + dK, dV out
    print(f"{name:9s} B={B} Tq={Tq} Tk={Tk}: {us:7.1f} us  {byts / us * 1e-6:.2f} TB/s algorithmic")
