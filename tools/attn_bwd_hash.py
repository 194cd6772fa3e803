"""md5 of attention forward's (O, log-sum-exp) and backward's outputs (dQ | dK | dV and the bias-gradient column sums) at the three training shapes: run with two
builds (KMB_LIB_PATH) to check that a kernel change returns the same bits.  B=256 python tools/attn_bwd_hash.py"""
import ctypes as C
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, attn_struct, stream  # noqa: E402
from kmbart import _lib  # noqa: E402
from kmbart._lib import check, ptr  # noqa: E402

lib = _lib.load()
B, H, d = int(os.environ.get("B", "256")), 12, 768
for name, Tq, Tk, causal in (("enc self", 64, 64, 0), ("dec self", 32, 32, 1), ("cross", 32, 64, 0), ("ragged", 23, 51, 0), ("self 32 non-causal", 32, 32, 0),
                            ("self 23 causal", 23, 23, 1), ("self 9", 9, 9, 0)):
    g = torch.Generator(device=DEV).manual_seed(1)
    qkv = (torch.randn(B * Tq, 3 * d, device=DEV, generator=g) * 0.5).bfloat16()
    kv = (torch.randn(B * Tk, 3 * d, device=DEV, generator=g) * 0.5).bfloat16() if Tk != Tq else qkv
    O = torch.empty(B * Tq, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B * H * Tq, dtype=torch.float32, device=DEV)
    mask = (torch.rand(B, Tk, device=DEV, generator=g) > 0.2).long()
    mask[:, 0] = 1
    a = attn_struct(qkv[:, :d], kv[:, d:2 * d], kv[:, 2 * d:], B, H, Tq, Tk, mask, causal, O, lse)
    check(lib.kmb_op_attn_fwd(C.byref(a), stream()))
    dO = (torch.randn(B * Tq, d, device=DEV, generator=g) * 0.1).bfloat16()
    dqkv = torch.zeros(B * Tq, 3 * d, dtype=torch.bfloat16, device=DEV)
    dkv = torch.zeros(B * Tk, 3 * d, dtype=torch.bfloat16, device=DEV) if Tk != Tq else dqkv
    cs = torch.zeros(B, 3 * d, dtype=torch.float32, device=DEV)
    a.dO, a.lddo = ptr(dO), d
    a.dQ, a.dK, a.dV = ptr(dqkv[:, :d]), ptr(dkv[:, d:2 * d]), ptr(dkv[:, 2 * d:])
    a.lddq, a.lddk, a.lddv = 3 * d, 3 * d, 3 * d
    a.dq_scale = 0.125
    a.dq_colsum, a.dk_colsum, a.dv_colsum, a.ld_colsum = ptr(cs[:, :d]), ptr(cs[:, d:2 * d]), ptr(cs[:, 2 * d:]), 3 * d
    check(lib.kmb_op_attn_bwd(C.byref(a), stream()))
    torch.cuda.synchronize()
    hh = hashlib.md5()
    for t in (O, lse, dqkv, dkv, cs):   # the forward's outputs too
        hh.update(t.cpu().view(torch.uint8).numpy().tobytes())
    print(name, Tq, Tk, hh.hexdigest())
