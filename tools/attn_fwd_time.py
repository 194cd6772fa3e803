"""Time attention forward at the training shapes (encoder self T = 64, decoder self T = 32 causal, cross Tq = 32 / Tk = 64),
interleaved layout [row, 3d] as the engine uses it; B from the environment (default 1024, the benchmark batch)."""
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
from kmbart._lib import check  # noqa: E402

lib = _lib.load()
B, H, d = int(os.environ.get("B", "1024")), 12, 768
for name, Tq, Tk, causal in (("enc self", 64, 64, 0), ("dec self", 32, 32, 1), ("cross", 32, 64, 0)):
    g = torch.Generator(device=DEV).manual_seed(1)
    qkv = (torch.randn(B * Tq, 3 * d, device=DEV, generator=g) * 0.5).bfloat16()
    kv = (torch.randn(B * Tk, 3 * d, device=DEV, generator=g) * 0.5).bfloat16() if Tk != Tq else qkv
    O = torch.empty(B * Tq, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B * H * Tq, dtype=torch.float32, device=DEV)
    mask = torch.ones(B, Tk, dtype=torch.int64, device=DEV)
    a = attn_struct(qkv[:, :d], kv[:, d:2 * d], kv[:, 2 * d:], B, H, Tq, Tk, mask, causal, O, lse)
    for _ in range(3):
        check(lib.kmb_op_attn_fwd(C.byref(a), stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        check(lib.kmb_op_attn_fwd(C.byref(a), stream()))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50.0
    byts = 2 * d * (B * Tq * 2 + B * Tk * 2)   # Q in + O out; K, V in
    print(f"{name:9s} B={B} Tq={Tq} Tk={Tk}: {us:7.1f} us  {byts / us * 1e-6:.2f} TB/s algorithmic")
