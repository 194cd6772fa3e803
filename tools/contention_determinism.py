"""Does a kernel give the same bits when another kernel shares the GPU?  Each op runs alone (reference) and then 30
times while a long GEMM runs on a second stream; any bit difference is reported."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, attn_struct, bf, check, gemm, ptr, stream  # noqa: E402
from kmbart import _lib  # noqa: E402

lib = _lib.load()
torch.manual_seed(0)
side = torch.cuda.Stream()
# the hog: a weight-gradient-like GEMM (M-contiguous operands, split-K), ~200 us
hA = bf(torch.randn(32768, 3072, device=DEV)); hB = bf(torch.randn(32768, 768, device=DEV))
hslab = torch.empty(3 * 3072 * 768, dtype=torch.float32, device=DEV)


def hog():
    with torch.cuda.stream(side):
        for _ in range(3):
            gemm(hA, hB, a_kc=False, b_kc=False, M=3072, N=768, K=32768, split_k=3, slab=hslab)


def run(name, fn, outs):
    fn()
    torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    bad = 0
    for _ in range(30):
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        hog()
        fn()
        torch.cuda.synchronize()
        bad += any(not torch.equal(a, b) for a, b in zip(ref, outs))
    print(f"{name}: {bad} of 30 runs beside another kernel differ from the solo result", flush=True)


# attention backward, decoder self-attention shape (causal, T = 32)
B, H, T = 64, 12, 32
d = H * 64
qkv = bf(torch.randn(B * T, 3 * d, device=DEV) * 0.7)
O = torch.zeros(B * T, d, dtype=torch.bfloat16, device=DEV); lse = torch.zeros(B, H, T, device=DEV)
a = attn_struct(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, T, T, None, 1, O, lse)
check(lib.kmb_op_attn_fwd(C.byref(a), stream()))
dO = bf(torch.randn(B * T, d, device=DEV)); dqkv = torch.zeros(B * T, 3 * d, dtype=torch.bfloat16, device=DEV)
a.dO, a.lddo = ptr(dO), d
a.dQ, a.lddq = ptr(dqkv), 3 * d
a.dK, a.dV, a.lddk, a.lddv = ptr(dqkv[:, d:]), ptr(dqkv[:, 2 * d:]), 3 * d, 3 * d
a.dq_scale = 0.125
run("attn_bwd causal T=32", lambda: check(lib.kmb_op_attn_bwd(C.byref(a), stream())), [dqkv])
# data-gradient GEMM 2048 x 768 x 768 with residual
X = bf(torch.randn(2048, 768, device=DEV)); W = bf(torch.randn(768, 768, device=DEV) * 0.05); R = bf(torch.randn(2048, 768, device=DEV))
out = torch.zeros(2048, 768, dtype=torch.bfloat16, device=DEV)
run("dgrad GEMM 2048x768x768 + residual", lambda: gemm(X, W, a_kc=True, b_kc=False, residual=R, out_bf16=out), [out])
X2 = bf(torch.randn(2048, 2304, device=DEV)); W2 = bf(torch.randn(2304, 768, device=DEV) * 0.05)
run("dgrad GEMM 2048x768x2304 + residual", lambda: gemm(X2, W2, a_kc=True, b_kc=False, residual=R, out_bf16=out), [out])
# forward-layout GEMM with GeLU
W3 = bf(torch.randn(3072, 768, device=DEV) * 0.05); o3 = torch.zeros(2048, 3072, dtype=torch.bfloat16, device=DEV)
p3 = torch.zeros(2048, 3072, dtype=torch.bfloat16, device=DEV); bias = torch.randn(3072, device=DEV)
run("fwd GEMM 2048x3072x768 GeLU", lambda: gemm(X, W3, bias=bias, act=1, preact=p3, out_bf16=o3), [o3, p3])
# LayerNorm backward, decoder shape
M, D = 2048, 768
z = bf(torch.randn(M, D, device=DEV)); gamma = torch.randn(D, device=DEV)
mean, rstd = z.float().mean(-1), (z.float().var(-1, unbiased=False) + 1e-5).rsqrt()
dy = bf(torch.randn(M, D, device=DEV)); dz = torch.zeros_like(z)
dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
scratch = torch.empty(int(lib.kmb_op_ln_bwd_scratch(M, D)), device=DEV)
run("ln_bwd 2048x768", lambda: check(lib.kmb_op_ln_bwd(ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(dz), None, None, None,
                                                        ptr(dg), ptr(db), ptr(scratch), M, D, stream())), [dz, dg, db])
M = 4096
z = bf(torch.randn(M, D, device=DEV)); mean, rstd = z.float().mean(-1), (z.float().var(-1, unbiased=False) + 1e-5).rsqrt()
dy = bf(torch.randn(M, D, device=DEV)); dz = torch.zeros_like(z)
scratch = torch.empty(int(lib.kmb_op_ln_bwd_scratch(M, D)), device=DEV)
run("ln_bwd 4096x768", lambda: check(lib.kmb_op_ln_bwd(ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(dz), None, None, None,
                                                        ptr(dg), ptr(db), ptr(scratch), M, D, stream())), [dz, dg, db])
# large forward / data-gradient shapes (the persistent variants when KMB_GEMM_VARIANT=11 / 12 is set)
Mb = 16384
Xb = bf(torch.randn(Mb, 768, device=DEV)); ob = torch.zeros(Mb, 3072, dtype=torch.bfloat16, device=DEV); pb = torch.zeros_like(ob)
run("fwd GEMM 16384x3072x768 bias + GeLU + pre-activation", lambda: gemm(Xb, W3, bias=bias, act=1, preact=pb, out_bf16=ob), [ob, pb])
Hb = bf(torch.randn(Mb, 3072, device=DEV)); W4 = bf(torch.randn(768, 3072, device=DEV) * 0.05); Rb = bf(torch.randn(Mb, 768, device=DEV))
o4 = torch.zeros(Mb, 768, dtype=torch.bfloat16, device=DEV); b4 = torch.randn(768, device=DEV)
run("fwd GEMM 16384x768x3072 bias + dropout + residual", lambda: gemm(Hb, W4, bias=b4, residual=Rb, drop_p=0.1, drop_seed=7, out_bf16=o4), [o4])
W5 = bf(torch.randn(768, 3072, device=DEV) * 0.05); aux = bf(torch.randn(Mb, 3072, device=DEV)); cs = torch.zeros((Mb + 63) // 64, 3072, device=DEV)
o5 = torch.zeros(Mb, 3072, dtype=torch.bfloat16, device=DEV)
run("dgrad GEMM 16384x3072x768 GeLU' + column sums", lambda: gemm(Xb, W5, a_kc=True, b_kc=False, act=2, aux=aux, colsum=cs, out_bf16=o5), [o5, cs])
