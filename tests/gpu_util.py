"""Thin helpers that call single HIP kernels through the C-ABI with torch tensors (GPU tests only)."""
import ctypes as C
import os

import torch

from kmbart import _lib
from kmbart._lib import KmbAttn, KmbAttnDecode, KmbDrop, KmbGemm, check, ptr

DEV = "cuda:0"


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bf(x):
    return x.to(torch.bfloat16).contiguous()


def gemm(A, B, a_kc=True, b_kc=True, M=None, N=None, K=None, bias=None, col_scale=1.0, col_scale_n=0, act=0,
         preact=None, aux=None, drop_p=0.0, drop_seed=0, residual=None, out_bf16=None, out_f32=None, beta=0.0,
         split_k=0, slab=None, colsum=None, tile_order=None, row_shift=None, row_sums=None, pick_col=None, pick_out=None,
         allrows=False):
    """A, B are bf16 2-D tensors in their STORAGE layout; M/N/K default from the shapes."""
    lib = _lib.load()
    if M is None:
        M = A.shape[0] if a_kc else A.shape[1]
    if K is None:
        K = A.shape[1] if a_kc else A.shape[0]
    if N is None:
        N = B.shape[0] if b_kc else B.shape[1]
    g = KmbGemm()
    g.A, g.B = ptr(A), ptr(B)
    g.lda, g.ldb = A.stride(0), B.stride(0)
    g.a_kc, g.b_kc = int(a_kc), int(b_kc)
    g.M, g.N, g.K = M, N, K
    g.bias = ptr(bias)
    g.col_scale, g.col_scale_n = col_scale, col_scale_n
    g.act = act
    if preact is not None:
        g.preact, g.ld_preact = ptr(preact), preact.stride(0)
    if aux is not None:
        g.aux, g.ld_aux = ptr(aux), aux.stride(0)
    thr = int(round(drop_p * 65536))
    g.drop_thr16, g.drop_seed = thr, drop_seed
    g.drop_scale = 1.0 / (1.0 - thr / 65536.0) if thr else 1.0
    if residual is not None:
        g.residual, g.ld_res = ptr(residual), residual.stride(0)
    if out_bf16 is not None:
        g.out_bf16, g.ld_out_bf16 = ptr(out_bf16), out_bf16.stride(0)
    if out_f32 is not None:
        g.out_f32, g.ld_out_f32 = ptr(out_f32), out_f32.stride(0)
    g.beta = beta
    g.split_k = split_k
    g.slab = ptr(slab)
    g.colsum = ptr(colsum)
    if row_sums is not None:
        g.row_shift, g.row_sums, g.row_sums_ld = ptr(row_shift), ptr(row_sums), row_sums.stride(0)
        g.pick_col, g.pick_out = ptr(pick_col), ptr(pick_out)
    g.tile_order = int(os.environ.get("KMB_TILE_ORDER", "0"), 0) if tile_order is None else tile_order
    check((lib.kmb_op_gemm_allrows if allrows else lib.kmb_op_gemm)(C.byref(g), stream()))


def attn_struct(Q, K, V, B, H, Tq, Tk, key_mask, causal, O, lse):
    a = KmbAttn()
    a.Q, a.K, a.V = ptr(Q), ptr(K), ptr(V)
    a.ldq, a.ldk, a.ldv = Q.stride(-2), K.stride(-2), V.stride(-2)
    a.B, a.H, a.Tq, a.Tk = B, H, Tq, Tk
    a.key_mask = ptr(key_mask)
    a.causal = int(causal)
    a.O, a.ldo = ptr(O), O.stride(-2)
    a.lse = ptr(lse)
    return a


def rel_err(a, b):
    a = a.float()
    b = b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def dropout_mask(seed, p, rows, cols):
    lib = _lib.load()
    keep = torch.empty((rows, cols), dtype=torch.uint8, device=DEV)
    check(lib.kmb_op_dropout_mask(seed, p, rows, cols, ptr(keep), stream()))
    return keep.bool()
