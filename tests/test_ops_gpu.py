"""GPU: every HIP kernel, called through the C-ABI, against a plain PyTorch fp32 computation of the same
op on the same (bf16-rounded) inputs.  Tolerances: fp32 outputs differ from the reference only by
accumulation order (<= 1e-4 relative); bf16 outputs carry one bf16 rounding (2^-9 = 2e-3)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from kmbart import _lib  # noqa: E402
from kmbart._lib import KmbAdamW, KmbAttnDecode, KmbDrop, check, ptr  # noqa: E402
from gpu_util import DEV, attn_struct, bf, dropout_mask, gemm, rel_err, stream  # noqa: E402

BF_TOL = 4e-3
F32_TOL = 1e-4


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


# --------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (64, 768, 768), (1024, 2304, 768), (37, 50320, 128)])
def test_gemm_forward_layout(M, N, K):
    A, B = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2))
    bias = rnd(N, seed=3)
    out = torch.empty((M, N), dtype=torch.float32, device=DEV)
    outb = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    gemm(A, B, bias=bias, out_f32=out, out_bf16=outb)
    ref = A.float() @ B.float().t() + bias
    assert rel_err(out, ref) < F32_TOL
    assert rel_err(outb, ref) < BF_TOL


def test_gemm_identity_asymmetric():
    """A = I against an asymmetric B catches transposed / permuted fragment maps exactly."""
    n = 128
    A = bf(torch.eye(n, device=DEV))
    B = bf((torch.arange(n * n, device=DEV).view(n, n) % 251).float() - 125.0)  # exact in bf16
    out = torch.empty((n, n), dtype=torch.float32, device=DEV)
    gemm(A, B, out_f32=out)
    assert torch.equal(out, B.float().t().contiguous())
    gemm(A, B, b_kc=False, out_f32=out)          # B stored [K, N]
    assert torch.equal(out, B.float())
    gemm(A, B, a_kc=False, b_kc=False, out_f32=out)  # A stored [K, M] (identity either way)
    assert torch.equal(out, B.float())
    gemm(B, A, a_kc=False, b_kc=False, out_f32=out)  # A = B stored [K, M] -> logical A = B^T
    assert torch.equal(out, B.float().t().contiguous())


@pytest.mark.parametrize("M,N,K", [(256, 768, 3072), (100, 264, 200), (64, 2056, 768)])
def test_gemm_dgrad_layout(M, N, K):
    """dX[M,N] = dY[M,K] W[K,N]: B operand is N-contiguous (a torch Linear weight read along its rows)."""
    dY, W = bf(rnd(M, K, seed=4)), bf(rnd(K, N, seed=5, scale=0.1))
    out = torch.empty((M, N), dtype=torch.float32, device=DEV)
    gemm(dY, W, a_kc=True, b_kc=False, out_f32=out)
    assert rel_err(out, dY.float() @ W.float()) < F32_TOL


@pytest.mark.parametrize("T,N,K", [(512, 768, 768), (72, 128, 2056), (1000, 264, 136), (9, 128, 64)])
def test_gemm_wgrad_layout(T, N, K):
    """dW[N,K] = dY[T,N]^T X[T,K]: both operands are read against their contiguous dimension."""
    dY, X = bf(rnd(T, N, seed=6)), bf(rnd(T, K, seed=7))
    out = torch.full((N, K), 7.0, dtype=torch.float32, device=DEV)
    gemm(dY, X, a_kc=False, b_kc=False, out_f32=out)
    ref = dY.float().t() @ X.float()
    assert rel_err(out, ref) < F32_TOL
    gemm(dY, X, a_kc=False, b_kc=False, out_f32=out, beta=1.0)  # accumulate
    assert rel_err(out, 2 * ref) < F32_TOL


@pytest.mark.parametrize("T,N,K,S", [(1024, 256, 128, 4), (4096, 768, 768, 8), (640, 128, 136, 3)])
def test_gemm_split_k_slabs(T, N, K, S):
    """split-K: slice s writes its partial sums to slab[s]; the slabs add up to the full product."""
    dY, X = bf(rnd(T, N, seed=15)), bf(rnd(T, K, seed=16))
    slab = torch.full((S, N, K), 9.0, dtype=torch.float32, device=DEV)
    gemm(dY, X, a_kc=False, b_kc=False, split_k=S, slab=slab)
    assert rel_err(slab.sum(0), dY.float().t() @ X.float()) < F32_TOL


@pytest.mark.parametrize("M,N,K", [(512, 256, 128), (1000, 3072, 768), (300, 264, 64)])
def test_gemm_epilogue_column_sums(M, N, K):
    """bias-gradient partials: one row of column sums per 64 rows of the stored (GeLU'-scaled) output."""
    A, B = bf(rnd(M, K, seed=17)), bf(rnd(K, N, seed=18, scale=0.1))
    u = bf(rnd(M, N, seed=19))
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    parts = torch.full(((M + 63) // 64, N), 5.0, dtype=torch.float32, device=DEV)
    uu = u.float().requires_grad_(True)
    F.gelu(uu).sum().backward()
    dg = bf(uu.grad)                      # act 2 multiplies by the STORED derivative (what the forward epilogue wrote)
    gemm(A, B, a_kc=True, b_kc=False, act=2, aux=dg, out_bf16=out, colsum=parts)
    ref = (A.float() @ B.float()) * dg.float()
    assert rel_err(out, ref) < BF_TOL
    assert rel_err(parts.sum(0), ref.sum(0)) < 1e-3


def test_gemm_wgrad_unpadded_output():
    """image-projection weight gradient: logical N = 2052 columns out of a 2056-wide padded operand."""
    T, N, Kp, Kv = 72, 128, 2056, 2052
    dY, X = bf(rnd(T, N, seed=8)), bf(rnd(T, Kp, seed=9))
    out = torch.zeros((N, Kv), dtype=torch.float32, device=DEV)
    gemm(dY, X, a_kc=False, b_kc=False, N=Kv, out_f32=out)
    assert rel_err(out, (dY.float().t() @ X.float())[:, :Kv]) < F32_TOL


def test_gemm_epilogues():
    M, N, K = 192, 256, 128
    A, B = bf(rnd(M, K, seed=10)), bf(rnd(N, K, seed=11, scale=0.2))
    bias = rnd(N, seed=12)
    base = A.float() @ B.float().t() + bias
    # q pre-scale on the first columns
    out = torch.empty((M, N), dtype=torch.float32, device=DEV)
    gemm(A, B, bias=bias, col_scale=0.125, col_scale_n=64, out_f32=out)
    ref = base.clone()
    ref[:, :64] *= 0.125
    assert rel_err(out, ref) < F32_TOL
    # exact GeLU + stored DERIVATIVE GeLU'(pre-activation) (one erf / exp evaluation serves both; backward multiplies)
    pre = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    gemm(A, B, bias=bias, act=1, preact=pre, out_f32=out)
    assert rel_err(out, F.gelu(base)) < F32_TOL
    bb = base.clone().requires_grad_(True)
    F.gelu(bb).sum().backward()
    assert rel_err(pre, bb.grad) < BF_TOL
    out2 = torch.empty_like(out)
    gemm(A, B, bias=bias, act=1, out_f32=out2)          # without the side output: same GeLU
    assert torch.equal(out, out2)
    # multiply by the stored derivative
    gemm(A, B, act=2, aux=pre, out_f32=out)
    assert rel_err(out, (A.float() @ B.float().t()) * pre.float()) < F32_TOL
    # residual
    res = bf(rnd(M, N, seed=14))
    gemm(A, B, bias=bias, residual=res, out_f32=out)
    assert rel_err(out, base + res.float()) < F32_TOL
    # dropout before the residual; mask must be the generator's mask for that seed
    gemm(A, B, bias=bias, drop_p=0.25, drop_seed=1234, residual=res, out_f32=out)
    keep = dropout_mask(1234, 0.25, M, N)
    scale = 1.0 / (1.0 - round(0.25 * 65536) / 65536.0)
    assert rel_err(out, torch.where(keep, base * scale, torch.zeros_like(base)) + res.float()) < F32_TOL
    assert abs(float(keep.float().mean()) - 0.75) < 0.01


# ------------------------------------------------------------------------------------ attention
def ref_attention(q, k, v, key_mask, causal):
    """q [B,H,Tq,64] (already scaled), fp32; HF3.0.2 SelfAttention arithmetic."""
    w = q @ k.transpose(-1, -2)
    Tq, Tk = w.shape[-2:]
    if causal:
        w = w + torch.triu(torch.full((Tq, Tk), float("-inf"), device=w.device), 1)
    if key_mask is not None:
        w = w.masked_fill(key_mask[:, None, None, :] == 0, float("-inf"))
    p = torch.softmax(w, dim=-1)
    return p @ v


ATTN_CASES = [
    dict(B=2, H=2, Tq=64, Tk=64, causal=False, mask=False, fused=True),
    dict(B=3, H=2, Tq=32, Tk=32, causal=True, mask=True, fused=True),
    dict(B=2, H=3, Tq=32, Tk=64, causal=False, mask=True, fused=False),
    dict(B=2, H=1, Tq=40, Tk=100, causal=False, mask=True, fused=False),
    dict(B=1, H=2, Tq=130, Tk=130, causal=True, mask=False, fused=True),
    # two heads per tile (Tq, Tk <= 32, even H; round 5): ragged length, per-sample key masks; and odd H, which keeps one head per tile
    dict(B=4, H=4, Tq=23, Tk=23, causal=True, mask=True, fused=True),
    dict(B=5, H=6, Tq=32, Tk=32, causal=False, mask=True, fused=True),
    dict(B=2, H=3, Tq=32, Tk=32, causal=True, mask=True, fused=True),
]


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_fwd_bwd(case):
    B, H, Tq, Tk = case["B"], case["H"], case["Tq"], case["Tk"]
    d = H * 64
    lib = _lib.load()
    if case["fused"]:  # self-attention: q|k|v interleaved rows of width 3d
        qkv = bf(rnd(B * Tq, 3 * d, seed=20, scale=0.7))
        Q, K, V = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    else:
        Q = bf(rnd(B * Tq, d, seed=21, scale=0.7))
        kv = bf(rnd(B * Tk, 2 * d, seed=22, scale=0.7))
        K, V = kv[:, :d], kv[:, d:]
    key_mask = None
    if case["mask"]:
        key_mask = torch.ones((B, Tk), dtype=torch.int64, device=DEV)
        for b in range(B):
            key_mask[b, Tk - 1 - 3 * b:] = 0  # right padding, different per sample; key 0 always kept
    O = torch.zeros((B * Tq, d), dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros((B, H, Tq), dtype=torch.float32, device=DEV)
    a = attn_struct(Q, K, V, B, H, Tq, Tk, key_mask, case["causal"], O, lse)
    check(lib.kmb_op_attn_fwd(C.byref(a), stream()))

    def heads(x, T):
        return x.float().reshape(B, T, H, 64).transpose(1, 2)

    qf = heads(Q, Tq).requires_grad_(True)
    kf = heads(K, Tk).requires_grad_(True)
    vf = heads(V, Tk).requires_grad_(True)
    ref = ref_attention(qf, kf, vf, key_mask, case["causal"])
    ref_rows = ref.transpose(1, 2).reshape(B * Tq, d)
    assert rel_err(O, ref_rows) < 1e-2  # P is rounded to bf16 before P.V
    w = qf @ kf.transpose(-1, -2)
    if case["causal"]:
        w = w + torch.triu(torch.full((Tq, Tk), float("-inf"), device=DEV), 1)
    if key_mask is not None:
        w = w.masked_fill(key_mask[:, None, None, :] == 0, float("-inf"))
    assert torch.allclose(lse, torch.logsumexp(w, dim=-1).detach(), atol=2e-3)

    # backward
    dO = bf(rnd(B * Tq, d, seed=23))
    ref.backward(heads(dO, Tq))
    dQ = torch.zeros((B * Tq, d), dtype=torch.bfloat16, device=DEV)
    dKV = torch.zeros((B * Tk, 2 * d), dtype=torch.bfloat16, device=DEV)
    a.dO, a.lddo = ptr(dO), d
    a.dQ, a.lddq = ptr(dQ), d
    a.dK, a.dV, a.lddk, a.lddv = ptr(dKV), ptr(dKV[:, d:]), 2 * d, 2 * d
    a.dq_scale = 0.125
    check(lib.kmb_op_attn_bwd(C.byref(a), stream()))

    def rows(x, T):
        return x.transpose(1, 2).reshape(B * T, d)

    assert rel_err(dQ, rows(qf.grad, Tq) * 0.125) < 2e-2
    assert rel_err(dKV[:, :d], rows(kf.grad, Tk)) < 2e-2
    assert rel_err(dKV[:, d:], rows(vf.grad, Tk)) < 2e-2


@pytest.mark.parametrize("Tq,Tk,causal,fused", [(64, 64, False, True), (32, 32, True, True), (32, 64, False, False), (20, 50, False, False)])
def test_attention_fwd_persistent_single_tile(Tq, Tk, causal, fused):
    """>= 1024 (batch, head) items of one query tile x one key tile run attn_fwd_small_kernel (persistent workgroups, the next
    item's loads in flight); the same items in batches of < 1024 run attn_fwd_kernel: same arithmetic, operation for operation,
    so the outputs and log-sum-exps must be IDENTICAL, and both match fp32 torch."""
    B, H = 160, 12          # 1920 items; ragged key masks
    d = H * 64
    lib = _lib.load()
    if fused:
        qkv = bf(rnd(B * Tq, 3 * d, seed=30, scale=0.7))
        Q, K, V = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    else:
        Q = bf(rnd(B * Tq, d, seed=31, scale=0.7))
        kv = bf(rnd(B * Tk, 2 * d, seed=32, scale=0.7))
        K, V = kv[:, :d], kv[:, d:]
    key_mask = torch.ones((B, Tk), dtype=torch.int64, device=DEV)
    for b in range(B):
        key_mask[b, Tk - (b % (Tk - 1)):] = 0 if b % 3 else 1
    O = torch.full((B * Tq, d), 7.0, dtype=torch.bfloat16, device=DEV)
    lse = torch.full((B, H, Tq), 7.0, dtype=torch.float32, device=DEV)
    a = attn_struct(Q, K, V, B, H, Tq, Tk, key_mask, causal, O, lse)
    check(lib.kmb_op_attn_fwd(C.byref(a), stream()))
    O2 = torch.full_like(O, 3.0)
    lse2 = torch.full_like(lse, 3.0)
    step = 40               # 480 items per call: the one-item-per-workgroup kernel
    for b0 in range(0, B, step):
        sl_q, sl_k = slice(b0 * Tq, (b0 + step) * Tq), slice(b0 * Tk, (b0 + step) * Tk)
        a2 = attn_struct(Q[sl_q], K[sl_k], V[sl_k], step, H, Tq, Tk, key_mask[b0: b0 + step], causal, O2[sl_q], lse2[b0: b0 + step])
        check(lib.kmb_op_attn_fwd(C.byref(a2), stream()))
    torch.cuda.synchronize()
    assert torch.equal(O, O2) and torch.equal(lse, lse2)

    def heads(x, T):
        return x.float().reshape(B, T, H, 64).transpose(1, 2)

    ref = ref_attention(heads(Q, Tq), heads(K, Tk), heads(V, Tk), key_mask, causal)
    assert rel_err(O, ref.transpose(1, 2).reshape(B * Tq, d)) < 1e-2


def test_attention_decode():
    lib = _lib.load()
    R, H, Tk, Tmax = 6, 2, 37, 48
    d = H * 64
    q = bf(rnd(R, d, seed=30, scale=0.5))
    Kc = bf(rnd(3, Tmax, d, seed=31, scale=0.5))
    Vc = bf(rnd(3, Tmax, d, seed=32))
    kv_row = torch.tensor([0, 0, 1, 1, 2, 2], dtype=torch.int32, device=DEV)
    mask = torch.ones((3, Tmax), dtype=torch.int64, device=DEV)
    mask[1, 30:] = 0
    O = torch.zeros((R, d), dtype=torch.bfloat16, device=DEV)
    a = KmbAttnDecode(Q=ptr(q), ldq=d, Kc=ptr(Kc), Vc=ptr(Vc), Tmax=Tmax, ldc=d, kv_row=ptr(kv_row),
                      key_mask=ptr(mask), mask_ld=Tmax, mask_row=ptr(kv_row), R=R, H=H, Tk=Tk, O=ptr(O), ldo=d)
    check(lib.kmb_op_attn_decode(C.byref(a), stream()))
    for r in range(R):
        c = int(kv_row[r])
        for h in range(H):
            s = Kc[c, :Tk, h * 64:(h + 1) * 64].float() @ q[r, h * 64:(h + 1) * 64].float()
            s = s.masked_fill(mask[c, :Tk] == 0, float("-inf"))
            ref = torch.softmax(s, 0) @ Vc[c, :Tk, h * 64:(h + 1) * 64].float()
            assert rel_err(O[r, h * 64:(h + 1) * 64], ref) < BF_TOL


# ----------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("M,D", [(70, 768), (33, 128), (16, 1024), (9, 2048)])   # 2048: the widest row the kernels take (106 KB of LDS in backward)
def test_layernorm_fwd_bwd(M, D):
    lib = _lib.load()
    z = bf(rnd(M, D, seed=40) * 2 + 0.3)
    gamma, beta = rnd(D, seed=41) * 0.1 + 1.0, rnd(D, seed=42) * 0.1
    y = torch.empty_like(z)
    mean = torch.empty(M, device=DEV)
    rstd = torch.empty(M, device=DEV)
    check(lib.kmb_op_ln_fwd(ptr(z), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), M, D, 1e-5, stream()))
    zf = z.float().requires_grad_(True)
    gf, bfp = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(zf, (D,), gf, bfp, 1e-5)
    assert rel_err(y, ref) < BF_TOL
    assert torch.allclose(mean, zf.mean(-1).detach(), atol=1e-5)
    dy = bf(rnd(M, D, seed=43))
    ref.backward(dy.float())
    dz = torch.empty_like(z)
    dg, db = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    scratch = torch.empty(int(lib.kmb_op_ln_bwd_scratch(M, D)), device=DEV)
    check(lib.kmb_op_ln_bwd(ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(dz), None, None, None, ptr(dg),
                            ptr(db), ptr(scratch), M, D, stream()))
    assert rel_err(dz, zf.grad) < BF_TOL
    assert rel_err(dg, gf.grad) < 1e-4
    assert rel_err(db, bfp.grad) < 1e-4


def test_layernorm_fwd_streamed_rows_same_bits():
    """>= 8192 rows go through ln_fwd_stream_kernel (resident waves, parameters in registers): the same bits as the one-row-per-wave kernel
    that the two halves of the same rows take."""
    lib = _lib.load()
    M, D = 8200 + 8192 * 3 + 5, 768   # the resident waves walk six or seven rows each
    z = bf(rnd(M, D, seed=44) * 3 - 0.2)
    gamma, beta = rnd(D, seed=45) * 0.1 + 1.0, rnd(D, seed=46) * 0.1
    outs = []
    for pieces in (1, 8):
        y = torch.zeros_like(z)
        mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
        step = (M + pieces - 1) // pieces
        assert pieces == 1 or step < 8192
        for r0 in range(0, M, step):
            n = min(step, M - r0)
            check(lib.kmb_op_ln_fwd(ptr(z[r0:]), ptr(gamma), ptr(beta), ptr(y[r0:]), ptr(mean[r0:]), ptr(rstd[r0:]), n, D, 1e-5, stream()))
        outs.append((y, mean, rstd))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert rel_err(outs[0][0], F.layer_norm(z.float(), (D,), gamma, beta, 1e-5)) < BF_TOL


@pytest.mark.parametrize("M,D", [(48, 256), (200, 768), (4100, 768)])
def test_layernorm_bwd_dropout_paths(M, D):
    """dy_drop masks the incoming gradient (embedding LN output dropout); out2 is dz under a second mask.  (d = 768 with an even row count: the
    two-rows-per-wave kernel of round 5; 4100 rows: several rows per wave and a ragged last block)"""
    lib = _lib.load()
    z = bf(rnd(M, D, seed=44))
    gamma = torch.ones(D, device=DEV)
    mean, rstd = z.float().mean(-1), (z.float().var(-1, unbiased=False) + 1e-5).rsqrt()
    dy = bf(rnd(M, D, seed=45))
    dz, out2 = torch.empty_like(z), torch.empty_like(z)
    dg, db = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    scratch = torch.empty(int(lib.kmb_op_ln_bwd_scratch(M, D)), device=DEV)
    thr = round(0.1 * 65536)
    sc = 1.0 / (1.0 - thr / 65536.0)
    d1, d2 = KmbDrop(thr, 77, sc), KmbDrop(thr, 99, sc)
    check(lib.kmb_op_ln_bwd(ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(dz), ptr(out2), C.byref(d1),
                            C.byref(d2), ptr(dg), ptr(db), ptr(scratch), M, D, stream()))
    k1, k2 = dropout_mask(77, 0.1, M, D), dropout_mask(99, 0.1, M, D)
    zf = z.float().requires_grad_(True)
    F.layer_norm(zf, (D,), gamma, None, 1e-5).backward(torch.where(k1, dy.float() * sc, torch.zeros(1, device=DEV)))
    assert rel_err(dz, zf.grad) < BF_TOL
    assert rel_err(out2, torch.where(k2, dz.float() * sc, torch.zeros(1, device=DEV))) < BF_TOL


def test_colsum():
    lib = _lib.load()
    for M, N in [(1000, 768), (37, 72), (5000, 2304)]:
        X = bf(rnd(M, N, seed=46))
        out = torch.empty(N, device=DEV)
        scratch = torch.empty(int(lib.kmb_op_colsum_scratch(M, N)), device=DEV)
        check(lib.kmb_op_colsum(ptr(X), N, M, N, ptr(out), ptr(scratch), stream()))
        assert rel_err(out, X.float().sum(0)) < 1e-4


# ----------------------------------------------------------------------------------- embedding
def test_embedding_forward_backward():
    lib = _lib.load()
    B, S, D, V, Fin, Fpad = 3, 20, 128, 300, 2052, 2056
    IMG, CLS, PAD = 290, 293, 1
    ids = torch.randint(3, 280, (B, S), device=DEV)
    ids[0, 2:7] = IMG
    ids[0, 4] = CLS
    ids[1, 1:3] = IMG
    ids[2, 5:9] = IMG      # sample 2 has placeholders but an EMPTY feature list -> stays token rows
    ids[2, -1] = PAD
    feat_off = torch.tensor([0, 5, 7, 7], dtype=torch.int32, device=DEV)
    img_src = torch.empty(B * S, dtype=torch.int32, device=DEV)
    status = torch.zeros(4, dtype=torch.int32, device=DEV)
    check(lib.kmb_op_img_rowmap(ptr(ids), ptr(feat_off), B, S, IMG, CLS, ptr(img_src), ptr(status), stream()))
    src = img_src.view(B, S).cpu()
    assert src[0, 2:7].tolist() == [0, 1, 2, 3, 4] and src[1, 1:3].tolist() == [5, 6]
    assert int((src[2] >= 0).sum()) == 0 and int(status[0]) == 0
    bad_off = torch.tensor([0, 4, 6, 6], dtype=torch.int32, device=DEV)  # 4 features for 5 placeholders
    check(lib.kmb_op_img_rowmap(ptr(ids), ptr(bad_off), B, S, IMG, CLS, ptr(img_src), ptr(status), stream()))
    assert int(status[0]) & 1
    check(lib.kmb_op_img_rowmap(ptr(ids), ptr(feat_off), B, S, IMG, CLS, ptr(img_src), ptr(status), stream()))

    feats = rnd(7, Fin, seed=50).abs()
    fb = torch.empty((7, Fpad), dtype=torch.bfloat16, device=DEV)
    check(lib.kmb_op_cast_pad(ptr(feats), 7, Fin, ptr(fb), Fpad, stream()))
    assert torch.equal(fb[:, :Fin], feats.to(torch.bfloat16)) and float(fb[:, Fin:].abs().sum()) == 0.0

    E, P = rnd(V, D, seed=51, scale=0.05), rnd(S + 2, D, seed=52, scale=0.05)
    img_emb = rnd(7, D, seed=53)
    gamma, beta = rnd(D, seed=54) * 0.1 + 1, rnd(D, seed=55) * 0.1
    M = B * S
    z, y = torch.empty((M, D), dtype=torch.bfloat16, device=DEV), torch.empty((M, D), dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    check(lib.kmb_op_embed_ln_fwd(ptr(ids), ptr(img_src), ptr(E), ptr(img_emb), ptr(P), 2, S, 1.0, ptr(gamma),
                                  ptr(beta), ptr(z), ptr(y), ptr(mean), ptr(rstd), M, D, 1e-5, None, stream()))
    emb = E[ids.view(-1)].clone()
    sel = img_src >= 0
    emb[sel] = img_emb[img_src[sel].long()]
    zr = emb + P[2:2 + S].repeat(B, 1)
    assert rel_err(z, zr) < BF_TOL
    assert rel_err(y, F.layer_norm(zr, (D,), gamma, beta, 1e-5)) < BF_TOL

    dz = bf(rnd(M, D, seed=56))
    dE = torch.zeros((V, D), device=DEV)
    dimg = torch.zeros((7, D), dtype=torch.bfloat16, device=DEV)
    check(lib.kmb_op_embed_bwd(ptr(dz), ptr(ids), ptr(img_src), 1.0, ptr(dE), ptr(dimg), PAD, M, D, stream()))
    ref_dE = torch.zeros((V, D), device=DEV)
    tok = (~sel) & (ids.view(-1) != PAD)
    ref_dE.index_add_(0, ids.view(-1)[tok], dz.float()[tok])
    assert rel_err(dE, ref_dE) < 1e-5
    assert torch.equal(dimg[img_src[sel].long()], dz[sel])
    dP = torch.full((S + 6, D), 3.0, device=DEV)
    check(lib.kmb_op_pos_bwd(ptr(dz), B, S, D, ptr(dP), 2, S + 6, stream()))
    assert rel_err(dP[2:2 + S], dz.float().view(B, S, D).sum(0)) < 1e-5
    assert float(dP[:2].abs().sum()) == 0.0 and float(dP[2 + S:].abs().sum()) == 0.0


# --------------------------------------------------------------------------------- loss / topk
def test_cross_entropy_and_grad():
    lib = _lib.load()
    rows, V, ld = 50, 50320, 50432
    logits = torch.zeros((rows, ld), device=DEV)
    logits[:, :V] = rnd(rows, V, seed=60) * 3
    logits[:, V:] = 1e30  # padding must be ignored
    labels = torch.randint(0, V, (rows,), device=DEV)
    labels[::7] = -100
    loss_rows = torch.empty(rows, device=DEV)
    dl = torch.empty((rows, ld), dtype=torch.bfloat16, device=DEV)
    count = torch.zeros(4, dtype=torch.int32, device=DEV)
    loss = torch.zeros(1, device=DEV)
    check(lib.kmb_op_ce(ptr(logits), ld, V, ptr(labels), rows, 1.0, ptr(loss_rows), ptr(dl), ptr(count), ptr(loss), stream()))
    lf = logits[:, :V].clone().requires_grad_(True)
    ref = F.cross_entropy(lf, labels)
    ref.backward()
    assert int(count[0]) == int((labels != -100).sum())
    assert abs(float(loss) - float(ref)) < 1e-4 * abs(float(ref))
    assert rel_err(dl[:, :V], lf.grad) < BF_TOL
    assert float(dl[:, V:].float().abs().sum()) == 0.0


class _TopkWithScratch:
    """kmb_logsoftmax_topk_ws behind kmb_logsoftmax_topk's signature (the decode loop's two-launch form)."""

    def __init__(self, lib, rows):
        self.lib = lib
        self.scr = torch.empty(int(lib.kmb_logsoftmax_topk_scratch(rows)), device=DEV)

    def kmb_logsoftmax_topk(self, *a):
        return self.lib.kmb_logsoftmax_topk_ws(*a[:-1], ptr(self.scr), self.scr.numel(), a[-1])


@pytest.mark.parametrize("ws", [False, True])
def test_logsoftmax_topk_degenerate_rows(ws):
    """Rows on which the split form's threshold selection has to give up (csrc/loss.hip wave_topk: more than 16 elements at
    or above the k-th value of a wave, or fewer than k finite values) -- the exact path; order = (value desc, index asc)."""
    lib = _lib.load()
    V, ld, k = 50320, 50432, 10
    base = rnd(8, V, seed=66) * 2
    rows_l = []
    rows_l.append(torch.zeros(V, device=DEV))                                   # every logit equal
    r = base[1].clone(); r[1000:1100] = 30.0; rows_l.append(r)                     # a hundred copies of the maximum
    r = torch.full((V,), -float("inf"), device=DEV); r[[7, 20000, 49000]] = torch.tensor([1.0, 3.0, 2.0], device=DEV); rows_l.append(r)
    r = base[3].clone(); r[12600:] = -float("inf"); rows_l.append(r)               # three of the four parts hold nothing finite
    r = base[4].clone(); r[::2] = r[1::2]; rows_l.append(r)                        # every value twice
    r = base[5].clone(); r[40000:40040] = r.max() + 1.0; rows_l.append(r)          # forty ties above everything, in one part
    r = torch.full((V,), -float("inf"), device=DEV); rows_l.append(r)              # nothing finite at all
    rows_l.append(base[7].clone())
    logits = torch.zeros((len(rows_l), ld), device=DEV)
    logits[:, :V] = torch.stack(rows_l)
    rows = logits.shape[0]
    if ws:
        lib = _TopkWithScratch(lib, rows)
    add = rnd(rows, seed=67)
    val = torch.empty((rows, k), device=DEV)
    idx = torch.empty((rows, k), dtype=torch.int32, device=DEV)
    check(lib.kmb_logsoftmax_topk(ptr(logits), ld, V, rows, ptr(add), -1, -1, k, ptr(val), ptr(idx), stream()))
    lp = torch.log_softmax(logits[:, :V], -1) + add[:, None]
    order = torch.sort(lp, dim=1, descending=True, stable=True)[1][:, :k]
    want = torch.gather(lp, 1, order)
    for i in range(rows):
        if i == 6:       # log_softmax of an all -inf row is NaN in torch; the kernel reports "no candidate" values
            continue
        finite = torch.isfinite(want[i])
        assert torch.equal(idx[i].long()[finite], order[i][finite]), i
        assert torch.allclose(val[i][finite], want[i][finite], atol=1e-4), i
        if i != 2:
            assert bool(finite.all()), i
    # row 2 has three finite values: they come first, in value order
    assert idx[2, :3].tolist() == [20000, 49000, 7]


@pytest.mark.parametrize("V,ld,ws", [(50320, 50432, False), (60000, 60032, False), (50320, 50432, True)])
def test_logsoftmax_topk(V, ld, ws):
    """one workgroup per row: register-resident kernel / 256-thread kernel; ws: rows split over four workgroups + combine"""
    lib = _lib.load()
    rows, k = 10, 10
    if ws:
        lib = _TopkWithScratch(lib, rows)
    logits = torch.zeros((rows, ld), device=DEV)
    logits[:, :V] = rnd(rows, V, seed=61) * 4
    add = rnd(rows, seed=62)
    val = torch.empty((rows, k), device=DEV)
    idx = torch.empty((rows, k), dtype=torch.int32, device=DEV)
    check(lib.kmb_logsoftmax_topk(ptr(logits), ld, V, rows, ptr(add), -1, -1, k, ptr(val), ptr(idx), stream()))
    lp = torch.log_softmax(logits[:, :V], -1) + add[:, None]
    rv, ri = torch.topk(lp, k, dim=1)
    assert torch.equal(idx.long(), ri)
    assert torch.allclose(val, rv, atol=1e-4)
    # ban_token (min_length): the EOS score is -inf AFTER the normalisation -- the other scores keep the full
    # log-sum-exp (masking the logit first would renormalise every row by -log(1 - p_eos))
    ban = int(ri[0, 0])
    check(lib.kmb_logsoftmax_topk(ptr(logits), ld, V, rows, ptr(add), -1, ban, k, ptr(val), ptr(idx), stream()))
    lpb = lp.clone()
    lpb[:, ban] = -float("inf")
    bv, bi = torch.topk(lpb, k, dim=1)
    assert torch.equal(idx.long(), bi) and torch.allclose(val, bv, atol=1e-4) and not bool((idx == ban).any())
    check(lib.kmb_logsoftmax_topk(ptr(logits), ld, V, rows, ptr(add), 2, -1, k, ptr(val), ptr(idx), stream()))
    assert torch.all(idx[:, 0] == 2) and torch.allclose(val[:, 0], add, atol=1e-6)
    assert torch.all(torch.isinf(val[:, 1:]))
    assert idx[0, 1:].tolist() == [0, 1] + list(range(3, k))
    # winners that share one thread's stripe (i % 256 equal), exact ties (lowest index first), a masked column
    logits[:, :V] = rnd(rows, V, seed=63)
    logits[:, 5:5 + 256 * 12:256] = 9.0 + torch.arange(12, device=DEV).float()
    logits[:, 7000] = logits[:, 5 + 256 * 11]
    logits[:, 300] = logits[:, 5 + 256 * 11]
    logits[:, 5 + 256 * 10] = -float("inf")
    check(lib.kmb_logsoftmax_topk(ptr(logits), ld, V, rows, ptr(add), -1, -1, k, ptr(val), ptr(idx), stream()))
    lp = torch.log_softmax(logits[:, :V], -1) + add[:, None]
    order = torch.sort(lp, dim=1, descending=True, stable=True)[1][:, :k]
    assert torch.equal(idx.long(), order)
    assert torch.allclose(val, torch.gather(lp, 1, order), atol=1e-4)
    # the same for the register-resident kernel's ownership (thread t holds the float4 chunks t, t + 1024, ...):
    # twelve winners inside ONE thread's chunks, two exact ties elsewhere, a masked winner
    logits[:, :V] = rnd(rows, V, seed=64)
    own = torch.tensor([4 * (5 + 1024 * j) + (j % 4) for j in range(12)], device=DEV)
    logits[:, own] = 9.0 + torch.arange(12, device=DEV).float()
    logits[:, 7001] = logits[:, own[11]]
    logits[:, 301] = logits[:, own[11]]
    logits[:, own[10]] = -float("inf")
    check(lib.kmb_logsoftmax_topk(ptr(logits), ld, V, rows, ptr(add), -1, -1, k, ptr(val), ptr(idx), stream()))
    lp = torch.log_softmax(logits[:, :V], -1) + add[:, None]
    order = torch.sort(lp, dim=1, descending=True, stable=True)[1][:, :k]
    assert torch.equal(idx.long(), order)
    assert torch.allclose(val, torch.gather(lp, 1, order), atol=1e-4)
    # and for the split form's ownership (thread t of a part holds the float4 chunks t, t + 256, ... of that part): twelve
    # winners inside ONE thread's chunks (its two cached keys run out: the rescan path), ties in other parts, a masked winner
    logits[:, :V] = rnd(rows, V, seed=65)
    own = torch.tensor([4 * (5 + 256 * j) + (j % 4) for j in range(12)], device=DEV)
    logits[:, own] = 9.0 + torch.arange(12, device=DEV).float()
    logits[:, 30001] = logits[:, own[11]]
    logits[:, 45001] = logits[:, own[11]]
    logits[:, own[10]] = -float("inf")
    check(lib.kmb_logsoftmax_topk(ptr(logits), ld, V, rows, ptr(add), -1, -1, k, ptr(val), ptr(idx), stream()))
    lp = torch.log_softmax(logits[:, :V], -1) + add[:, None]
    order = torch.sort(lp, dim=1, descending=True, stable=True)[1][:, :k]
    assert torch.equal(idx.long(), order)
    assert torch.allclose(val, torch.gather(lp, 1, order), atol=1e-4)


# ---------------------------------------------------------------------------------------- AdamW
def test_adamw_matches_hf_form():
    from oracle.kmbart_oracle import HFAdamW
    lib = _lib.load()
    n = 100003
    p0 = rnd(n, seed=70)
    p = p0.clone()
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    ref_p = torch.nn.Parameter(p0.clone().cpu())
    opt = HFAdamW([ref_p], lr=1e-3, eps=1e-6, weight_decay=0.01)
    for step in range(1, 4):
        g = rnd(n, seed=70 + step) * 0.1
        hp = KmbAdamW(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.01, step=step, correct_bias=1, grad_scale=1.0)
        check(lib.kmb_op_adamw(ptr(p), ptr(g), ptr(m), ptr(v), ptr(pb), n, C.byref(hp), stream()))
        ref_p.grad = g.cpu()
        opt.step()
    assert torch.allclose(p.cpu(), ref_p.detach(), atol=2e-7, rtol=1e-6)
    assert torch.equal(pb, p.to(torch.bfloat16))


@pytest.mark.parametrize("nb,k", [(5, 10), (8, 16), (16, 16)])
def test_beam_merge_matches_torch_topk_over_all_beams(nb, k):
    """kmb_beam_merge: per batch item the best k of its beams' top-k lists == torch.topk over the num_beams * V scores
    (ties: the earlier candidate position first), including forced-token steps where most candidates are -inf.
    nb * k = 50 / 128 / 256 candidates: one, two and four keys per lane of the merging wave."""
    lib = _lib.load()
    B, V, ld = 7, 50320, 50432
    logits = torch.zeros((B * nb, ld), device=DEV)
    logits[:, :V] = rnd(B * nb, V, seed=91) * 3
    logits[3, 100] = logits[3, 7]                    # an exact tie inside one row
    add = rnd(B * nb, seed=92)
    for force in (-1, 2):
        val = torch.empty((B * nb, k), device=DEV)
        idx = torch.empty((B * nb, k), dtype=torch.int32, device=DEV)
        check(lib.kmb_logsoftmax_topk(ptr(logits), ld, V, B * nb, ptr(add), force, -1, k, ptr(val), ptr(idx), stream()))
        out = torch.empty((B, k, 2), dtype=torch.int32, device=DEV)
        check(lib.kmb_beam_merge(ptr(val), ptr(idx), B, nb, k, V, ptr(out), stream()))
        got_scores = out[:, :, 0].contiguous().view(torch.float32)
        got_ids = out[:, :, 1].long()
        if force < 0:
            lp = torch.log_softmax(logits[:, :V], -1) + add[:, None]
        else:
            lp = torch.full((B * nb, V), -float("inf"), device=DEV)
            lp[:, force] = add
        flat = lp.view(B, nb * V)
        order = torch.sort(flat, dim=1, descending=True, stable=True)[1][:, :k]
        n_finite = int(torch.isfinite(torch.gather(flat, 1, order)).sum(1).min())
        assert torch.equal(got_ids[:, :n_finite], order[:, :n_finite])
        assert torch.allclose(got_scores[:, :n_finite], torch.gather(flat, 1, order)[:, :n_finite], atol=1e-4)
        assert bool(torch.isinf(got_scores[:, n_finite:]).all())
        # every candidate is distinct
        for b in range(B):
            assert len(set(got_ids[b].tolist())) == k


def test_beam_merge_select_picks_the_beams_the_host_bookkeeping_sends_on():
    """kmb_beam_merge_select: same candidates as kmb_beam_merge, and the next step's (score, token, cache row) per beam
    = the first num_beams non-EOS candidates in order -- the reference's selection loop (transformers 3.0.2
    _generate_beam_search) -- also when EOS is the best token of several beams and on a forced-EOS step."""
    lib = _lib.load()
    B, nb, V, ld, eos = 6, 5, 50320, 50432, 2
    k = 2 * nb
    logits = torch.zeros((B * nb, ld), device=DEV)
    logits[:, :V] = rnd(B * nb, V, seed=93) * 3
    logits[0:3, eos] = 40.0                          # EOS on top of three beams of item 0
    logits[7, eos] = 40.0
    add = rnd(B * nb, seed=94)
    for force in (-1, eos):
        val = torch.empty((B * nb, k), device=DEV)
        idx = torch.empty((B * nb, k), dtype=torch.int32, device=DEV)
        check(lib.kmb_logsoftmax_topk(ptr(logits), ld, V, B * nb, ptr(add), force, -1, k, ptr(val), ptr(idx), stream()))
        ref = torch.empty((B, k, 2), dtype=torch.int32, device=DEV)
        check(lib.kmb_beam_merge(ptr(val), ptr(idx), B, nb, k, V, ptr(ref), stream()))
        out = torch.empty((B, k, 2), dtype=torch.int32, device=DEV)
        ns = torch.empty(B * nb, device=DEV)
        nt = torch.empty(B * nb, dtype=torch.int64, device=DEV)
        ni = torch.empty(B * nb, dtype=torch.int32, device=DEV)
        check(lib.kmb_beam_merge_select(ptr(val), ptr(idx), B, nb, k, V, ptr(out), eos, ptr(ns), ptr(nt), ptr(ni), stream()))
        assert torch.equal(out, ref)
        scores = ref[:, :, 0].contiguous().view(torch.float32).cpu()
        ids = ref[:, :, 1].cpu()
        for b in range(B):
            want = [(float(scores[b, r]), int(ids[b, r]) % V, b * nb + int(ids[b, r]) // V) for r in range(k)
                    if int(ids[b, r]) % V != eos][:nb]
            assert len(want) == nb
            got = list(zip(ns[b * nb:(b + 1) * nb].tolist(), nt[b * nb:(b + 1) * nb].tolist(), ni[b * nb:(b + 1) * nb].tolist()))
            for (ws, wt, wi), (gs, gt, gi) in zip(want, got):
                assert (wt, wi) == (gt, gi) and (ws == gs or (ws != ws and gs != gs))


@pytest.mark.parametrize("B,nb,k", [(6, 5, 10), (64, 5, 10), (3, 8, 16), (2, 16, 16), (5, 1, 2)])
def test_beam_step_equals_topk_plus_merge_select(B, nb, k):
    """kmb_beam_step (the decode loop's form: the rows' top-k lists stay in the workgroup that merges them) returns exactly what
    kmb_logsoftmax_topk_ws + kmb_beam_merge_select return -- free, forced and min_length (banned EOS) steps."""
    lib = _lib.load()
    V, ld, eos = 50320, 50432, 2
    R = B * nb
    logits = torch.zeros((R, ld), device=DEV)
    logits[:, :V] = rnd(R, V, seed=95 + B) * 3
    logits[0, eos] = 40.0
    add = rnd(R, seed=96)
    scr = torch.empty(int(lib.kmb_logsoftmax_topk_scratch(R)), device=DEV)
    for force, ban in ((-1, -1), (eos, -1), (-1, eos), (0, -1)):
        val = torch.empty((R, k), device=DEV)
        idx = torch.empty((R, k), dtype=torch.int32, device=DEV)
        check(lib.kmb_logsoftmax_topk_ws(ptr(logits), ld, V, R, ptr(add), force, ban, k, ptr(val), ptr(idx), ptr(scr), scr.numel(), stream()))
        ref = torch.empty((B, k, 2), dtype=torch.int32, device=DEV)
        rs = torch.empty(R, device=DEV)
        rt = torch.empty(R, dtype=torch.int64, device=DEV)
        ri = torch.empty(R, dtype=torch.int32, device=DEV)
        check(lib.kmb_beam_merge_select(ptr(val), ptr(idx), B, nb, k, V, ptr(ref), eos, ptr(rs), ptr(rt), ptr(ri), stream()))
        out = torch.full((B, k, 2), -7, dtype=torch.int32, device=DEV)
        ns = torch.full((R,), -7.0, device=DEV)
        nt = torch.full((R,), -7, dtype=torch.int64, device=DEV)
        ni = torch.full((R,), -7, dtype=torch.int32, device=DEV)
        check(lib.kmb_beam_step(ptr(logits), ld, V, B, nb, ptr(add), force, ban, k, ptr(out), eos, ptr(ns), ptr(nt), ptr(ni),
                                ptr(scr), scr.numel(), stream()))
        torch.cuda.synchronize()
        assert torch.equal(out, ref), (force, ban)
        assert torch.equal(ns.view(torch.int32), rs.view(torch.int32)) and torch.equal(nt, rt) and torch.equal(ni, ri), (force, ban)
    with pytest.raises(RuntimeError):   # k beyond the fused form: the caller takes the two-call path
        check(lib.kmb_beam_step(ptr(logits), ld, V, B, nb, ptr(add), -1, -1, 17, ptr(out), eos, ptr(ns), ptr(nt), ptr(ni),
                                ptr(scr), scr.numel(), stream()))


@pytest.mark.parametrize("M,N,K", [(320, 50320, 768), (160, 50320, 768), (129, 1000, 128), (300, 4100, 64), (1, 512, 192)])
def test_gemm_all_rows_kernel_is_bit_identical(M, N, K):
    """The vocabulary projection of a decode step (R = batch x beams <= 320 rows) runs gemm_kernel_allrows: one workgroup per
    256 columns holds every row.  Same MFMA, same k order, same fp32 bias add as the tiled kernels: identical bits; columns
    past N and the output's padding stay untouched."""
    A = bf(rnd(M, K, seed=40, scale=0.5))
    Bm = bf(rnd(N, K, seed=41, scale=0.5))
    bias = rnd(N, seed=42)
    ld = ((N + 127) // 128) * 128
    out1 = torch.full((M, ld), 7.0, dtype=torch.float32, device=DEV)
    out2 = torch.full((M, ld), 7.0, dtype=torch.float32, device=DEV)
    gemm(A, Bm, bias=bias, out_f32=out1)
    gemm(A, Bm, bias=bias, out_f32=out2, allrows=True)
    torch.cuda.synchronize()
    assert torch.equal(out1, out2)
    assert bool((out2[:, N:] == 7.0).all())
    ref = A.float() @ Bm.float().t() + bias
    assert rel_err(out2[:, :N], ref) < 1e-5
    with pytest.raises(RuntimeError):      # more rows than one workgroup holds
        gemm(bf(rnd(321, K, seed=43)), Bm, bias=bias, out_f32=torch.empty((321, ld), device=DEV), allrows=True)
