"""GPU: the head GEMM's exp-with-row-sums epilogue (KmbGemm.act = 5, include/kmbart.h) against fp32 torch: the stored
bf16 matrix exp(A B^T + bias - shift[row]), the per-row partial sums (one slot per 64 columns), the shifted value picked at
each row's label column (rows labelled -100 are left alone), on every variant that carries the epilogue (persistent
256x256 / 256x128 with four waves, 256x256 with eight) -- all three bit-identical.  This is the forward half of the
tied-head cross-entropy (reference src/model/model.py:398-402: CrossEntropyLoss over F.linear(h, shared.weight) + bias)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "km-bart_amd"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(M, N, K, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    B = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    bias = (torch.randn(N, generator=g) * 0.3).to(DEV)
    bias[-37:] = -1e30                                    # padded vocabulary columns
    labels = torch.randint(0, N - 37, (M,), generator=g)
    labels[::7] = -100
    labels = labels.to(DEV)
    v = A.float() @ B.float().t() + bias
    shift = v.gather(1, labels.clamp(min=0)[:, None])[:, 0] + (torch.randn(M, generator=g) * 0.01).to(DEV)
    return A, B, bias, labels, shift.contiguous(), v


def _run_and_check(M, N, K, tag):
    """one launch in this process (variant: whatever KMB_GEMM_VARIANT forces, else the tuner's pick); returns checksums"""
    import hashlib
    from gpu_util import gemm
    A, B, bias, labels, shift, v = _case(M, N, K, seed=M + N)
    ref = torch.exp(v - shift[:, None])
    P = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
    sums = torch.full((M, N // 64), float("nan"), device=DEV)
    pick = torch.full((M,), 123.0, device=DEV)
    gemm(A, B, bias=bias, act=5, out_bf16=P, row_shift=shift, row_sums=sums, pick_col=labels, pick_out=pick)
    torch.cuda.synchronize()
    # bf16 inputs, fp32 accumulation, one bf16 rounding of the stored value
    err = ((P.float() - ref).abs() / (ref.abs() + 1e-6)).max().item()
    assert err < 6e-3, (tag, err)
    assert bool((P[:, -37:] == 0).all())
    assert torch.allclose(sums.sum(1), ref.sum(1), rtol=2e-4, atol=0), tag
    valid = labels >= 0
    want = (v.gather(1, labels.clamp(min=0)[:, None])[:, 0] - shift)
    assert torch.allclose(pick[valid], want[valid], rtol=0, atol=2e-3), tag
    assert bool((pick[~valid] == 123.0).all())
    h = hashlib.sha256()
    h.update(P.view(torch.int16).cpu().numpy().tobytes())
    h.update(pick.cpu().numpy().tobytes())
    return h.hexdigest(), sums.sum(1).cpu()


@pytest.mark.parametrize("M,N,K", [(1024, 8192, 768), (2048, 50432, 768), (4096, 2048, 256)])
def test_exp_epilogue_against_torch(M, N, K):
    import subprocess
    mine, _ = _run_and_check(M, N, K, "tuner's pick")
    for variant in ("11", "12", "14"):     # the variant is fixed per process (read once): one child per variant
        env = dict(os.environ, KMB_GEMM_VARIANT=variant)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(M), str(N), str(K)], capture_output=True, text=True,
                           timeout=900, env=env)
        assert r.returncode == 0, (variant, r.stdout[-2000:], r.stderr[-2000:])
        assert ("sha " + mine) in r.stdout, (variant, mine, r.stdout[-300:])


def test_exp_epilogue_saturates_instead_of_overflowing():
    """A row whose logits exceed the shift by more than 80 (a label the model gives probability e^-80) stores 2^115 and a
    finite row sum; every other row is untouched."""
    from gpu_util import gemm
    M, N, K = 1024, 8192, 768
    A, B, bias, labels, shift, v = _case(M, N, K, seed=11)
    shift2 = shift.clone()
    shift2[5] -= 300.0
    P = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    sums = torch.empty((M, N // 64), device=DEV)
    pick = torch.zeros((M,), device=DEV)
    gemm(A, B, bias=bias, act=5, out_bf16=P, row_shift=shift2, row_sums=sums, pick_col=labels, pick_out=pick)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(P.float()).all()) and bool(torch.isfinite(sums).all())
    assert float(P[5, : N - 37].float().min()) == 2.0 ** 115
    ref = torch.exp(v - shift[:, None])
    keep = torch.ones(M, dtype=torch.bool, device=DEV)
    keep[5] = False
    assert ((P.float() - ref).abs() / (ref.abs() + 1e-6))[keep].max().item() < 6e-3


def test_exp_epilogue_rejects_shapes_it_cannot_run():
    from gpu_util import gemm
    A, B, bias, labels, shift, _ = _case(1024, 8192, 768, seed=3)
    P = torch.empty((1000, 8192), dtype=torch.bfloat16, device=DEV)
    sums = torch.empty((1000, 128), device=DEV)
    pick = torch.empty((1000,), device=DEV)
    with pytest.raises(RuntimeError):      # M not a multiple of 256
        gemm(A[:1000], B, bias=bias, act=5, out_bf16=P, row_shift=shift, row_sums=sums, pick_col=labels, pick_out=pick)
    with pytest.raises(RuntimeError):      # no row sums
        gemm(A, B, bias=bias, act=5, out_bf16=torch.empty((1024, 8192), dtype=torch.bfloat16, device=DEV))


if __name__ == "__main__":
    sha, _ = _run_and_check(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), os.environ.get("KMB_GEMM_VARIANT", "?"))
    print("sha " + sha)
