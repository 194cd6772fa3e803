"""GPU: grouped weight gradients (csrc/gemm.hip gemm_group_wgrad_kernel, kmb_op_gemm_group; round 5).  With a short token
reduction (the reference's default batch of 64, vcg_train.py:330) kmb_backward sends a layer's four to six weight gradients
out as ONE launch over all their 128 x 128 tiles instead of one split-K GEMM + slab reduction each.

  * every output of a grouped launch has the SAME BITS as the same problem through kmb_op_gemm (same tile body, same
    accumulation order), for mixed shapes, edge tiles, padded leading dimensions, beta accumulation;
  * problems the kernel cannot take are refused with a message;
  * a training step with the grouped path and with KMB_WGRAD_GROUP=0 (two processes, same seed, dropout on) agrees on the
    loss bit for bit (forward is untouched) and on every weight gradient to fp32 summation order."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from kmbart import _lib  # noqa: E402
from kmbart._lib import KmbGemm, ptr  # noqa: E402
from gpu_util import DEV, bf, gemm, stream  # noqa: E402


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def wgrad_problem(dY, X, out, N=None, K=None, beta=0.0):
    """dW[N, K] = dY[T, N]^T X[T, K] (+ beta * dW)"""
    g = KmbGemm()
    g.A, g.B = ptr(dY), ptr(X)
    g.lda, g.ldb = dY.stride(0), X.stride(0)
    g.a_kc, g.b_kc = 0, 0
    g.M, g.N, g.K = (dY.shape[1] if N is None else N), (X.shape[1] if K is None else K), dY.shape[0]
    g.col_scale, g.drop_scale = 1.0, 1.0
    g.out_f32, g.ld_out_f32 = ptr(out), out.stride(0)
    g.beta = beta
    return g


SHAPES = [  # (tokens, out features, in features, padded ld of dY, padded ld of X)
    (4096, 768, 768, 768, 768), (4096, 2304, 768, 2304, 768), (2048, 768, 3072, 768, 3072), (2048, 3072, 768, 3072, 768),
    (192, 200, 136, 208, 136), (64, 128, 72, 128, 72), (2304, 768, 2052, 768, 2112), (128, 8, 8, 8, 8),
]


@pytest.mark.parametrize("count", [1, 2, 6, 8])
def test_grouped_launch_is_bit_identical_to_single_launches(count):
    lib = _lib.load()
    keep, probs, outs, refs = [], (KmbGemm * count)(), [], []
    for i, (T, N, K, ldy, ldx) in enumerate(SHAPES[:count]):
        dY = bf(rnd(T, ldy, seed=10 + i))
        X = bf(rnd(T, ldx, seed=50 + i))
        beta = 0.5 if i == 1 else 0.0
        init = rnd(N, K, seed=90 + i)
        out, ref = init.clone(), init.clone()
        probs[i] = wgrad_problem(dY, X, out, N=N, K=K, beta=beta)
        gemm(dY, X, a_kc=False, b_kc=False, M=N, N=K, K=T, out_f32=ref, beta=beta)
        keep += [dY, X]
        outs.append(out)
        refs.append(ref)
    _lib.check(lib.kmb_op_gemm_group(probs, count, stream()))
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert torch.equal(o, r), "problem %d %s differs: max |d| %g" % (i, SHAPES[i], float((o - r).abs().max()))
    # and against a plain fp32 product of the same bf16 operands
    T, N, K, ldy, ldx = SHAPES[0]
    ref = keep[0][:, :N].float().t() @ keep[1][:, :K].float()
    assert float((outs[0] - ref).norm() / ref.norm()) < 1e-5


def test_grouped_launch_refuses_what_it_cannot_run():
    lib = _lib.load()
    dY, X = bf(rnd(128, 128, seed=1)), bf(rnd(128, 128, seed=2))
    out = torch.zeros((128, 128), dtype=torch.float32, device=DEV)
    good = wgrad_problem(dY, X, out)

    def err(p, n=1):
        arr = (KmbGemm * max(n, 1))()
        for i in range(max(n, 1)):
            arr[i] = p
        rc = lib.kmb_op_gemm_group(arr, n, stream())
        return rc, (lib.kmb_last_error() or b"").decode()

    assert err(good)[0] == 0
    fwd = wgrad_problem(dY, X, out)
    fwd.a_kc, fwd.b_kc = 1, 1
    rc, msg = err(fwd)
    assert rc != 0 and "weight-gradient layout" in msg
    short = wgrad_problem(dY[:96], X[:96], out)   # 96 tokens: not a multiple of the 64-deep K step
    rc, msg = err(short)
    assert rc != 0 and "multiple of 64" in msg
    sp = wgrad_problem(dY, X, out)
    sp.split_k = 2
    rc, msg = err(sp)
    assert rc != 0
    rc, msg = err(good, n=9)
    assert rc != 0 and "problems" in msg
    torch.cuda.synchronize()


_CHILD = r"""
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "km-bart_amd"))
import torch
from src.data.synthetic import make_batch
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration
from tests.test_fullsize_parity_gpu import BASE
torch.manual_seed(0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(dict(BASE, dropout=0.1))).to("cuda:0").train()
model._engine.set_seed(77)
b = make_batch(8, seed=99)
batch = {k: (v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in b.items()}
batch["image_features"] = [f.to("cuda:0") for f in b["image_features"]]
loss = model.train_step_fwd_bwd(batch)
torch.cuda.synchronize()
eng = model._engine
g = eng.grads
out = {"loss": float(loss), "norm": {}, "head": {}}
for n, (o, r, c) in eng.index.items():
    v = g[o: o + r * c].double()
    out["norm"][n] = float(v.norm())
    if r > 1 and n.endswith("weight") and ("layers.0." in n or "layers.5." in n):
        out["head"][n] = v[:512].cpu().numpy().tolist()
print("JSON" + json.dumps(out))
"""


def test_training_step_grouped_against_one_launch_per_weight_gradient():
    res = {}
    for flag in ("0", "1"):
        env = dict(os.environ, KMB_WGRAD_GROUP=flag)
        r = subprocess.run([sys.executable, "-c", _CHILD % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=600,
                           cwd=ROOT)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("JSON")]
        assert r.returncode == 0 and line, r.stderr[-3000:]
        res[flag] = json.loads(line[0][4:])
    a, b = res["0"], res["1"]
    assert a["loss"] == b["loss"]
    worst = 0.0
    for n, na in a["norm"].items():
        nb = b["norm"][n]
        worst = max(worst, abs(na - nb) / (abs(na) + 1e-30))
    assert worst < 1e-5, worst
    for n, va in a["head"].items():
        ta, tb = torch.tensor(va), torch.tensor(b["head"][n])
        assert float((ta - tb).norm() / (ta.norm() + 1e-30)) < 1e-5, n
    print("grouped vs single weight-gradient launches: worst gradient-norm difference %.2e over %d parameters" % (worst, len(a["norm"])))
