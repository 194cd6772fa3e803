"""GPU: model-level correctness AT THE BATCH THE BENCHMARK RUNS (vcg_base, per-GPU b = 512 and 1024: Md = 16384 / 32768
decoder rows).  These sizes switch on code no smaller test reaches: the three-slice split of the tied 50320 x 768 weight
gradient and its 3 x V x d slab, the 256 x 192 persistent tile, the LM-head data-gradient split, the L2-prefetch rule.

  * loss and EVERY gradient of the big batch == the token-weighted sum over its 64-sample chunks (mean-token CE, reference
    src/model/model.py:400-402: whole-batch gradient = sum_c (n_c / n) x chunk gradient), each chunk run through the same
    engine at the reference's default batch (vcg_train.py:330) -- a path whose kernels (small-batch split-K, one-tile
    weight gradients) differ from the big batch's;
  * chunk 0 itself against the CPU oracle's autograd (loss and every gradient), so the chain big batch -> chunks -> oracle
    is closed at full model size;
  * weight gradients on the side stream vs all on one stream: every gradient bit for bit (the tied matrix to the last
    bit of its fp32 atomics).

Label lengths are ragged so the chunk weights differ; dropout is off (a chunk's rows would draw other masks)."""
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
CHUNK = 64


@pytest.fixture(scope="module")
def setup():
    import bench
    from oracle import goldenlib as G
    from oracle import kmbart_oracle as O
    from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration
    cfgd = dict(bench.VCG_BASE, dropout=0.0)
    ocfg = O.OracleConfig.from_dict(cfgd)
    sd = G.golden_state_dict(ocfg, seed=5)      # non-trivial biases / LayerNorm parameters
    m = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(cfgd))
    m.load_state_dict(sd, strict=False)
    return m.to(DEV).eval(), ocfg, sd


def _ragged_batch(bsz, seed):
    from src.data.synthetic import make_batch
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(1, 33, (bsz,), generator=g).tolist()
    regions = torch.randint(0, 37, (bsz,), generator=g).tolist()
    ev = [int(torch.randint(1, 64 - 5 - r + 1, (1,), generator=g)) for r in regions]
    return make_batch(bsz, seed=seed, regions=regions, event_lens=ev, label_lens=lab)


def _dev(b, lo=None, hi=None):
    sl = slice(lo, hi)
    out = {k: v[sl].to(DEV) for k, v in b.items() if torch.is_tensor(v)}
    out["image_features"] = [f.to(DEV) for f in b["image_features"][sl]]
    return out


def _fwd_bwd(model, d, scale=1.0):
    eng = model._engine
    loss, _, _ = eng.forward(d["input_ids"], d["image_features"], d["attention_mask"], d["decoder_input_ids"],
                             d["decoder_attention_mask"], d["labels"], train=False, need_grad=True, want_encoder=False)
    eng.backward(scale)
    torch.cuda.synchronize()
    return float(loss), eng.grads.clone()


def _per_param_err(eng, got, ref):
    out = []
    for name, (off, rows, cols) in eng.index.items():
        a, r = got[off: off + rows * cols].double(), ref[off: off + rows * cols].double()
        out.append((float((a - r).norm() / (r.norm() + 1e-30)), float(r.norm()), name))
    out.sort(reverse=True)
    return out


@pytest.mark.parametrize("bsz", [512, 1024])
def test_bench_batch_equals_the_token_weighted_sum_of_its_chunks(setup, bsz):
    model, _, _ = setup
    eng = model._engine
    b = _ragged_batch(bsz, seed=900 + bsz)
    n = [(b["labels"][c: c + CHUNK] != -100).sum().item() for c in range(0, bsz, CHUNK)]
    ntot = sum(n)
    assert len(set(n)) > 1
    loss_big, g_big = _fwd_bwd(model, _dev(b))
    eng.check_inputs()
    acc = torch.zeros_like(g_big, dtype=torch.float64)
    loss_acc = 0.0
    for i, c in enumerate(range(0, bsz, CHUNK)):
        w = n[i] / ntot
        lc, gc = _fwd_bwd(model, _dev(b, c, c + CHUNK), scale=w)   # backward at loss scale w: the chunk's contribution
        acc += gc.double()
        loss_acc += w * lc
    assert abs(loss_big - loss_acc) <= 2e-4 * abs(loss_acc), (loss_big, loss_acc)
    errs = _per_param_err(eng, g_big, acc)
    print(f"[b={bsz}] loss {loss_big:.6f} vs chunk-weighted {loss_acc:.6f}; worst gradient errors:", errs[:6])
    # the big batch and its chunks hold the same per-row quantities in bf16, rounded after scaling by 1/n resp. w/n_c
    # (not powers of two), and sum them in different orders: ~2^-9 per element, norm-wise well under 1e-2.
    # k_proj.bias has a zero true gradient (softmax is shift-invariant): rounding noise only.
    worst = max(e for e, nr, name in errs if "k_proj.bias" not in name)
    assert worst < 2e-2, errs[:6]     # measured 7.5e-3 at b = 1024 (decoder q / k projections)
    # and the side stream changes nothing: same batch with every weight gradient on the caller's stream
    from kmbart import _lib
    lib = _lib.load()
    lib.kmb_set_side_stream(eng.h, 0)
    try:
        loss_ser, g_ser = _fwd_bwd(model, _dev(b))
    finally:
        lib.kmb_set_side_stream(eng.h, 1)
    loss_again, g_again = _fwd_bwd(model, _dev(b))
    off, rows, cols = eng.index["model.shared.weight"]
    for tag, g in (("serial", g_ser), ("side stream again", g_again)):
        same = g == g_big
        same[off: off + rows * cols] = True
        assert bool(same.all()), (tag, int((~same).sum()))
        assert torch.allclose(g[off: off + rows * cols], g_big[off: off + rows * cols], rtol=0, atol=1e-5), tag
    assert loss_ser == loss_big == loss_again


def test_chunk_zero_against_the_oracle(setup):
    """b = 64 (the reference's default per-GPU batch) at full model size: loss and every gradient vs oracle autograd."""
    from oracle import kmbart_oracle as O
    from test_fullsize_parity_gpu import check_grads
    model, ocfg, sd = setup
    b = _ragged_batch(512, seed=900 + 512)
    c = {k: (v[:CHUNK] if torch.is_tensor(v) else v[:CHUNK]) for k, v in b.items()}
    osd = {k: v.clone().requires_grad_(k != "final_logits_bias") for k, v in sd.items()}
    ref_loss, _, _ = O.forward(osd, ocfg, c["input_ids"], c["image_features"], c["attention_mask"], c["decoder_input_ids"],
                               c["decoder_attention_mask"], c["labels"])
    ref_loss.backward()
    d = _dev(c)
    loss = model(input_ids=d["input_ids"], image_features=d["image_features"], attention_mask=d["attention_mask"],
                 decoder_input_ids=d["decoder_input_ids"], decoder_attention_mask=d["decoder_attention_mask"],
                 labels=d["labels"])[0]
    dl = abs(float(loss) - float(ref_loss)) / float(ref_loss)
    print(f"[vcg_base ragged b=64] loss {float(loss):.6f} vs oracle {float(ref_loss):.6f} (rel {dl:.2e})")
    assert dl < 1e-3
    loss.backward()
    torch.cuda.synchronize()
    check_grads(model, {k: v.grad for k, v in osd.items() if v.grad is not None}, "vcg_base ragged b=64")


@pytest.mark.parametrize("bsz", [64, 512])
def test_cross_entropy_fused_into_the_head_gemm_vs_the_two_kernel_path(setup, bsz):
    """The tied head's loss two ways (engine.cpp forward_impl): the head GEMM's epilogue storing exp(logit - label logit)
    plus row sums (no pass over the logits), against bf16 logits + the softmax kernel (KMB_FUSED_CE=0).  Same mathematics
    (reference src/model/model.py:398-402), other roundings: the probabilities are bf16 of exp(v - c) instead of bf16 of
    the scaled difference, the per-row factor multiplies H (bf16) for the weight gradient and the fp32 sum for dH."""
    model, _, _ = setup
    eng = model._engine
    b = _ragged_batch(bsz, seed=77 + bsz)
    d = _dev(b)
    os.environ["KMB_FUSED_CE"] = "0"
    try:
        loss_two, g_two = _fwd_bwd(model, d)
    finally:
        os.environ.pop("KMB_FUSED_CE", None)
    loss_fused, g_fused = _fwd_bwd(model, d)
    loss_fused2, g_fused2 = _fwd_bwd(model, d)
    print(f"[b={bsz}] loss fused {loss_fused:.6f} vs two-kernel {loss_two:.6f}")
    assert abs(loss_fused - loss_two) <= 2e-4 * abs(loss_two)
    errs = _per_param_err(eng, g_fused, g_two.double())
    print("worst gradient differences:", errs[:6])
    worst = max(e for e, nr, name in errs if "k_proj.bias" not in name)
    assert worst < 1.5e-2, errs[:6]
    # deterministic apart from the tied matrix's fp32 atomics
    off, rows, cols = eng.index["model.shared.weight"]
    same = g_fused == g_fused2
    same[off: off + rows * cols] = True
    assert bool(same.all()) and loss_fused == loss_fused2
    assert torch.allclose(g_fused[off: off + rows * cols], g_fused2[off: off + rows * cols], rtol=0, atol=1e-5)
