"""GPU: the fused decode blocks (csrc/decode.hip) -- [LayerNorm ->] projection [-> attention] of a KV-cached decode
step -- at the benchmark's full size (d = 768, 12 heads, ffn 3072, V = 50320):

  * every kind of block against a plain torch fp32 computation on the same bf16 inputs (through the C-ABI op);
  * the whole decode path against the CPU oracle: teacher-forced step-by-step logits of `kmb_gen_step` (3 beams per
    batch item, ragged regions and a padded encoder input) vs the oracle's full decoder forward over the same tokens
    (reference src/model/modules.py decoder with use_cache == without);
  * fused vs launch-per-operation path (KMB_GEN_FUSED=0) on the same steps.
Tolerances: bf16 storage with fp32 accumulation, as in test_fullsize_parity_gpu.py."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, bf, rel_err, stream  # noqa: E402
from kmbart import _lib  # noqa: E402
from kmbart._lib import KmbDecodeBlock, check, ptr  # noqa: E402
from oracle import goldenlib as G  # noqa: E402
from oracle import kmbart_oracle as O  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402
from test_fullsize_parity_gpu import BASE  # noqa: E402


def _ln(z, g, b, eps=1e-5):
    z = z.float()
    mu = z.mean(-1, keepdim=True)
    var = ((z - mu) ** 2).mean(-1, keepdim=True)
    return (z - mu) * torch.rsqrt(var + eps) * g + b


def _pack(W):
    out = torch.empty_like(W)
    check(_lib.load().kmb_op_decode_pack(ptr(W), W.stride(0), W.shape[0], W.shape[1], ptr(out), stream()))
    return out


def _block(**kw):
    b = KmbDecodeBlock()
    keep = []
    kw["W"] = _pack(kw["W"])     # the blocks read the weight in fragment order
    for k, v in kw.items():
        if torch.is_tensor(v):
            keep.append(v)
            v = ptr(v)
        setattr(b, "in_" if k == "inp" else k, v)
    check(_lib.load().kmb_op_decode_block(C.byref(b), stream()))
    torch.cuda.synchronize()
    return keep


@pytest.mark.parametrize("R,K,N,ln,act,res", [(320, 768, 768, False, 0, True), (320, 768, 3072, True, 1, False),
                                               (320, 3072, 768, False, 0, True), (37, 768, 768, True, 0, True),
                                               (5, 1536, 128, False, 0, False), (40, 768, 1536, True, 1, False),
                                               # more rows than one round of workgroups
                                               (700, 768, 768, True, 0, True), (1300, 768, 3072, True, 1, False),
                                               (1300, 2304, 768, False, 0, True), (650, 3072, 768, False, 0, True)])
def test_projection_block(R, K, N, ln, act, res):
    torch.manual_seed(R + K + N)
    x = bf(torch.randn(R, K, device=DEV) * 1.5 + 0.3)
    W = bf(torch.randn(N, K, device=DEV) * 0.04)
    bias = torch.randn(N, device=DEV) * 0.1
    g = torch.rand(K, device=DEV) + 0.5
    be = torch.randn(K, device=DEV) * 0.1
    r = bf(torch.randn(R, N, device=DEV))
    out = torch.zeros(R, N, dtype=torch.bfloat16, device=DEV)
    ln_out = torch.zeros(R, K, dtype=torch.bfloat16, device=DEV)
    kw = dict(kind=0, inp=x, ld_in=K, W=W, bias=bias, R=R, K=K, N=N, act=act, out=out, ld_out=N, eps=1e-5)
    if ln:
        kw.update(gamma=g, beta=be, ln_out=ln_out)
    if res:
        kw.update(residual=r, ld_res=N)
    _block(**kw)
    a = _ln(x, g, be) if ln else x.float()
    if ln:
        assert rel_err(ln_out, a) < 4e-3
        a = ln_out.float()     # the projection consumes the bf16-rounded normalised rows
    y = a @ W.float().t() + bias
    if act:
        y = torch.nn.functional.gelu(y)
    if res:
        y = y + r.float()
    e = rel_err(out, y)
    print(f"[decode projection R={R} K={K} N={N} ln={ln} act={act} res={res}] rel {e:.2e}")
    assert e < 4e-3


@pytest.mark.parametrize("R,Tk", [(320, 1), (320, 7), (37, 20), (16, 64), (700, 9), (1290, 19)])
def test_self_attention_block(R, Tk):
    torch.manual_seed(R * 31 + Tk)
    H, d, Tmax = 12, 768, max(Tk, 20)
    z = bf(torch.randn(R, d, device=DEV))
    g = torch.rand(d, device=DEV) + 0.5
    be = torch.randn(d, device=DEV) * 0.1
    W = bf(torch.randn(3 * d, d, device=DEV) * 0.05)
    bias = torch.randn(3 * d, device=DEV) * 0.1
    Kc = bf(torch.randn(R, Tmax, d, device=DEV))
    Vc = bf(torch.randn(R, Tmax, d, device=DEV))
    K0, V0 = Kc.clone(), Vc.clone()
    out = torch.zeros(R, d, dtype=torch.bfloat16, device=DEV)
    ln_out = torch.zeros(R, d, dtype=torch.bfloat16, device=DEV)
    _block(kind=1, inp=z, ld_in=d, gamma=g, beta=be, eps=1e-5, ln_out=ln_out, W=W, bias=bias, R=R, K=d, N=3 * d, out=out,
           ld_out=d, H=H, q_scale=0.125, Kc=Kc, Vc=Vc, Tmax=Tmax, ldc=d, Tk=Tk)
    x = ln_out.float()
    assert rel_err(ln_out, _ln(z, g, be)) < 4e-3
    qkv = x @ W.float().t() + bias
    q = bf(qkv[:, :d] * 0.125).float()
    k = bf(qkv[:, d:2 * d])
    v = bf(qkv[:, 2 * d:])
    # the new key / value row went into the cache at Tk - 1, nothing else changed
    assert rel_err(Kc[:, Tk - 1], k) < 4e-3 and rel_err(Vc[:, Tk - 1], v) < 4e-3
    keep = torch.ones(Tmax, dtype=torch.bool, device=DEV)
    keep[Tk - 1] = False
    assert torch.equal(Kc[:, keep], K0[:, keep]) and torch.equal(Vc[:, keep], V0[:, keep])
    Kf, Vf = Kc[:, :Tk].float().view(R, Tk, H, 64), Vc[:, :Tk].float().view(R, Tk, H, 64)
    s = torch.einsum("rhe,rthe->rht", q.view(R, H, 64), Kf)
    o = torch.einsum("rht,rthe->rhe", torch.softmax(s, -1), Vf).reshape(R, d)
    e = rel_err(out, o)
    print(f"[decode self-attention R={R} Tk={Tk}] rel {e:.2e}")
    assert e < 6e-3


@pytest.mark.parametrize("B,nb,S,grouped", [(64, 5, 100, True), (64, 5, 100, False), (3, 4, 37, True), (7, 6, 120, True),
                                            (5, 3, 50, True), (2, 5, 130, True), (140, 5, 100, True), (259, 5, 100, True),
                                            (130, 5, 100, False)])
def test_cross_attention_block(B, nb, S, grouped):
    """grouped: kv_group = beams per item (keys / values staged once per item in LDS when the tile allows it);
    3 beams per item and 130 keys fall back to per-row reads inside the same call."""
    torch.manual_seed(B + S)
    H, d, R = 12, 768, B * nb
    z = bf(torch.randn(R, d, device=DEV))
    g = torch.rand(d, device=DEV) + 0.5
    be = torch.randn(d, device=DEV) * 0.1
    W = bf(torch.randn(3 * d, d, device=DEV) * 0.05)     # q | k | v rows as the engine stores them; only q is read
    bias = torch.randn(3 * d, device=DEV) * 0.1
    ckv = bf(torch.randn(B, S, 2 * d, device=DEV))
    mask = torch.ones(B, S, dtype=torch.int64, device=DEV)
    for b in range(B):
        mask[b, S - (b % 7):] = 0
    kv_row = (torch.arange(R, device=DEV) // nb).to(torch.int32)
    out = torch.zeros(R, d, dtype=torch.bfloat16, device=DEV)
    ln_out = torch.zeros(R, d, dtype=torch.bfloat16, device=DEV)
    Vc = ckv.view(-1)[d:]
    _block(kind=2, inp=z, ld_in=d, gamma=g, beta=be, eps=1e-5, ln_out=ln_out, W=W[:d], bias=bias, R=R, K=d, N=d, out=out,
           ld_out=d, H=H, q_scale=0.125, Kc=ckv, Vc=Vc, Tmax=S, ldc=2 * d, Tk=S, kv_row=kv_row, key_mask=mask, mask_ld=S, kv_group=nb if grouped else 0)
    x = ln_out.float()
    q = bf((x @ W[:d].float().t() + bias[:d]) * 0.125).float().view(R, H, 64)
    Kf = ckv[:, :, :d].float().view(B, S, H, 64)[kv_row.long()]
    Vf = ckv[:, :, d:].float().view(B, S, H, 64)[kv_row.long()]
    s = torch.einsum("rhe,rthe->rht", q, Kf)
    s = s.masked_fill(mask[kv_row.long()][:, None, :] == 0, float("-inf"))
    o = torch.einsum("rht,rthe->rhe", torch.softmax(s, -1), Vf).reshape(R, d)
    e = rel_err(out, o)
    print(f"[decode cross-attention B={B} beams={nb} S={S} grouped={grouped}] rel {e:.2e}")
    assert e < 6e-3


def _teacher_forced_logits(model, b, nb, T, fused):
    os.environ["KMB_GEN_FUSED"] = "1" if fused else "0"
    try:
        eng = model._engine
        B = b["input_ids"].shape[0]
        eng.gen_begin(b["input_ids"].to(DEV), [f.to(DEV) for f in b["image_features"]], b["attention_mask"].to(DEV), nb, T + 1)
        eng.check_inputs()
        out = []
        for t in range(T):
            tok = b["decoder_input_ids"][:, t].repeat_interleave(nb).to(DEV)
            lg = eng.gen_step(tok, t)[:, : model.config.vocab_size].float().clone()
            out.append(lg.view(B, nb, -1))
            # an identity reorder: exercises the cache ping-pong exactly as generate() does
            eng.gen_reorder(torch.arange(B * nb, dtype=torch.int32, device=DEV), t)
        torch.cuda.synchronize()
        return torch.stack(out, dim=2)     # [B, nb, T, V]
    finally:
        os.environ.pop("KMB_GEN_FUSED", None)


def test_decode_steps_match_oracle_and_unfused_path():
    ocfg = O.OracleConfig.from_dict(BASE)
    sd = G.golden_state_dict(ocfg, seed=5)
    T, nb = 6, 3
    b = make_batch(2, seed=1234, regions=(36, 20), event_lens=(23, 7), label_lens=(32, 19))
    with torch.no_grad():
        _, ref, _ = O.forward(sd, ocfg, b["input_ids"], b["image_features"], b["attention_mask"],
                              b["decoder_input_ids"][:, :T], torch.ones(2, T, dtype=torch.long), None)
    model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(BASE))
    model.load_state_dict(sd, strict=False)
    model.to(DEV).eval()
    fused = _teacher_forced_logits(model, b, nb, T, True).cpu()
    plain = _teacher_forced_logits(model, b, nb, T, False).cpu()
    for name, got in (("fused", fused), ("launch-per-op", plain)):
        worst = max(rel_err(got[:, j], ref) for j in range(nb))
        print(f"[decode {name}] teacher-forced logits vs oracle, {T} steps x {nb} beams: worst norm-wise rel {worst:.2e}")
        assert worst < 2e-2
        # beams of one batch item were fed the same tokens: identical rows
        assert torch.equal(got[:, 0], got[:, 1]) and torch.equal(got[:, 0], got[:, 2])
    d = rel_err(fused, plain)
    print(f"[decode] fused vs launch-per-op logits: rel {d:.2e}")
    assert d < 1.5e-2
