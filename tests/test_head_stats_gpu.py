"""The decode step's vocabulary projection with its statistics epilogue (csrc/gemm.hip gemm_kernel_allrows<true>) and the beam
step that selects from those statistics (csrc/loss.hip beam_stats_merge_kernel; C-ABI kmb_gen_step + kmb_gen_beam_step, op-level
kmb_op_gemm_allrows_stats + kmb_beam_step_stats) against the two-launch beam step over the logits (kmb_beam_step), which the
oracle tests pin (tests/test_ops_gpu.py, tests/test_decode_fused_gpu.py).  Reference: one step of transformers 3.0.2
_generate_beam_search as reached from /root/reference/src/model/mixins.py:336-361 (scores: mixins.py:386-417).

Bar: logits bit-identical to the plain all-rows kernel; block maxima exact; block sums within 1e-5 relative; candidate tokens, next
tokens and next beams IDENTICAL to kmb_beam_step; scores within 4e-6 absolute (the log-sum-exp is grouped by 197 blocks of 256
columns instead of 4 parts; fp32, |lse| ~ 11)."""
import ctypes as C
import os

import pytest
import torch

from gpu_util import DEV, bf, gemm, stream
from kmbart import _lib
from kmbart._lib import KmbGemm, check, ptr

pytestmark = pytest.mark.gpu

ROWS, COLS = 320, 256   # the statistics' row stride and the columns per block


def _gemm_stats(A, W, V, bias, out):
    lib = _lib.load()
    g = KmbGemm()
    g.A, g.B = ptr(A), ptr(W)
    g.lda, g.ldb = A.stride(0), W.stride(0)
    g.a_kc, g.b_kc = 1, 1
    g.M, g.N, g.K = A.shape[0], V, A.shape[1]
    g.bias = ptr(bias)
    g.col_scale = 1.0
    g.drop_scale = 1.0
    g.out_f32, g.ld_out_f32 = ptr(out), out.stride(0)
    stats = torch.full((int(lib.kmb_op_gemm_allrows_stats_floats(V)),), float("nan"), device=DEV)
    check(lib.kmb_op_gemm_allrows_stats(C.byref(g), ptr(stats), stream()))
    nblk = (V + COLS - 1) // COLS
    assert stats.numel() == nblk * 2 * ROWS
    return stats, nblk


def _problem(R, V, seed, scale=0.05, zero_rows=(), const_bias=None):
    g = torch.Generator(device=DEV).manual_seed(seed)
    ld = (V + 127) // 128 * 128
    A = bf(torch.randn(R, 768, device=DEV, generator=g) * 0.5)
    for r in zero_rows:
        A[r] = 0
    W = bf(torch.randn(ld, 768, device=DEV, generator=g) * scale)
    bias = torch.randn(V, device=DEV, generator=g) if const_bias is None else torch.full((V,), const_bias, device=DEV)
    return A, W, bias, ld


@pytest.mark.parametrize("R,V", [(320, 50320), (300, 50320), (257, 1000), (16, 50320)])
def test_statistics_epilogue_logits_maxima_sums(R, V):
    A, W, bias, ld = _problem(R, V, seed=R + V)
    want = torch.zeros((R, ld), dtype=torch.float32, device=DEV)
    gemm(A, W, N=V, bias=bias, out_f32=want, allrows=True)
    got = torch.zeros((R, ld), dtype=torch.float32, device=DEV)
    stats, nblk = _gemm_stats(A, W, V, bias, got)
    assert torch.equal(got, want)                       # the logits are the plain kernel's, bit for bit (pad columns untouched)
    st = stats.view(ROWS, nblk, 2).permute(1, 2, 0)      # -> [block, max / sum, row]
    x = want[:, :V].double()
    xp = torch.full((R, nblk * COLS), float("-inf"), dtype=torch.float64, device=DEV)
    xp[:, :V] = x
    xb = xp.view(R, nblk, COLS)
    m = xb.max(dim=2).values                            # [R, nblk]
    assert torch.equal(st[:, 0, :R].t().double(), m)    # maxima: exact
    s = torch.exp(xb - m[:, :, None]).sum(dim=2)
    rel = ((st[:, 1, :R].t().double() - s).abs() / s).max()
    assert float(rel) < 1e-5, float(rel)
    lse = torch.logsumexp(x, dim=1)
    mine = torch.logsumexp(st[:, 0, :R].t().double() + torch.log(st[:, 1, :R].t().double()), dim=1)
    assert float((mine - lse).abs().max()) < 2e-6


def _both_steps(logits, ld, V, B, nb, k, add, force, ban, eos, stats, nblk):
    lib = _lib.load()
    R = B * nb
    outs = []
    for use_stats in (False, True):
        cand = torch.zeros((B, k, 2), dtype=torch.int32, device=DEV)
        ns = torch.zeros((R,), dtype=torch.float32, device=DEV)
        nt = torch.zeros((R,), dtype=torch.int64, device=DEV)
        ni = torch.zeros((R,), dtype=torch.int32, device=DEV)
        if use_stats:
            check(lib.kmb_beam_step_stats(ptr(logits), ld, V, B, nb, ptr(add), force, ban, k, ptr(cand), eos, ptr(ns), ptr(nt), ptr(ni),
                                          ptr(stats), nblk, stream()))
        else:
            scr = torch.empty(int(lib.kmb_logsoftmax_topk_scratch(R)), device=DEV)
            check(lib.kmb_beam_step(ptr(logits), ld, V, B, nb, ptr(add), force, ban, k, ptr(cand), eos, ptr(ns), ptr(nt), ptr(ni),
                                    ptr(scr), scr.numel(), stream()))
        torch.cuda.synchronize()
        outs.append((cand.cpu(), ns.cpu(), nt.cpu(), ni.cpu()))
    return outs


def _assert_same_step(a, b, tol=4e-6):
    (ca, nsa, nta, nia), (cb, nsb, ntb, nib) = a, b
    assert torch.equal(ca[:, :, 1], cb[:, :, 1])                                 # beam * V + token of every candidate, in order
    sa, sb = ca[:, :, 0].contiguous().view(torch.float32), cb[:, :, 0].contiguous().view(torch.float32)
    fin = torch.isfinite(sa)
    assert torch.equal(fin, torch.isfinite(sb))
    assert float((sa[fin] - sb[fin]).abs().max()) <= tol
    assert torch.equal(nta, ntb) and torch.equal(nia, nib)
    assert float((nsa - nsb).abs().max()) <= tol


@pytest.mark.parametrize("B,nb,k,V", [(64, 5, 10, 50320), (60, 5, 10, 50320), (40, 8, 16, 50320), (320, 1, 2, 50320), (64, 5, 10, 1000),
                                       (64, 5, 10, 3000)])
def test_beam_step_from_statistics_equals_the_two_launch_step(B, nb, k, V):
    """free steps, min_length steps (the banned token is each row's BEST token for half the rows) and a forced step"""
    R = B * nb
    A, W, bias, ld = _problem(R, V, seed=B * 7 + k)
    logits = torch.zeros((R, ld), dtype=torch.float32, device=DEV)
    stats, nblk = _gemm_stats(A, W, V, bias, logits)
    g = torch.Generator(device=DEV).manual_seed(5)
    add = -torch.rand(R, device=DEV, generator=g) * 3.0
    eos = 2
    _assert_same_step(*_both_steps(logits, ld, V, B, nb, k, add, -1, -1, eos, stats, nblk))
    _assert_same_step(*_both_steps(logits, ld, V, B, nb, k, add, -1, eos, eos, stats, nblk))
    top = int(logits[0, :V].argmax())                   # ban row 0's best token: it must vanish from row 0's candidates
    a, b = _both_steps(logits, ld, V, B, nb, k, add, -1, top, eos, stats, nblk)
    _assert_same_step(a, b)
    assert not bool(((b[0][0, :, 1] % V == top) & (b[0][0, :, 1] // V == 0)).any())
    a, b = _both_steps(logits, ld, V, B, nb, k, add, 0, -1, eos, stats, nblk)   # forced BOS: the logits are not read
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[1], b[1])


def test_tied_logits_go_to_the_smaller_index():
    """rows of a zero hidden state: every logit is the (constant) bias -- 50320 equal candidates per row, the slow exact form;
    rows with a handful of exact duplicates of the maximum spread over blocks; and the banned token inside the tie"""
    B, nb, k, V = 64, 5, 10, 50320
    R = B * nb
    A, W, bias, ld = _problem(R, V, seed=3, zero_rows=range(0, R, 3), const_bias=0.25)
    W[4097] = W[17]; W[30001] = W[17]; W[50319] = W[17]; W[255] = W[256]       # duplicated vocabulary rows: exact ties across blocks
    logits = torch.zeros((R, ld), dtype=torch.float32, device=DEV)
    stats, nblk = _gemm_stats(A, W, V, bias, logits)
    add = torch.zeros(R, device=DEV)
    add[::nb] = 0.0
    for ban in (-1, 0, 3, 17):
        a, b = _both_steps(logits, ld, V, B, nb, k, add, -1, ban, 2, stats, nblk)
        _assert_same_step(a, b)
    a, b = _both_steps(logits, ld, V, B, nb, k, add, -1, -1, 2, stats, nblk)
    tok0 = (b[0][0, :, 1] % V).tolist()                 # item 0: beam 0 is a zero row, all beams' add are 0
    assert tok0[:2] == [0, 1] or b[0][0, 0, 1] // V != 0


@pytest.mark.parametrize("knob,exact", [("KMB_GEN_HEAD_STATS", False), ("KMB_GEN_FOLD_EMBED", True), ("KMB_GEN_HIST", True)])
def test_generation_with_and_without_the_statistics_path_returns_the_same_ids(knob, exact):
    """model.generate at the benchmarked shape's row count (64 items x 5 beams = 320 rows: the all-rows kernel and its statistics run)
    with KMB_GEN_HEAD_STATS=0 (two-launch beam step over the logits) and by default: same ids; sequence scores within 1e-5.
    KMB_GEN_FOLD_EMBED=0 (the next step's embedding by kmb_gen_step's own launch instead of the beam step's) and KMB_GEN_HIST=0 (physical
    cache reorder by kmb_gen_reorder instead of the history gather folded into the beam step): same ids AND bit-identical scores."""
    from oracle import goldenlib as G
    from oracle import kmbart_oracle as O
    from src.data.synthetic import make_batch
    from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration
    from test_fullsize_parity_gpu import BASE
    ocfg = O.OracleConfig.from_dict(BASE)
    sd = G.golden_state_dict(ocfg, seed=11)
    sd["model.shared.weight"] = sd["model.shared.weight"] * 8.0
    sd["model.decoder.embed_positions.weight"] = sd["model.decoder.embed_positions.weight"] * 40.0
    for k_ in list(sd):
        if k_.endswith("out_proj.weight") or k_.endswith("fc2.weight"):
            sd[k_] = sd[k_] * 3.0
    model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(BASE))
    model.load_state_dict(sd, strict=False)
    model.to(DEV).eval()
    b = make_batch(64, seed=77, regions=tuple(36 if i % 3 else 9 for i in range(64)), event_lens=tuple(7 + (i * 5) % 17 for i in range(64)),
                   label_lens=(32,) * 64)
    kw = dict(input_ids=b["input_ids"].to(DEV), image_features=[f.to(DEV) for f in b["image_features"]],
              attention_mask=b["attention_mask"].to(DEV), num_beams=5, max_length=12, early_stopping=True)
    got, got_sc = model.generate(return_scores=True, **kw)
    os.environ[knob] = "0"
    try:
        want, want_sc = model.generate(return_scores=True, **kw)
    finally:
        os.environ.pop(knob, None)
    assert torch.equal(got, want)
    assert float((got_sc.float() - want_sc.float()).abs().max()) < 1e-5
    if exact:
        assert torch.equal(got_sc, want_sc)
    assert len({tuple(r) for r in want.tolist()}) > 8


def _drive(eng, b, nb, steps, folded, fresh_token_buffer=False):
    """gen_step + beam_step for `steps` steps through the engine API; folded: the beam step also reorders (and embeds the next step's tokens),
    otherwise gen_reorder is called separately (and gen_step embeds).  Returns the logits of every step and the chosen tokens."""
    V = eng.config.vocab_size
    B = b["input_ids"].shape[0]
    R = B * nb
    k = max(2, 2 * nb)
    eng.gen_begin(b["input_ids"].to(DEV), [f.to(DEV) for f in b["image_features"]], b["attention_mask"].to(DEV), nb, steps + 2)
    tok = torch.full((R,), 2, dtype=torch.int64, device=DEV)
    add = torch.zeros(R, device=DEV)
    out = []
    for t in range(steps):
        lg = eng.gen_step(tok, t)
        out.append(lg[:, :V].clone())
        cand, add, ntok, nidx = eng.beam_step(lg, nb, k, add, eos_token=-1, reorder_step=t if folded else -1)
        if not folded:
            eng.gen_reorder(nidx, t)
        tok = ntok.clone() if fresh_token_buffer else ntok    # a clone: another buffer -- the prepared embedding must not be used blindly
        out.append(ntok.clone())
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("B,nb", [(64, 5), (300, 1), (12, 4)])
def test_folded_reorder_and_embedding_equal_the_separate_calls(B, nb):
    """kmb_gen_beam_step(reorder_step = t) == kmb_gen_beam_step(-1) + kmb_gen_reorder(t) + kmb_gen_step's own embedding, bit for bit over
    four decode steps: 64 x 5 (statistics path), 300 x 1 (one beam per item: the row -> item table is gathered too), 12 x 4 (48 rows: the
    two-launch beam step carries the fold); and a token buffer that is NOT the one the beam step embedded is embedded again."""
    from oracle import goldenlib as G
    from oracle import kmbart_oracle as O
    from src.data.synthetic import make_batch
    from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration
    from test_fullsize_parity_gpu import BASE
    ocfg = O.OracleConfig.from_dict(BASE)
    sd = G.golden_state_dict(ocfg, seed=11)
    sd["model.shared.weight"] = sd["model.shared.weight"] * 8.0
    sd["model.decoder.embed_positions.weight"] = sd["model.decoder.embed_positions.weight"] * 40.0
    model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(BASE))
    model.load_state_dict(sd, strict=False)
    model.to(DEV).eval()
    b = make_batch(B, seed=B + nb)
    eng = model._engine
    want = _drive(eng, b, nb, 4, folded=False)
    for fresh in (False, True):
        got = _drive(eng, b, nb, 4, folded=True, fresh_token_buffer=fresh)
        assert len(got) == len(want)
        for i, (g_, w_) in enumerate(zip(got, want)):
            assert torch.equal(g_, w_), (fresh, i)
    assert len({int(t) for t in want[-1].tolist()}) > 1


def test_rows_of_nans_still_hand_on_real_columns():
    """a diverged model: hidden rows of NaN -> rows of NaN logits (block maxima skip NaNs, so the fast form finds no candidate there).  The
    tokens handed to the next step are gathered from the embedding table by index: they must be real columns (< V) whatever the scores
    are, as with the two-launch path; items without such rows are unaffected."""
    B, nb, k, V = 64, 5, 10, 50320
    R = B * nb
    A, W, bias, ld = _problem(R, V, seed=9)
    A[0:5] = float("nan")        # all five beams of item 0
    A[7] = float("nan")          # one beam of item 1
    logits = torch.zeros((R, ld), dtype=torch.float32, device=DEV)
    stats, nblk = _gemm_stats(A, W, V, bias, logits)
    assert bool(torch.isnan(logits[0, :V]).all())
    add = torch.zeros(R, device=DEV)
    a, b = _both_steps(logits, ld, V, B, nb, k, add, -1, 2, 2, stats, nblk)
    for (cand, ns, nt, ni) in (a, b):
        tok = cand[:, :, 1] % V
        assert int(nt.min()) >= 0 and int(nt.max()) < V
        assert int(tok.min()) >= 0 and int(tok.max()) < V and int((cand[:, :, 1] // V).max()) < nb
        assert int(ni.min()) >= 0 and int(ni.max()) < R
    assert torch.equal(a[0][2:, :, 1], b[0][2:, :, 1]) and torch.equal(a[2][10:], b[2][10:]) and torch.equal(a[3][10:], b[3][10:])
