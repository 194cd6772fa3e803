"""GPU: the remaining corners of the reference's Python surface on the engine.

  * MultiModalBartModel.forward -> (decoder states, encoder states) (reference src/model/model.py:39-103);
  * forward(encoder_outputs=...) reuses a precomputed encoder (model.py:76-83);
  * outputs[1] of a training forward = the actual training logits, produced lazily from the saved decoder states;
  * beam search with min_length (EOS banned AFTER log_softmax) and beam-search multinomial sampling
    (src/model/mixins.py:336-361 -> transformers 3.0.2 _generate_beam_search) against the oracle with a shared sampler;
  * AdamW: one bias-correction step per step() with several parameter groups; torch-format optimizer state loads."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import goldenlib as G  # noqa: E402
from oracle import kmbart_oracle as O  # noqa: E402
from oracle.make_golden import tiny_batch  # noqa: E402
from kmbart.optim import AdamW  # noqa: E402
from src.model import MultiModalBartModel  # noqa: E402
from test_model_gpu import ACT_TOL, DEV, build, cfg_from_oracle, rel, run_fwd  # noqa: E402


def dev_batch(b):
    return dict(input_ids=b["input_ids"].to(DEV), image_features=[f.to(DEV) for f in b["image_features"]],
                attention_mask=b["attention_mask"].to(DEV))


def test_bare_model_returns_decoder_and_encoder_states():
    ocfg = G.tiny_config()
    sd = G.golden_state_dict(ocfg, seed=17)
    b = tiny_batch(seed=61)
    with torch.no_grad():
        enc = O.encoder_forward(sd, ocfg, b["input_ids"], b["image_features"], b["attention_mask"])
        pm, causal = O.prepare_decoder_masks(ocfg, b["decoder_input_ids"], b["decoder_attention_mask"])
        hdec, _ = O.decoder_forward(sd, ocfg, b["decoder_input_ids"], enc, b["attention_mask"], pm, causal)
    model = MultiModalBartModel(cfg_from_oracle(ocfg))
    model.load_state_dict({k: v for k, v in sd.items()}, strict=False)
    model.to(DEV).eval()
    out = model(decoder_input_ids=b["decoder_input_ids"].to(DEV), use_cache=False,
                decoder_attention_mask=b["decoder_attention_mask"].to(DEV), **dev_batch(b))
    assert isinstance(out, tuple) and len(out) == 2
    dm, am = b["decoder_attention_mask"].bool(), b["attention_mask"].bool()
    assert out[0].shape == hdec.shape and out[1].shape == enc.shape
    assert rel(out[0].cpu()[dm], hdec[dm]) < ACT_TOL and rel(out[1].cpu()[am], enc[am]) < ACT_TOL
    # bare-model checkpoints carry no `model.` prefix and no logits bias
    keys = model.state_dict().keys()
    assert "shared.weight" in keys and "final_logits_bias" not in keys and not any(k.startswith("model.") for k in keys)
    # the second call reuses the encoder output (model.py:76-83): identical decoder states
    again = model(decoder_input_ids=b["decoder_input_ids"].to(DEV), encoder_outputs=(out[1],), use_cache=False,
                  decoder_attention_mask=b["decoder_attention_mask"].to(DEV), **dev_batch(b))
    assert torch.equal(again[0], out[0])


def test_forward_with_encoder_outputs_scores_like_the_full_forward():
    ocfg = G.tiny_config()
    sd = G.golden_state_dict(ocfg, seed=19)
    b = tiny_batch(seed=63)
    model = build(ocfg, sd).eval()
    with torch.no_grad():
        loss, logits, enc = run_fwd(model, b, return_logits=True)
        loss2, logits2, enc2 = run_fwd(model, b, return_logits=True, encoder_outputs=(enc, [], []))
    assert torch.equal(logits, logits2) and float(loss) == float(loss2) and torch.equal(enc, enc2)


def test_forward_with_encoder_outputs_is_differentiable():
    """Reference src/model/model.py:76-83 with a tensor that requires grad: backward stops at the given states and hands their
    gradient on; the decoder's parameter gradients are those of the oracle at the same states, the encoder's are zero."""
    from oracle import kmbart_oracle as O
    ocfg = G.tiny_config(dropout=0.0)
    sd = G.golden_state_dict(ocfg, seed=29)
    b = tiny_batch(seed=67)
    model = build(ocfg, sd).train()
    with torch.no_grad():
        enc = run_fwd(model.eval(), b, return_logits=True)[2]
    model.train()
    enc_in = enc.detach().clone().requires_grad_(True)
    out = run_fwd(model, b, encoder_outputs=(enc_in,))
    assert out[0].requires_grad
    model.zero_grad()
    out[0].backward()
    torch.cuda.synchronize()
    got_enc = enc_in.grad.float().cpu()
    got = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()}
    # the oracle at the same (bf16-rounded) encoder states
    osd = {k: v.clone().float().requires_grad_(k != "final_logits_bias") for k, v in sd.items()}
    e0 = enc.float().cpu().clone().requires_grad_(True)
    loss, _, _ = O.forward(osd, ocfg, b["input_ids"], b["image_features"], b["attention_mask"], b["decoder_input_ids"],
                           b["decoder_attention_mask"], b["labels"], training=False, encoder_out=e0)
    loss.backward()
    assert abs(float(out[0].detach()) - float(loss.detach())) <= 2e-3 * abs(float(loss.detach()))
    am = b["attention_mask"].bool()
    assert rel(got_enc[am], e0.grad[am]) < 3e-2, rel(got_enc[am], e0.grad[am])
    worst = 0.0
    for n, g in got.items():
        ref = osd[n].grad
        if n.startswith("model.encoder."):
            assert float(g.abs().max()) == 0.0, n            # the encoder took no part in this loss
            assert ref is None or float(ref.abs().max()) == 0.0
        elif n.startswith("model.decoder.") or n == "model.shared.weight":
            if float(ref.norm()) < 1e-6:       # k_proj.bias: zero by the softmax's shift invariance
                assert float(g.norm()) < 1e-2, n
                continue
            worst = max(worst, rel(g, ref))
            assert rel(g, ref) < 3e-2, (n, rel(g, ref))
    assert worst > 0.0
    # a second, ordinary step afterwards differentiates the encoder again
    model.zero_grad()
    run_fwd(model, b)[0].backward()
    torch.cuda.synchronize()
    assert float(dict(model.named_parameters())["model.encoder.layers.0.fc1.weight"].grad.abs().max()) > 0.0


def test_lazy_logits_are_the_training_logits_and_expire():
    ocfg = G.tiny_config()
    sd = G.golden_state_dict(ocfg, seed=23)
    b = tiny_batch(seed=65)
    model = build(ocfg, sd, dropout=0.1).train()
    model._engine.set_seed(5)
    eager = run_fwd(model, b, return_logits=True)[1].clone()
    model._engine.set_seed(5)
    out = run_fwd(model, b)
    lazy = out[1]
    out[0].backward()                       # backward leaves the saved decoder states alone
    assert torch.equal(lazy[:, :, :], eager)          # same dropout masks: the ACTUAL training logits
    # the encoder states of a training forward are lazy too (round 6: no [B, S, d] copy per step): read after backward they are
    # the states an eager copy of the same forward (same dropout seed) holds
    from src.model.model import LazyEncoderStates
    assert isinstance(out[2], LazyEncoderStates)
    lazy_enc = out[2][:, :, :].clone()
    model._engine.set_seed(5)
    with torch.no_grad():
        eager_enc = run_fwd(model, b)[2]
    assert torch.is_tensor(eager_enc) and torch.equal(lazy_enc, eager_enc)
    opt = AdamW(model.parameters(), lr=1e-3)
    out2 = run_fwd(model, b)
    out2[0].backward()
    opt.step()
    with pytest.raises(RuntimeError, match="gone"):
        out2[1].shape
    with pytest.raises(RuntimeError, match="gone"):
        out2[2].shape


def test_beam_search_min_length_and_sampling_follow_the_oracle():
    from oracle.make_golden import copy_task_batch
    ocfg = G.tiny_config()
    sd = G.trained_state_dict()                                # trained reverse-copy model: peaked, input-dependent
    model = build(ocfg, sd).eval()
    b = copy_task_batch(7, 4)
    kw = dict(num_beams=3, max_length=10, min_length=7, early_stopping=True, num_return_sequences=2)
    got, sc = model.generate(return_scores=True, **dev_batch(b), **kw)
    ref, rsc = O.generate(sd, ocfg, b["input_ids"], b["image_features"], b["attention_mask"], return_scores=True, **kw)
    assert got.cpu().tolist() == ref.tolist()
    assert torch.allclose(sc, rsc, atol=5e-2)
    assert all((row[1:7] != ocfg.eos_token_id).all() for row in got.cpu())
    without = model.generate(**dev_batch(b), **dict(kw, min_length=0))
    assert without.cpu().tolist() != got.cpu().tolist()        # the ban changed the search

    # beam-search multinomial sampling: the same draws in both implementations (inverse-CDF on a fixed uniform
    # stream), so the whole bookkeeping -- no forced BOS/EOS, all-zero initial beam scores, top-k/top-p with
    # min_tokens_to_keep=2, 2*num_beams draws, sort, batch replication for num_return_sequences -- is compared
    def make_sampler():
        g = torch.Generator().manual_seed(1234)

        def sampler(probs, n):   # without replacement, like torch.multinomial: Gumbel top-n on a fixed noise stream
            p = probs.detach().double().cpu()
            u = torch.rand(p.shape, generator=g, dtype=torch.float64).clamp_(1e-12, 1 - 1e-12)
            keys = torch.log(p) - torch.log(-torch.log(u))
            return torch.topk(keys, n, dim=-1)[1].to(probs.device)
        return sampler

    skw = dict(num_beams=3, max_length=9, do_sample=True, top_k=8, top_p=0.9, temperature=1.3, early_stopping=True,
               num_return_sequences=2)
    model._sampler = make_sampler()
    got = model.generate(**dev_batch(b), **skw)
    ref = O.generate(sd, ocfg, b["input_ids"], b["image_features"], b["attention_mask"], sampler=make_sampler(), **skw)
    model._sampler = None
    assert got.shape[0] == b["input_ids"].shape[0] * 2
    # bf16 logits: a draw that lands within rounding of a CDF step may pick the neighbouring token; demand that most
    # rows agree exactly and every row starts from the decoder start token
    same = sum(a == r for a, r in zip(got.cpu().tolist(), ref.tolist()))
    assert same * 2 >= got.shape[0], (got.cpu().tolist(), ref.tolist())
    assert (got[:, 0] == ocfg.decoder_start_token_id).all()
    # default sampler (torch.multinomial): runs, shapes right, reproducible per seed, and different seeds differ once the
    # distribution is flat enough to leave a choice (the trained model is peaked: top_p = 0.9 keeps one or two tokens)
    hot = dict(skw, temperature=20.0, top_k=0, top_p=1.0)
    torch.manual_seed(1)
    a = model.generate(**dev_batch(b), **hot)
    torch.manual_seed(2)
    c = model.generate(**dev_batch(b), **hot)
    torch.manual_seed(1)
    a2 = model.generate(**dev_batch(b), **hot)
    assert a.shape[0] == c.shape[0] == 2 * b["input_ids"].shape[0] and torch.equal(a, a2)
    assert a.shape != c.shape or not torch.equal(a, c)


def test_adamw_param_groups_and_torch_state_dict():
    ocfg = G.tiny_config()
    sd = G.golden_state_dict(ocfg, seed=29)
    b = tiny_batch(seed=69)

    def step_with(groups):
        model = build(ocfg, sd).train()
        ps = list(model.parameters())
        opt = AdamW([{"params": ps[:40]}, {"params": ps[40:]}] if groups == 2 else ps, lr=1e-3)
        for _ in range(2):
            run_fwd(model, b)[0].backward()
            opt.step()
        torch.cuda.synchronize()
        return model._engine.params.clone(), model._engine.step_count

    p1, n1 = step_with(1)
    p2, n2 = step_with(2)
    assert n1 == n2 == 2          # one bias-correction step per optimizer.step(), whatever the number of groups
    off, rows, cols = build(ocfg, sd)._engine.index["model.shared.weight"]
    mask = torch.ones_like(p1, dtype=torch.bool)
    mask[off: off + rows * cols] = False
    assert torch.equal(p1[mask], p2[mask])
    # a torch-format state ({state: {i: {step, exp_avg, exp_avg_sq}}, param_groups}), as the reference's
    # training_data.pt holds (src/utils.py:20-39), loads into the arena
    model = build(ocfg, sd).train()
    params = list(model.parameters())
    opt = AdamW(params, lr=1e-3)
    g = torch.Generator().manual_seed(0)
    # ... whose indices follow the REFERENCE model's parameters() order (shared first, k/v/q interleaved with their biases,
    # layernorm_embedding after the layers: tests/test_host_logic_cpu.py pins that order on transformers' BART), not
    # this engine's arena order
    from kmbart.optim import reference_parameter_order
    by_name = dict(model.named_parameters())
    order = reference_parameter_order(list(by_name))
    assert order[0] == "model.shared.weight" and order != [n for n, _ in model.named_parameters()]
    state = {"state": {i: {"step": 7, "exp_avg": torch.randn(by_name[n].shape, generator=g),
                           "exp_avg_sq": torch.rand(by_name[n].shape, generator=g)} for i, n in enumerate(order)},
             "param_groups": [{"lr": 5e-4, "betas": (0.9, 0.999), "eps": 1e-6, "weight_decay": 0.0,
                               "correct_bias": True, "params": list(range(len(order)))}]}
    opt.load_state_dict(state)
    eng = model._engine
    assert eng.step_count == 7 and opt.param_groups[0]["lr"] == 5e-4
    for i, n in enumerate(order):
        o, cnt = by_name[n]._kmb_range
        assert torch.equal(eng.exp_avg[o: o + cnt].cpu(), state["state"][i]["exp_avg"].reshape(-1)), n
        assert torch.equal(eng.exp_avg_sq[o: o + cnt].cpu(), state["state"][i]["exp_avg_sq"].reshape(-1)), n
    with pytest.raises(ValueError):
        opt.load_state_dict({"foo": 1})
    # the optimizer's OWN format records the arena layout: it round-trips, a state saved under a DIFFERENT arena
    # order is remapped by name, and a state without a layout is refused instead of being copied blindly
    own = opt.state_dict()
    assert set(own["layout"][0]) == set(eng.index)
    m_before, v_before = eng.exp_avg.clone(), eng.exp_avg_sq.clone()
    eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
    opt.load_state_dict(own)
    assert torch.equal(eng.exp_avg, m_before) and torch.equal(eng.exp_avg_sq, v_before)
    names = sorted(eng.index)                     # "another build": the same parameters packed in name order
    perm_layout, cur = {}, 0
    for n in names:
        cnt = eng.index[n][1] * eng.index[n][2]
        perm_layout[n] = (cur, cnt)
        cur += cnt + 3                            # and with different alignment gaps
    pm, pv = torch.zeros(cur), torch.zeros(cur)
    for n in names:
        so, cnt = perm_layout[n]
        o = eng.index[n][0]
        pm[so: so + cnt] = m_before[o: o + cnt].cpu()
        pv[so: so + cnt] = v_before[o: o + cnt].cpu()
    other = dict(own, exp_avg=[pm], exp_avg_sq=[pv], layout=[perm_layout])
    eng.exp_avg.fill_(9.0); eng.exp_avg_sq.fill_(9.0)
    opt.load_state_dict(other)
    for n in names:
        o, cnt = eng.index[n][0], eng.index[n][1] * eng.index[n][2]
        assert torch.equal(eng.exp_avg[o: o + cnt], m_before[o: o + cnt]), n
        assert torch.equal(eng.exp_avg_sq[o: o + cnt], v_before[o: o + cnt]), n
    legacy = {k: v for k, v in own.items() if k != "layout"}
    with pytest.raises(ValueError):
        opt.load_state_dict(legacy)


def test_generate_text_on_the_hip_path_equals_the_reference_records(gold_dir):
    """The records the REFERENCE's generate_text returned over the oracle (tests/golden/generate_text_reference.json,
    oracle/make_golden_reference_api.py) against the product's generate_text over the HIP model: same ids, same text."""
    import json
    import os
    import types
    from oracle.make_golden_reference_api import gen_loader
    from src.generation import generate_text
    ref = json.load(open(os.path.join(gold_dir, "generate_text_reference.json")))
    ocfg = G.tiny_config()
    model = build(ocfg, G.trained_state_dict()).eval()
    model.config.max_length = 12            # the fixture's adapter generated with max_length 12
    for case in ref["cases"]:
        recs = generate_text(model, gen_loader(), G.IdTokenizer(), types.SimpleNamespace(amp=False, **case["args"]),
                             torch.device(DEV), logger=types.SimpleNamespace(info=lambda m: None))
        assert recs == case["records"], (case["args"], recs[:2], case["records"][:2])


def test_cached_forward_returns_logits_cache_and_encoder_states():
    """forward(use_cache=True, decoder_cached_states=...) (reference src/model/model.py:384-397, mixins.py:386-434): the
    last position's logits, a cache handle and the encoder states; step by step it reproduces the teacher-forced logits of
    the same model (and through them the oracle's), _reorder_cache moves whole sequences, and a hand-written greedy loop
    over get_encoder() / prepare_inputs_for_generation returns what generate() returns."""
    from oracle.make_golden import copy_task_batch
    from src.model.model import DecoderCache
    ocfg = G.tiny_config()
    sd = G.trained_state_dict()
    model = build(ocfg, sd).eval()
    b = copy_task_batch(13, 4)
    kw = dev_batch(b)
    g = torch.Generator().manual_seed(3)
    dec = torch.randint(3, 400, (4, 8), generator=g)      # no pads: the cached step has no decoder padding mask (HF 3.0.2)
    dec[:, 0] = 0
    dec = dec.to(DEV)
    V = ocfg.vocab_size
    with torch.no_grad():
        full_logits, enc_full = model(decoder_input_ids=dec, use_cache=False, **kw)[:2]
        ref_logits = O.forward(sd, ocfg, b["input_ids"], b["image_features"], b["attention_mask"], dec.cpu(), None, None)[1]
    assert rel(full_logits, ref_logits) < ACT_TOL
    # (a) one call with several columns = prefill: logits of the last position, the cache, the encoder states
    out = model(decoder_input_ids=dec[:, :7], use_cache=True, **kw)
    assert len(out) == 3 and isinstance(out[1], DecoderCache) and tuple(out[0].shape) == (4, 1, V)
    assert out[1].length == 7 and tuple(out[2].shape) == tuple(enc_full.shape)
    assert rel(out[0][:, 0], full_logits[:, 6]) < 1e-2 and rel(out[2], enc_full) < 1e-2
    # (b) column by column through the returned cache
    cache = None
    for t in range(1, 8):
        first = dict(kw) if cache is None else {"input_ids": None, "image_features": None}
        step = model(decoder_input_ids=dec[:, :t], use_cache=True, decoder_cached_states=cache, **first)
        cache = step[1]
        assert rel(step[0][:, 0], full_logits[:, t - 1]) < 1e-2, t
    # (c) _reorder_cache with a permutation ACROSS sequences: every row keeps decoding its own sequence
    perm = torch.tensor([2, 0, 3, 1], device=DEV)
    (enc_p, mask_p), cache = model._reorder_cache(((step[2], kw["attention_mask"]), cache), perm)
    assert torch.equal(enc_p, step[2][perm]) and torch.equal(mask_p, kw["attention_mask"][perm])
    nxt = model(input_ids=None, image_features=None, decoder_input_ids=dec[perm][:, :8], use_cache=True, decoder_cached_states=cache)
    assert rel(nxt[0][:, 0], full_logits[perm][:, 7]) < 1e-2
    # a stale cache is refused (another forward re-used the workspace); labels switch the cache off (model.py:381-382)
    model(decoder_input_ids=dec, use_cache=False, **kw)
    with pytest.raises(RuntimeError):
        model(input_ids=None, image_features=None, decoder_input_ids=dec, use_cache=True, decoder_cached_states=cache)
    three = model(decoder_input_ids=dec, labels=dec, use_cache=True, **kw)
    assert three[0].dim() == 0 and tuple(three[1].shape) == (4, 8, V)
    # (d) the reference's generation plumbing by hand: greedy search over the cached forward == generate(num_beams=1)
    want = model.generate(max_length=12, num_beams=1, **kw)
    enc_out = model.get_encoder()(kw["input_ids"], kw["image_features"], kw["attention_mask"])
    ids = torch.full((4, 1), model.config.decoder_start_token_id, dtype=torch.long, device=DEV)
    unfinished = torch.ones(4, dtype=torch.long, device=DEV)
    cache = None
    while ids.shape[1] < 12:
        inp = model.prepare_inputs_for_generation(ids, past=(enc_out, cache), attention_mask=kw["attention_mask"], use_cache=True)
        logits, cache = model(**inp)[:2]
        tok = logits[:, -1].argmax(-1) * unfinished + model.config.pad_token_id * (1 - unfinished)
        ids = torch.cat([ids, tok[:, None]], dim=1)
        unfinished = unfinished * (tok != model.config.eos_token_id).long()
        if int(unfinished.max()) == 0:
            break
    assert ids.cpu().tolist() == want.cpu().tolist()
    # (e) use_cache=None falls back to config.use_cache (True) when there are no labels (reference src/model/model.py:59):
    # an eval forward returns the last position and the cache, exactly what use_cache=True returns
    dflt = model(decoder_input_ids=dec[:, :7], **kw)
    assert len(dflt) == 3 and isinstance(dflt[1], DecoderCache) and tuple(dflt[0].shape) == (4, 1, V)
    assert torch.equal(dflt[0], model(decoder_input_ids=dec[:, :7], use_cache=True, **kw)[0])
    model.config.use_cache = False
    assert tuple(model(decoder_input_ids=dec[:, :7], **kw)[0].shape) == (4, 7, V)
    model.config.use_cache = True
    # (f) the reference's BEAM plumbing (mixins.py:281-324): the encoder output is expanded num_beams-fold with index_select
    # (a new tensor: nothing rides on it) before the first cached step, which gets input_ids=None
    k = 3
    enc_out = model.get_encoder()(kw["input_ids"], kw["image_features"], kw["attention_mask"])
    expand = torch.arange(4, device=DEV).repeat_interleave(k)
    enc_exp = (enc_out[0].index_select(0, expand),)
    mask_exp = kw["attention_mask"].index_select(0, expand)
    start = torch.full((4 * k, 1), model.config.decoder_start_token_id, dtype=torch.long, device=DEV)
    inp = model.prepare_inputs_for_generation(start, past=(enc_exp, None), attention_mask=mask_exp, use_cache=True)
    lg, cache = model(**inp)[:2]
    assert tuple(lg.shape) == (4 * k, 1, V) and cache.rows == 4 * k
    one = model(decoder_input_ids=start[:4], use_cache=True, **kw)[0]
    assert rel(lg[::k], one) < 1e-3 and rel(lg[1::k], one) < 1e-3     # every copy of an item decodes that item
    # (g) ADVICE r5: encoder states expanded in ANOTHER order (or edited) are refused, not silently replaced by a recomputed encoder;
    # input_ids passed together with encoder_outputs: the given encoder_outputs win (reference src/model/model.py:76-83) when they
    # are this model's pending ones, and raise when they are not
    enc_out = model.get_encoder()(kw["input_ids"], kw["image_features"], kw["attention_mask"])
    wrong = (enc_out[0].repeat(k, 1, 1),)                            # batch-major instead of item-major
    with pytest.raises(ValueError, match="not the states"):
        model(input_ids=None, image_features=None, encoder_outputs=wrong, decoder_input_ids=start, use_cache=True)
    enc_out = model.get_encoder()(kw["input_ids"], kw["image_features"], kw["attention_mask"])
    edited = (enc_out[0] * 2,)
    with pytest.raises(ValueError, match="not the states"):
        model(input_ids=None, image_features=None, encoder_outputs=edited, decoder_input_ids=start[:4], use_cache=True)
    enc_out = model.get_encoder()(kw["input_ids"], kw["image_features"], kw["attention_mask"])
    both = model(encoder_outputs=enc_out, decoder_input_ids=start[:4], use_cache=True, **kw)
    assert torch.equal(both[0], one)
    with pytest.raises(ValueError, match="encoder_outputs"):          # no pending get_encoder() call any more
        model(encoder_outputs=enc_out, decoder_input_ids=start[:4], use_cache=True, **kw)


def test_bare_model_cached_step_and_taps():
    """MultiModalBartModel.forward(use_cache / decoder_cached_states / output_*) (reference src/model/model.py:39-103): the
    cached step returns the new position's decoder states + cache + encoder states; the hidden-state / attention taps of the
    teacher-forced forward sit where the reference's filtered tuple puts them."""
    from oracle.make_golden import copy_task_batch
    from src.model.model import DecoderCache
    ocfg = G.tiny_config()
    sd = G.trained_state_dict()
    model = MultiModalBartModel(cfg_from_oracle(ocfg))
    model.load_state_dict({k: v for k, v in sd.items()}, strict=False)
    model.to(DEV).eval()
    b = copy_task_batch(13, 4)
    kw = dev_batch(b)
    g = torch.Generator().manual_seed(5)
    dec = torch.randint(3, 400, (4, 6), generator=g)
    dec[:, 0] = 0
    dec = dec.to(DEV)
    with torch.no_grad():
        full = model(decoder_input_ids=dec, use_cache=False, **kw)
        assert len(full) == 2
        cache = None
        for t in range(1, 7):
            first = dict(kw) if cache is None else {"input_ids": None, "image_features": None}
            step = model(decoder_input_ids=dec[:, :t], decoder_cached_states=cache, **first)   # use_cache defaults to True
            assert len(step) == 3 and isinstance(step[1], DecoderCache) and tuple(step[0].shape) == (4, 1, ocfg.d_model)
            cache = step[1]
            assert rel(step[0][:, 0], full[0][:, t - 1]) < 1e-2, t
        assert rel(step[2], full[1]) < 1e-2
        taps = model(decoder_input_ids=dec, use_cache=False, output_hidden_states=True, output_attentions=True, **kw)
    dec_states, dec_hidden, dec_attn, enc, enc_hidden, enc_attn = taps
    assert torch.equal(dec_states, full[0]) and torch.equal(enc, full[1])
    assert len(dec_hidden) == ocfg.decoder_layers and len(dec_attn) == ocfg.decoder_layers
    assert len(enc_hidden) == ocfg.encoder_layers + 1 and len(enc_attn) == ocfg.encoder_layers
    assert torch.equal(enc_hidden[-1], enc)


def test_pretraining_forward_passes_encoder_outputs_and_taps_through():
    """MultiModalBartForPreTraining.forward(encoder_outputs=..., output_*) (reference src/model/model.py:225-242 hands them to
    self.model; :291 / :309 put the model's remaining outputs behind the logits)."""
    from src.data.synthetic import make_pretrain_batch
    from src.model import MultiModalBartForPreTraining
    ocfg = G.tiny_config(num_labels=37, num_attributes=11, num_relations=9)
    sd = G.golden_state_dict(ocfg, seed=33)
    b = make_pretrain_batch(3, enc_len=24, dec_len=16, num_regions=6, seed=77, num_labels=37, num_attributes=11,
                            num_relations=9, vocab_hi=G.TINY_SPECIAL_BASE, img_feat_id=ocfg.img_feat_id,
                            special_base=G.TINY_SPECIAL_BASE, cls_id=ocfg.cls_token_id, mrm_probability=0.3)
    b["image_features"] = G.golden_features([6, 6, 6])
    cfg = cfg_from_oracle(ocfg, num_labels=37, num_attributes=11, num_relations=9)
    torch.manual_seed(3)
    model = MultiModalBartForPreTraining(cfg)
    model.load_state_dict(sd, strict=False)
    model.to(DEV).eval()
    kw = dict(input_ids=b["input_ids"].to(DEV), image_features=[f.to(DEV) for f in b["image_features"]],
              attention_mask=b["attention_mask"].to(DEV), decoder_input_ids=b["decoder_input_ids"].to(DEV),
              decoder_attention_mask=b["decoder_attention_mask"].to(DEV), labels=b["labels"].to(DEV),
              mrm_labels=b["mrm_labels"], mrm_mask=b["mrm_mask"], attribute_labels=b["attribute_labels"],
              attribute_mask=b["attribute_mask"], relation_labels=b["relation_labels"])
    with torch.no_grad():
        base = model(**kw)
        assert len(base) == 3 and isinstance(base[0], dict)            # (losses, logits, encoder states)
        given = model(encoder_outputs=(base[2],), **kw)
        for k_ in base[0]:
            assert abs(float(given[0][k_]) - float(base[0][k_])) <= 1e-6 * abs(float(base[0][k_])) + 1e-7, k_
        assert torch.equal(given[2], base[2])
        taps = model(output_hidden_states=True, output_attentions=True, **kw)
    losses, logits, dec_hidden, dec_attn, enc, enc_hidden, enc_attn = taps
    assert len(dec_hidden) == ocfg.decoder_layers and len(enc_hidden) == ocfg.encoder_layers + 1
    assert len(dec_attn) == ocfg.decoder_layers and len(enc_attn) == ocfg.encoder_layers
    assert abs(float(losses["loss"]) - float(base[0]["loss"])) <= 1e-6 * abs(float(base[0]["loss"]))
    # without any label it is the conditional-generation forward, cache included (model.py:222-223 only switch it off WITH labels)
    nolab = {k_: v for k_, v in kw.items() if k_ in ("input_ids", "image_features", "attention_mask")}
    with torch.no_grad():
        out = model(decoder_input_ids=kw["decoder_input_ids"][:, :1], **nolab)
    assert len(out) == 3 and tuple(out[0].shape) == (kw["input_ids"].shape[0], 1, ocfg.vocab_size)


def test_embedding_accessors_and_resize():
    """get_output_embeddings (a bias-free Linear over model.shared, reference mixins.py:439-440), get_input_embeddings,
    resize_token_embeddings (mixins.py:442-455: rows kept, final_logits_bias cut / zero-extended; before .to(device))."""
    from src.model import MultiModalBartForConditionalGeneration
    ocfg = G.tiny_config()
    sd = G.golden_state_dict(ocfg, seed=3)
    model = MultiModalBartForConditionalGeneration(cfg_from_oracle(ocfg))
    model.load_state_dict(sd, strict=False)
    old = dict(model.named_parameters())["model.shared.weight"].detach().clone()
    flb_old = model.final_logits_bias.clone()
    emb = model.resize_token_embeddings(ocfg.vocab_size + 16)
    w = dict(model.named_parameters())["model.shared.weight"]
    assert tuple(w.shape) == (ocfg.vocab_size + 16, ocfg.d_model) and emb.weight.shape == w.shape
    assert torch.equal(w[: ocfg.vocab_size], old) and model.config.vocab_size == ocfg.vocab_size + 16
    assert model.final_logits_bias.shape == (1, ocfg.vocab_size + 16)
    assert torch.equal(model.final_logits_bias[:, : ocfg.vocab_size], flb_old) and float(model.final_logits_bias[:, ocfg.vocab_size:].abs().sum()) == 0.0
    model.resize_token_embeddings(ocfg.vocab_size)       # and back: the kept rows are the original ones
    assert torch.equal(dict(model.named_parameters())["model.shared.weight"], old)
    model.to(DEV).eval()
    lin = model.get_output_embeddings()
    wdev = dict(model.named_parameters())["model.shared.weight"]
    assert lin.bias is None and lin.weight.data_ptr() == wdev.data_ptr() and tuple(lin.weight.shape) == tuple(wdev.shape)
    assert model.get_input_embeddings().weight.data_ptr() == wdev.data_ptr()
    with pytest.raises(RuntimeError):
        model.resize_token_embeddings(ocfg.vocab_size + 8)
    # the resized-and-restored model still computes the golden loss
    b = tiny_batch(seed=5)
    loss = run_fwd(model, b)[0]
    ref = O.forward(sd, ocfg, b["input_ids"], b["image_features"], b["attention_mask"], b["decoder_input_ids"],
                    b["decoder_attention_mask"], b["labels"])[0]
    assert abs(float(loss) - float(ref)) / float(ref) < 1e-3


def test_generate_score_processors_follow_the_fixture(gold_dir):
    """repetition_penalty / no_repeat_ngram_size / bad_words_ids in generate (reference src/model/mixins.py:150-235;
    transformers 3.0.2 postprocess_next_token_scores): token ids equal to the fixture the oracle wrote (each case identical
    to transformers 5.15 generate() there), beam search and greedy, and the argument checks of the reference."""
    import json
    import os
    from oracle.make_golden_reference_api import processors_batch
    fx = json.load(open(os.path.join(gold_dir, "tiny_generate_processors.json")))
    ocfg = G.tiny_config()
    model = build(ocfg, G.trained_state_dict()).eval()
    b = processors_batch()
    kwb = dev_batch(b)
    plain = model.generate(num_beams=3, max_length=12, early_stopping=True, **kwb)
    assert plain.cpu().tolist() == fx["plain_ids"]
    for case in fx["cases"]:
        kw = case["kwargs"]
        out = model.generate(return_scores=kw.get("num_beams", 1) > 1, **kw, **kwb)
        ids = out[0] if isinstance(out, tuple) else out
        assert ids.cpu().tolist() == case["ids"], (kw, ids.cpu().tolist(), case["ids"])
        if isinstance(out, tuple):
            assert torch.allclose(out[1].float(), torch.tensor(case["scores"]), atol=3e-2), kw
    with pytest.raises(AssertionError):
        model.generate(repetition_penalty=0.5, **kwb)
    with pytest.raises(AssertionError):
        model.generate(bad_words_ids=[1, 2], **kwb)


def test_output_hidden_states_and_attentions_follow_the_oracle():
    """forward(output_hidden_states=True, output_attentions=True) (reference src/model/modules.py:143-165, transformers
    3.0.2 BartDecoder): (logits, decoder layer inputs, decoder self-attention weights, encoder states, encoder hidden
    states, encoder attention weights) against the oracle's taps; the probabilities are recomputed from the saved q | k
    and log-sum-exp by kmb_attention_probs."""
    ocfg = G.tiny_config()
    sd = G.golden_state_dict(ocfg, seed=17)
    b = tiny_batch(seed=23)                       # ragged: encoder and decoder padding masks
    model = build(ocfg, sd).eval()
    with torch.no_grad():
        out = model(input_ids=b["input_ids"].to(DEV), image_features=[f.to(DEV) for f in b["image_features"]],
                    attention_mask=b["attention_mask"].to(DEV), decoder_input_ids=b["decoder_input_ids"].to(DEV),
                    decoder_attention_mask=b["decoder_attention_mask"].to(DEV), output_hidden_states=True, output_attentions=True)
    logits, dec_hidden, dec_attn, enc, enc_hidden, enc_attn = out
    et, dt = {}, {}
    enc_ref = O.encoder_forward(sd, ocfg, b["input_ids"], b["image_features"], b["attention_mask"], taps=et)
    pm, causal = O.prepare_decoder_masks(ocfg, b["decoder_input_ids"], b["decoder_attention_mask"])
    O.decoder_forward(sd, ocfg, b["decoder_input_ids"], enc_ref, b["attention_mask"], pm, causal, taps=dt)
    assert len(enc_hidden) == ocfg.encoder_layers + 1 and len(enc_attn) == ocfg.encoder_layers
    assert len(dec_hidden) == ocfg.decoder_layers and len(dec_attn) == ocfg.decoder_layers
    am, dm = b["attention_mask"].bool(), b["decoder_attention_mask"].bool()
    for got, ref in zip(enc_hidden, et["hidden"]):
        assert rel(got.cpu()[am], ref[am]) < ACT_TOL
    for got, ref in zip(dec_hidden, dt["hidden"]):
        assert rel(got.cpu()[dm], ref[dm]) < ACT_TOL
    for got, ref in zip(enc_attn, et["attn"]):        # rows of padded queries are unspecified in both; compare valid queries
        g_, r_ = got.cpu().permute(0, 2, 1, 3)[am], ref.permute(0, 2, 1, 3)[am]
        assert float((g_ - r_).abs().max()) < 2e-2 and float((g_.sum(-1) - 1).abs().max()) < 2e-2
    for got, ref in zip(dec_attn, dt["attn"]):
        g_, r_ = got.cpu().permute(0, 2, 1, 3)[dm], ref.permute(0, 2, 1, 3)[dm]
        assert float((g_ - r_).abs().max()) < 2e-2 and float((g_.sum(-1) - 1).abs().max()) < 2e-2
        assert float(torch.triu(got.cpu(), 1).abs().max()) == 0.0      # causal: nothing above the diagonal
    assert rel(enc.cpu()[am], enc_ref[am]) < ACT_TOL
