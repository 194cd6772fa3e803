"""GPU: the fp32 VALIDATION mode (kmb_set_precision, csrc/fp32_validate.hip) against the fp32 CPU oracle.

north_star: "logits/loss within 1e-3 rel fp32 of the reference PyTorch path on identical inputs".  The bf16 product
path meets the loss bound (7e-5) but its logits sit ~1e-2 norm-wise from the oracle: twelve layers of bf16 storage
rounding.  This mode keeps the host orchestration (workspace layout, region row map, positions, masks, head chunking,
CE) and swaps the four bf16 kernel families for exact-fp32 ones; logits AND encoder states must then agree with the
oracle to < 1e-3 (measured ~1e-6), which pins the 1e-2 of the product path on bf16 rounding alone.  Both numbers are
printed side by side.  Round 5: generate() in this mode (an uncached fp32 forward per step) reproduces the oracle's
searches exactly -- tiny trained model and full-size beam-5."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import goldenlib as G  # noqa: E402
from oracle import kmbart_oracle as O  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

DEV = "cuda:0"
FP32_TOL = 1e-3     # norm-wise, logits and encoder states (north_star)
BASE = dict(activation_dropout=0.0, attention_dropout=0.0, d_model=768, decoder_attention_heads=12,
            decoder_ffn_dim=3072, decoder_layers=6, dropout=0.0, encoder_attention_heads=12, encoder_ffn_dim=3072,
            encoder_layers=6, init_std=0.02, max_position_embeddings=1024, vocab_size=50320, cls_token_id=50276,
            img_feat_id=50273)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(model, b):
    return model(input_ids=b["input_ids"].to(DEV), image_features=[f.to(DEV) for f in b["image_features"]],
                 attention_mask=b["attention_mask"].to(DEV), decoder_input_ids=b["decoder_input_ids"].to(DEV),
                 decoder_attention_mask=b["decoder_attention_mask"].to(DEV), labels=b["labels"].to(DEV),
                 return_logits=True)


@pytest.mark.parametrize("case", ["uniform", "ragged"])
def test_vcg_base_b2_logits_within_1e3_in_fp32_mode(case):
    ocfg = O.OracleConfig.from_dict(BASE)
    sd = O.init_state_dict(ocfg, seed=0) if case == "uniform" else G.golden_state_dict(ocfg, seed=5)
    if case == "uniform":   # BASELINE config 1: b = 2, 36 regions, S = 64, T = 32, seed 1234
        b = make_batch(2, seed=1234)
    else:                   # SURVEY 8d ragged variant
        b = make_batch(2, seed=1234, regions=(36, 20), event_lens=(23, 7), label_lens=(32, 19))
    with torch.no_grad():
        ref_loss, ref_logits, ref_enc = O.forward(sd, ocfg, b["input_ids"], b["image_features"], b["attention_mask"],
                                                  b["decoder_input_ids"], b["decoder_attention_mask"], b["labels"])
    model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(BASE))
    model.load_state_dict(sd, strict=False)
    model.to(DEV).eval()
    dm, am = b["decoder_attention_mask"].bool(), b["attention_mask"].bool()
    res = {}
    for mode in ("bf16", "fp32"):
        model._engine.set_precision(mode == "fp32")
        with torch.no_grad():
            loss, logits, enc = run(model, b)
        assert enc.dtype == (torch.float32 if mode == "fp32" else torch.bfloat16)
        res[mode] = (abs(float(loss) - float(ref_loss)) / float(ref_loss), rel(logits.cpu()[dm], ref_logits[dm]),
                     rel(enc.cpu()[am], ref_enc[am]))
    model._engine.set_precision(False)
    for mode, (dl, el, ee) in res.items():
        print(f"[vcg_base b=2 {case}] {mode}: loss rel {dl:.2e}  logits rel {el:.2e}  encoder rel {ee:.2e}")
    dl, el, ee = res["fp32"]
    assert dl < 1e-4 and el < FP32_TOL and ee < FP32_TOL
    assert res["bf16"][0] < 1e-3     # the product path's own loss bound


def test_fp32_mode_refuses_training():
    ocfg = G.tiny_config()
    sd = G.golden_state_dict(ocfg)
    from oracle.make_golden import tiny_batch
    from test_model_gpu import build
    model = build(ocfg, sd).eval()
    b = tiny_batch(seed=3)
    model._engine.set_precision(True)
    model.train()
    with pytest.raises(Exception, match="validation"):
        run(model, b)
    # and back: the bf16 path still works on the same handle
    model._engine.set_precision(False)
    model.eval()
    with torch.no_grad():
        loss = run(model, b)[0]
    assert torch.isfinite(loss)


def test_fp32_mode_generation_matches_the_golden_searches_tightly(gold_dir):
    """generate() in the validation mode (no KV cache: an exact-fp32 eval forward over the rows' tokens per step, the
    reference's host bookkeeping on its logits): every golden greedy / beam case of the TRAINED tiny model -- ids identical,
    scores to 1e-4 (the bf16 product path: 3e-2, tests/test_model_gpu.py)."""
    import json
    import os
    import numpy as np
    from test_model_gpu import build
    gen = json.load(open(os.path.join(gold_dir, "tiny_generate.json")))
    ocfg = G.tiny_config()
    model = build(ocfg, G.trained_state_dict()).eval()
    ids, am = torch.tensor(gen["input_ids"]), torch.tensor(gen["attention_mask"])
    feats = G.golden_features(gen["regions"], seed=gen["seed"])
    model._engine.set_precision(True)
    try:
        worst = 0.0
        for case in gen["cases"]:
            kw = case["kwargs"]
            out = model.generate(input_ids=ids.to(DEV), image_features=[f.to(DEV) for f in feats], attention_mask=am.to(DEV),
                                 return_scores="scores" in case, **kw)
            if "scores" in case:
                got, scores = out
                assert got.cpu().tolist() == case["ids"], kw
                worst = max(worst, float(np.abs(scores.numpy() - np.array(case["scores"])).max()))
            else:
                assert out.cpu().tolist() == case["ids"], kw
        print("[fp32 generation, tiny trained model] %d cases, ids identical, worst score gap %.2e" % (len(gen["cases"]), worst))
        assert worst < 1e-4
    finally:
        model._engine.set_precision(False)


def test_fp32_mode_full_size_beam5_search_is_the_oracles():
    """BASELINE config 5 at full size in the validation mode: vcg_base, b = 4 ragged, num_beams = 5, max_length = 10 on the
    re-scaled random weights of tests/test_decode_fused_gpu.py (beams compete, hypotheses overtake): EVERY row's ids equal
    oracle.generate's and the length-normalised scores agree to 1e-3 -- the bf16 product path's 3e-2 and its one allowed tie
    are bf16 rounding, not the search."""
    ocfg = O.OracleConfig.from_dict(BASE)
    sd = G.golden_state_dict(ocfg, seed=11)
    sd["model.shared.weight"] = sd["model.shared.weight"] * 8.0
    sd["model.decoder.embed_positions.weight"] = sd["model.decoder.embed_positions.weight"] * 40.0
    for k_ in list(sd):
        if k_.endswith("out_proj.weight") or k_.endswith("fc2.weight"):
            sd[k_] = sd[k_] * 3.0
    b = make_batch(4, seed=4321, regions=(36, 20, 36, 7), event_lens=(23, 7, 15, 23), label_lens=(32, 19, 32, 8))
    kw = dict(max_length=10, num_beams=5, num_return_sequences=1, early_stopping=True)
    with torch.no_grad():
        ref_ids, ref_sc = O.generate(sd, ocfg, b["input_ids"], b["image_features"], b["attention_mask"], return_scores=True, **kw)
    model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(BASE))
    model.load_state_dict(sd, strict=False)
    model.to(DEV).eval()
    model._engine.set_precision(True)
    try:
        got, sc = model.generate(input_ids=b["input_ids"].to(DEV), image_features=[f.to(DEV) for f in b["image_features"]],
                                 attention_mask=b["attention_mask"].to(DEV), return_scores=True, **kw)
    finally:
        model._engine.set_precision(False)
    print("[fp32 generation, vcg_base] ids %s oracle %s score gap %.2e" % (got.cpu().tolist(), ref_ids.tolist(),
                                                                          float((sc.float().cpu() - ref_sc.float()).abs().max())))
    assert len({tuple(r) for r in ref_ids.tolist()}) > 1                 # the searches depend on the batch item
    assert any(len(set(r[2:])) > 1 for r in ref_ids.tolist())           # ... and move along the sequence
    assert got.cpu().tolist() == ref_ids.tolist()
    assert float((sc.float().cpu() - ref_sc.float()).abs().max()) < 1e-3
