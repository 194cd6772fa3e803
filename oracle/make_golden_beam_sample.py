"""TEST INFRASTRUCTURE (build container only; needs the installed transformers 5.15).  An independent pin of the beam-search SAMPLING branch
(`do_sample=True` with `num_beams > 1`, reached from the reference's src/model/mixins.py:336-361 -> transformers 3.0.2 `_generate_beam_search`),
which rounds 1-5 could only compare through shared bookkeeping because multinomial draws are not reproducible across implementations.

The trick: with `top_k=2` every beam keeps exactly two tokens (the filter runs on log-prob + beam score with min_tokens_to_keep=2), so the
2 * num_beams draws WITHOUT replacement from the 2 * num_beams non-zero entries of a batch item take all of them -- the candidate set, and after the
sort the whole search, no longer depends on the random stream.  What is then pinned against transformers 5.15 `generate(do_sample=True, num_beams=k,
top_k=2)`: every beam starts at score 0 (no -1e9 on beams 1..k-1 as in the greedy beam search), no forced BOS / EOS, the filter placement, the
2k-candidate bookkeeping, hypotheses and early stopping.  Ids AND length-normalised scores are identical whenever every search ends on EOS before
max_length (a hypothesis cut at max_length is scored differently by 4.x+: printed, not asserted -- the same documented difference as for
early_stopping=False in make_golden.py).  Writes tests/golden/tiny_generate_beam_sample.json.

    python oracle/make_golden_beam_sample.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
from oracle import goldenlib as G  # noqa: E402
from oracle import kmbart_oracle as O  # noqa: E402
from oracle.make_golden import copy_task_batch, hf_model  # noqa: E402

CASES = (dict(num_beams=3, max_length=14, early_stopping=True, top_k=2),
         dict(num_beams=5, max_length=14, early_stopping=True, top_k=2),
         dict(num_beams=4, max_length=14, early_stopping=True, top_k=2, min_length=5),
         dict(num_beams=3, max_length=14, early_stopping=True, top_k=2, length_penalty=2.0))


def main():
    from transformers.modeling_outputs import BaseModelOutput
    cfg, sd = G.tiny_config(), G.trained_state_dict()
    b = copy_task_batch(5, 6)
    hf = hf_model(cfg, sd)
    with torch.no_grad():
        enc = O.encoder_forward(sd, cfg, b["input_ids"], b["image_features"], b["attention_mask"])
    rec = {"seed": 5, "batch": 6, "input_ids": b["input_ids"].tolist(), "attention_mask": b["attention_mask"].tolist(),
           "regions": [len(f) for f in b["image_features"]], "cases": []}
    for kw in CASES:
        outs = []
        for seed in (0, 123, 99):   # the search must not depend on the random stream
            torch.manual_seed(seed)
            outs.append(O.generate(sd, cfg, b["input_ids"], b["image_features"], b["attention_mask"], do_sample=True, return_scores=True, **kw))
        ref, rsc = outs[0]
        assert all(torch.equal(o[0], ref) and torch.equal(o[1], rsc) for o in outs[1:]), kw
        assert int((ref == cfg.eos_token_id).any(dim=1).sum()) == ref.shape[0] and ref.shape[1] < kw["max_length"], "every search must end on EOS"
        torch.manual_seed(7)
        out = hf.generate(encoder_outputs=BaseModelOutput(last_hidden_state=enc.clone()), attention_mask=b["attention_mask"],
                          decoder_start_token_id=0, do_sample=True, output_scores=True, return_dict_in_generate=True,
                          forced_bos_token_id=None, forced_eos_token_id=None, **kw)
        hs = out.sequences
        L = max(hs.shape[1], ref.shape[1])
        pad = lambda t: torch.nn.functional.pad(t, (0, L - t.shape[1]), value=cfg.pad_token_id)   # noqa: E731
        same_ids = bool((pad(hs) == pad(ref)).all())
        same_sc = bool(torch.allclose(out.sequences_scores, rsc, atol=1e-5))
        print(f"[beam-sample crosscheck vs transformers 5.15] {kw}: ids {'==' if same_ids else '!='} scores {'==' if same_sc else '!='}")
        assert same_ids and same_sc, kw
        rec["cases"].append({"kwargs": kw, "ids": ref.tolist(), "scores": [float(x) for x in rsc], "identical_to_transformers_5_15": True})
    json.dump(rec, open(os.path.join(ROOT, "tests", "golden", "tiny_generate_beam_sample.json"), "w"))
    print("wrote tests/golden/tiny_generate_beam_sample.json")


if __name__ == "__main__":
    main()
