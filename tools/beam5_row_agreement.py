"""How often does a row of the FULL-SIZE beam-5 search differ from the oracle's, and is a differing row a tie?  (Weak 2:
tests/test_decode_fused_gpu.py::test_full_size_beam5_search_matches_the_oracle accepts 3 of 4 identical rows.)  Same construction as that test --
vcg_base dimensions, random weights re-scaled until the searches depend on item and position, b = 4 ragged, num_beams = 5, max_length = 10,
early_stopping -- over several weight / batch seeds and both sublayer scales; prints per case the identical rows and, for a differing row, the
oracle's OWN score of the product's sequence against its winner's.
    python tools/beam5_row_agreement.py [n_seeds=6]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "km-bart_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from test_decode_fused_gpu import BASE, DEV, G, O, _oracle_sequence_score, make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ocfg = O.OracleConfig.from_dict(BASE)
rows = same_rows = ties = 0
worst_gap = 0.0
for seed in range(n_seeds):
    for scale in (3.0, 1.0):
        sd = G.golden_state_dict(ocfg, seed=11 + seed)
        sd["model.shared.weight"] = sd["model.shared.weight"] * 8.0
        sd["model.decoder.embed_positions.weight"] = sd["model.decoder.embed_positions.weight"] * 40.0
        for k_ in list(sd):
            if k_.endswith("out_proj.weight") or k_.endswith("fc2.weight"):
                sd[k_] = sd[k_] * scale
        b = make_batch(4, seed=4321 + seed, regions=(36, 20, 36, 7), event_lens=(23, 7, 15, 23), label_lens=(32, 19, 32, 8))
        kw = dict(max_length=10, num_beams=5, num_return_sequences=1, early_stopping=True)
        with torch.no_grad():
            ref_ids, ref_sc = O.generate(sd, ocfg, b["input_ids"], b["image_features"], b["attention_mask"], return_scores=True, **kw)
        model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(BASE))
        model.load_state_dict(sd, strict=False)
        model.to(DEV).eval()
        got, sc = model.generate(input_ids=b["input_ids"].to(DEV), image_features=[f.to(DEV) for f in b["image_features"]],
                                 attention_mask=b["attention_mask"].to(DEV), return_scores=True, **kw)
        got = got.cpu()
        n = max(got.shape[1], ref_ids.shape[1])
        pad = lambda t: torch.nn.functional.pad(t, (0, n - t.shape[1]), value=ocfg.pad_token_id)   # noqa: E731
        same = (pad(got) == pad(ref_ids)).all(dim=1)
        note = ""
        for r in range(4):
            rows += 1
            if bool(same[r]):
                same_rows += 1
                continue
            alt = _oracle_sequence_score(sd, ocfg, b, r, got[r].tolist())
            best = _oracle_sequence_score(sd, ocfg, b, r, ref_ids[r].tolist())
            worst_gap = max(worst_gap, best - alt)
            ties += int(alt >= best - 2e-2)
            note += "  row %d differs: oracle scores its own winner %.4f, the product's sequence %.4f" % (r, best, alt)
        print("weights seed %d, sublayers x %g: %d / 4 rows identical; distinct oracle sequences %d%s"
              % (11 + seed, scale, int(same.sum()), len({tuple(x) for x in ref_ids.tolist()}), note), flush=True)
        del model
print("TOTAL: %d of %d rows identical to the oracle (%.1f %%); %d differing rows, %d of them ties by the oracle's own scoring (gap <= 2e-2), worst gap %.4f"
      % (same_rows, rows, 100.0 * same_rows / rows, rows - same_rows, ties, worst_gap))
