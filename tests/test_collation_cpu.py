"""Batch construction against outputs of the REFERENCE's own ConditionTokenizer / Collator.

tests/golden/collation_cases.json was written by oracle/make_golden_collation.py, which runs the reference classes
on closed-form synthetic dataset entries with the small vocabulary in tests/golden/tiny_bpe_tokenizer.json.  Every
integer tensor must match exactly, including the seeded MLM / MRM masks."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "km-bart_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle.make_golden_collation import entry_from_recipe  # noqa: E402  (recipe -> entry; no reference import)
from src.data.collation import Collator  # noqa: E402
from src.data.offline_tokenizer import load_base_tokenizer  # noqa: E402
from src.data.tokenization import ConditionTokenizer  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
G = json.load(open(os.path.join(GOLD, "collation_cases.json")))


def T(j):
    return torch.tensor(j["data"], dtype=getattr(torch, j["dtype"])).view(j["shape"])


@pytest.fixture(scope="module")
def tokenizer():
    return ConditionTokenizer(base_tokenizer=load_base_tokenizer(os.path.join(GOLD, "tiny_bpe_tokenizer.json")))


def test_marker_ids_and_size(tokenizer):
    for k, v in G["special_ids"].items():
        assert getattr(tokenizer, k) == v, k
    assert len(tokenizer) == G["len"]
    # the 16 markers sit right behind the base vocabulary, in the reference's order (tokenization.py:36-57)
    assert tokenizer.begin_img_id == tokenizer.vocab_size
    assert tokenizer.region_caption_id == tokenizer.vocab_size + 15


def test_encode_condition_and_label(tokenizer):
    cond = tokenizer.encode_condition(task_type=["intent", "caption"], img_num=[2, 0], event=["PersonX walks", ""],
                                      mlm=["a man", "two dogs are playing"])
    assert set(cond.keys()) == set(G["encode_condition"].keys())
    for k, v in G["encode_condition"].items():
        assert cond[k].dtype == getattr(torch, v["dtype"]) and torch.equal(cond[k], T(v)), k
    lab = tokenizer.encode_label(label=["a red umbrella", "the park"], img_num=[1, 3])
    assert set(lab.keys()) == set(G["encode_label_img"].keys())
    for k, v in G["encode_label_img"].items():
        assert torch.equal(lab[k], T(v)), k
    lab = tokenizer.encode_label(label=["a red umbrella", ""])
    assert set(lab.keys()) == set(G["encode_label"].keys())
    for k, v in G["encode_label"].items():
        assert torch.equal(lab[k], T(v)), k
    with pytest.raises(ValueError):
        tokenizer.encode_condition(task_type="nonsense")


@pytest.mark.parametrize("case", G["cases"], ids=[c["name"] for c in G["cases"]])
def test_collator_matches_reference(tokenizer, case):
    batch = [entry_from_recipe(r) for r in case["recipes"]]
    originals = [None if "image_features" not in e else e["image_features"].copy() for e in batch]
    collate = Collator(tokenizer, **case["collator"])
    torch.manual_seed(case["seed"])
    out = collate(batch)
    exp = case["outputs"]
    feats = out.pop("image_features")
    soft = out.pop("mrm_labels", None)
    assert set(out.keys()) == set(exp.keys()) - {"mrm_label_rows"}
    for k, v in exp.items():
        if k == "mrm_label_rows":
            continue
        if isinstance(v, dict) and "dtype" in v:
            assert out[k].dtype == getattr(torch, v["dtype"]), k
            assert torch.equal(out[k], T(v)), k
        elif k == "attribute_labels":
            assert [t.tolist() for t in out[k]] == [t["data"] for t in v]
        else:
            assert out[k] == v, k
    # region features: one packed buffer that still reads like the reference's list
    assert len(feats) == len(batch)
    max_img = case["collator"].get("max_img_num", 30)
    for i, (f, s) in enumerate(zip(feats, case["image_features"])):
        if originals[i] is None:
            assert f.numel() == 0 and s["regions"] == 0
            continue
        assert f.shape[0] == s["regions"]
        if s["regions"] == 0:
            continue
        src = torch.from_numpy(originals[i][:max_img])
        zeroed = (f[:, :2048].abs().sum(1) == 0).nonzero().flatten().tolist()
        assert zeroed == s["zeroed"]
        keep = [r for r in range(f.shape[0]) if r not in zeroed]
        assert torch.equal(f[keep], src[keep])
        assert torch.equal(f[zeroed][:, 2048:], src[zeroed][:, 2048:])       # box coordinates survive MRM
        assert abs(float(f.double().sum()) - s["sum"]) < 1e-6 * max(1.0, abs(s["sum"]))
    assert int(feats.offsets[-1]) == feats.n_total == sum(s["regions"] for s in case["image_features"])
    if "mrm_label_rows" in exp:
        for e, got, want in zip(batch, soft, exp["mrm_label_rows"]):
            assert list(got.shape) == want["shape"]
            assert torch.equal(got, torch.from_numpy(e["mrm_labels"])[want["rows"]])


def test_collator_argument_checks(tokenizer):
    with pytest.raises(ValueError):
        Collator(tokenizer, has_label=False, mlm_enabled=True)
    with pytest.raises(ValueError):
        Collator(tokenizer, ap_enabled=True, mrm_enabled=False)
    with pytest.raises(ValueError):
        Collator(tokenizer, rp_enabled=True, has_label=False)


def test_collator_drops_missing_entries_and_packs_for_the_engine(tokenizer):
    from kmbart.engine import pack_features
    batch = [entry_from_recipe({"task_type": "intent", "event": "PersonX walks", "labels": "read", "regions": 3, "seed": 1}),
             None,
             entry_from_recipe({"task_type": "after", "event": "a man", "labels": "the park"})]
    out = Collator(tokenizer)(batch)
    assert out["input_ids"].shape[0] == 2 and out["task_type"] == ["intent", "after"]
    packed, offsets, n = pack_features(out["image_features"], 2052, "cpu")
    assert n == 3 and offsets.tolist() == [0, 3, 3] and packed.shape == (3, 2052)
    # one <img_feat> per region: the invariant the multimodal embedding checks (modules.py:98-100)
    assert int((out["input_ids"] == tokenizer.img_feat_id).sum()) == 3
