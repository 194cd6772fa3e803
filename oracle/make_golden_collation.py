"""Golden vectors for the batch-construction row.  TEST INFRASTRUCTURE ONLY.

Runs the REFERENCE's own `src.data.collation.Collator` and `src.data.tokenization.ConditionTokenizer`
(/root/reference, imported here, in this container only) on closed-form synthetic dataset entries and writes

    tests/golden/tiny_bpe_tokenizer.json     the vocabulary used (a byte-level BPE trained below on a fixed corpus)
    tests/golden/collation_cases.json        inputs (by recipe) + every tensor the reference collator returned

The reference tokenizer wraps `BartTokenizer.from_pretrained('facebook/bart-large')`; that vocabulary is not in
this image (no network), so the name `BartTokenizer` inside the reference module is pointed at a loader that returns
the small vocabulary above -- data the image lacks, not code: every line of tokenization / collation logic that
runs is the reference's.  `python oracle/make_golden_collation.py` (needs /root/reference; never run on the GPU box).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

CORPUS = [
    "PersonX walks into the kitchen and opens the fridge to get a cold drink",
    "a man is holding a red umbrella in the rain near the bus stop",
    "two dogs are playing with a ball on the green grass in the park",
    "PersonX wants to ask the woman behind the counter for directions to the station",
    "the tall building next to the river has many small windows and a blue door",
    "before PersonX needed to buy a ticket and after PersonX will sit down and read",
]
SPECIALS = ["<s>", "<pad>", "</s>", "<unk>", "<mask>"]   # BART's order for the first four; <mask> follows


def build_tokenizers_json(path):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=420, special_tokens=SPECIALS, show_progress=False,
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet())
    tok.train_from_iterator(CORPUS * 3, trainer)
    tok.save(path)


def hf_tokenizer(path):
    from transformers import PreTrainedTokenizerFast
    return PreTrainedTokenizerFast(tokenizer_file=path, bos_token="<s>", eos_token="</s>", pad_token="<pad>",
                                   unk_token="<unk>", mask_token="<mask>")


def uniform(n, seed):
    """splitmix64 of (index, seed) -> [0, 1): same generator as oracle/goldenlib.lcg_uniform, restated so that this
    script runs with only the reference on sys.path"""
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) \
            + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return ((x >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def entry_from_recipe(r):
    """recipe (JSON-able) -> dataset entry, the dict the reference datasets return (src/data/dataset.py:11-21)"""
    e = {"task_type": r["task_type"]}
    for k in ("event", "labels", "index", "question_id", "dataset_index", "object_ids", "objects", "relations"):
        if k in r:
            e[k] = r[k]
    if r.get("regions", -1) >= 0:
        n = r["regions"]
        e["image_features"] = uniform(n * 2052, r["seed"]).reshape(n, 2052)
        if r.get("mrm", False):
            s = uniform(n * 1601, r["seed"] + 7).reshape(n, 1601).astype(np.float64)
            e["mrm_labels"] = (s / s.sum(1, keepdims=True)).astype(np.float32)
    return e


def vg_recipe(seed, n_obj, index):
    objs = [{"object_id": 100 + i, **({"attribute_ids": [(7 * i + seed) % 129, 3]} if i % 3 != 1 else {})}
            for i in range(n_obj)]
    rels = [{"object_id": 100 + (i % n_obj), "subject_id": 100 + ((i * 2 + 1) % n_obj), "predicate_id": (5 * i + 1) % 201}
            for i in range(6)] + [{"object_id": 999, "subject_id": 100, "predicate_id": 1}]
    return {"task_type": "region_caption", "labels": CORPUS[4], "index": index, "regions": n_obj + 2, "seed": seed,
            "mrm": True, "object_ids": [100 + i for i in range(n_obj)], "objects": objs, "relations": rels}


CASES = [
    dict(name="vcg_fine_tune", seed=0,
         collator=dict(has_label=True, event_max_len=6, lm_max_len=5, max_img_num=4),
         recipes=[
             {"task_type": "intent", "event": CORPUS[0], "labels": CORPUS[3], "index": 11, "regions": 6, "seed": 3},
             {"task_type": "before", "event": CORPUS[1], "labels": CORPUS[5], "index": 12, "regions": 2, "seed": 4},
             {"task_type": "after", "event": "PersonX", "labels": "read", "index": 13},
             {"task_type": "after", "event": CORPUS[2], "labels": CORPUS[0], "index": 14, "regions": 0, "seed": 5},
         ]),
    dict(name="vcg_generate", seed=0,
         collator=dict(has_label=False),
         recipes=[
             {"task_type": "intent", "event": CORPUS[0], "index": 1, "regions": 3, "seed": 6},
             {"task_type": "after", "event": CORPUS[2], "index": 2, "regions": 5, "seed": 7},
         ]),
    dict(name="pretrain_all_tasks", seed=1234,
         collator=dict(has_label=True, mlm_enabled=True, mrm_enabled=True, rp_enabled=True, ap_enabled=True,
                       mlm_probability=0.4, mrm_probability=0.5, lm_max_len=12, max_img_num=8, max_rel_count=4),
         recipes=[
             {"task_type": "caption", "labels": CORPUS[1], "index": 0, "regions": 5, "seed": 8, "mrm": True},
             vg_recipe(9, 4, 1),
             {"task_type": "before", "event": CORPUS[0], "labels": CORPUS[5], "index": 2, "regions": 10, "seed": 10,
              "mrm": True, "dataset_index": 2},
             vg_recipe(11, 9, 3),
         ]),
    dict(name="pretrain_mlm_only", seed=77,
         collator=dict(has_label=True, mlm_enabled=True, mrm_enabled=False, mlm_probability=0.6, lm_max_len=30,
                       max_img_num=30),
         recipes=[
             {"task_type": "caption", "labels": CORPUS[2], "index": 0, "regions": 3, "seed": 12},
             {"task_type": "intent", "event": CORPUS[3], "labels": CORPUS[0], "index": 1, "regions": 1, "seed": 13},
         ]),
]


def jsonable(v):
    if torch.is_tensor(v):
        return {"dtype": str(v.dtype).replace("torch.", ""), "shape": list(v.shape),
                "data": v.to(torch.float64 if v.is_floating_point() else torch.int64).flatten().tolist()}
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    if isinstance(v, dict):
        return {k: jsonable(x) for k, x in v.items()}
    if isinstance(v, (np.integer,)):
        return int(v)
    return v


def main():
    os.makedirs(GOLD, exist_ok=True)
    tok_path = os.path.join(GOLD, "tiny_bpe_tokenizer.json")
    build_tokenizers_json(tok_path)

    sys.path.insert(0, REF)
    import src.data.tokenization as ref_tok
    from src.data.collation import Collator as RefCollator

    class _Loader:
        @staticmethod
        def from_pretrained(name):
            return hf_tokenizer(tok_path)

    ref_tok.BartTokenizer = _Loader
    tokenizer = ref_tok.ConditionTokenizer()

    out = {"special_ids": {k: getattr(tokenizer, k) for k in (
        "begin_img_id", "end_img_id", "begin_event_id", "end_event_id", "before_id", "intent_id", "after_id",
        "img_feat_id", "caption_id", "begin_mlm_id", "end_mlm_id", "cls_token_id", "region_caption_id", "vocab_size",
        "bos_token_id", "eos_token_id", "pad_token_id", "unk_token_id")}, "len": len(tokenizer), "cases": []}

    # the tokenizer methods on their own (tokenization.py:100-250)
    cond = tokenizer.encode_condition(task_type=["intent", "caption"], img_num=[2, 0], event=["PersonX walks", ""],
                                      mlm=["a man", "two dogs are playing"])
    lab = tokenizer.encode_label(label=["a red umbrella", "the park"], img_num=[1, 3])
    lab2 = tokenizer.encode_label(label=["a red umbrella", ""])
    out["encode_condition"] = jsonable(dict(cond))
    out["encode_label_img"] = jsonable(lab)
    out["encode_label"] = jsonable(lab2)

    for case in CASES:
        batch = [entry_from_recipe(r) for r in case["recipes"]]
        collate = RefCollator(tokenizer, **case["collator"])
        torch.manual_seed(case["seed"])
        res = collate(batch)
        feats = res.pop("image_features")
        summary = []
        for f in feats:
            if f.numel() == 0:
                summary.append({"regions": 0, "zeroed": [], "sum": 0.0})
            else:
                zeroed = (f[:, :2048].abs().sum(1) == 0).nonzero().flatten().tolist()
                summary.append({"regions": int(f.shape[0]), "zeroed": zeroed, "sum": float(f.double().sum())})
        if "mrm_labels" in res:  # rows of the entry's own soft labels: store WHICH rows instead of 1601 floats each
            picked = []
            for e, m in zip(batch, res.pop("mrm_labels")):
                src = torch.Tensor(e["mrm_labels"])
                rows = [int((src == row).all(1).nonzero()[0]) for row in m]
                picked.append({"shape": list(m.shape), "rows": rows})
            res["mrm_label_rows"] = picked
        rec = {"name": case["name"], "seed": case["seed"], "collator": case["collator"], "recipes": case["recipes"],
               "image_features": summary, "outputs": jsonable(res)}
        out["cases"].append(rec)
        print(case["name"], {k: (tuple(v.shape) if torch.is_tensor(v) else type(v).__name__) for k, v in res.items()})
    with open(os.path.join(GOLD, "collation_cases.json"), "w") as f:
        json.dump(out, f)
    print("wrote", os.path.join(GOLD, "collation_cases.json"), os.path.getsize(os.path.join(GOLD, "collation_cases.json")), "bytes")


if __name__ == "__main__":
    main()
