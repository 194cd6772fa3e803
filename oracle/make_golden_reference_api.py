"""Pins two host-side rows of SURVEY section 8 on the REFERENCE ITSELF (test infrastructure; build container only):

    python -m oracle.make_golden_reference_api          # from /root/repo, needs /root/reference

  a1  `MultiModalBartConfig` -- imports the reference's own class (/root/reference/src/model/config.py:4-92; it subclasses
      the installed transformers' BartConfig, which imports here) and dumps, for every attribute the product's config
      defines: the defaults, `from_dict(config/vcg_base.json)` and `from_dict(config/pretrain_base.json)` ->
      tests/golden/config_reference.json.  tests/test_host_logic_cpu.py requires the product config to equal it.
      Not dumped from the reference: the generation defaults of `PretrainedConfig` (max_length, num_beams, ...): the
      installed transformers is 5.15, whose config no longer carries the 3.0.2 values the reference relies on
      (mixins.py:150-173); those stay restated from SURVEY section 8 a1 and are listed under "restated" in the file.
  a15 `generate_text` -- imports the reference's own function (/root/reference/src/generation.py:6-52; it needs only
      torch.cuda.amp.autocast) and drives it over an adapter whose `.generate` is the ORACLE's beam search on the trained
      tiny fixture; records the keyword arguments the reference passes to `generate` and the records it returns ->
      tests/golden/generate_text_reference.json.  The product's `src.generation.generate_text` must pass the same
      keyword arguments and return the same records, on the same adapter (CPU test) and on the HIP path (GPU test).

What this does NOT pin: the arithmetic (transformers 3.0.2 is absent; DESIGN.md section 2, "parity unpinned" stays).
"""
import importlib
import json
import os
import sys
import types

import torch

from . import goldenlib as G
from . import kmbart_oracle as O
from .make_golden import copy_task_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

# generation defaults of transformers 3.0.2's PretrainedConfig (restated: the installed 5.15 dropped them from the config)
RESTATED = ("max_length", "min_length", "do_sample", "early_stopping", "num_beams", "temperature", "top_k", "top_p",
            "repetition_penalty", "length_penalty", "no_repeat_ngram_size", "bad_words_ids", "num_return_sequences",
            "use_cache", "output_attentions", "output_hidden_states", "model_type")


class _reference_modules:
    """`src.*` resolved from /root/reference for the duration of the block (the product has same-named modules)."""

    def __enter__(self):
        self.saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "src" or k.startswith("src.")}
        sys.path.insert(0, REF)
        return self

    def __exit__(self, *exc):
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
            del sys.modules[k]
        sys.modules.update(self.saved)


def _plain(v):
    if isinstance(v, tuple):
        return list(v)
    return v


def dump_config():
    sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
    from src.model.config import _DEFAULTS   # the product's attribute list (names only are used)
    names = [k for k in _DEFAULTS if k not in RESTATED]
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    with _reference_modules():
        rc = importlib.import_module("src.model.config")
        assert os.path.realpath(rc.__file__).startswith(REF), rc.__file__
        ref_cls = rc.MultiModalBartConfig

        def attrs(cfg):
            return {k: _plain(getattr(cfg, k)) for k in names}

        out = {"generated_by": "oracle/make_golden_reference_api.py from /root/reference/src/model/config.py",
               "restated_not_dumped": list(RESTATED),
               "defaults": attrs(ref_cls())}
        for tag in ("vcg_base", "pretrain_base"):
            d = json.load(open(os.path.join(REF, "config", tag + ".json")))
            out[tag] = attrs(ref_cls.from_dict(d))
        # the attribute writes of vcg_train.py:71-83 after from_dict
        cfg = ref_cls.from_dict(json.load(open(os.path.join(REF, "config", "vcg_base.json"))))
        cfg.dropout, cfg.attention_dropout, cfg.classif_dropout, cfg.activation_dropout = 0.3, 0.2, 0.1, 0.05
        out["vcg_base_after_cli_writes"] = attrs(cfg)
    json.dump(out, open(os.path.join(GOLD, "config_reference.json"), "w"), indent=1, sort_keys=True)
    print("[config] %d attributes x 4 dumps from the reference class" % len(names))
    return out


class OracleGenerateAdapter:
    """What the reference's generate_text needs from a model: eval() and generate(**kw) -> LongTensor; served by the oracle."""

    def __init__(self, cfg, sd, calls=None):
        self.cfg, self.sd, self.calls = cfg, sd, calls if calls is not None else []

    def eval(self):
        return self

    def generate(self, **kw):
        self.calls.append({k: (v if isinstance(v, (int, float, bool)) or v is None else type(v).__name__) for k, v in kw.items()})
        return O.generate(self.sd, self.cfg, kw["input_ids"], kw["image_features"], kw["attention_mask"],
                          num_beams=kw["num_beams"], num_return_sequences=kw["num_return_sequences"],
                          do_sample=kw["do_sample"], top_p=kw["top_p"], top_k=kw["top_k"],
                          early_stopping=kw["early_stopping"], max_length=12)


def gen_loader(n_batches=2, bsz=3):
    out = []
    for i in range(n_batches):
        b = copy_task_batch(40 + i, bsz)
        b["index"] = [100 * i + j for j in range(bsz)]
        b["task_type"] = [("before", "intent", "after")[(i + j) % 3] for j in range(bsz)]
        out.append(b)
    return out


def dump_generate_text():
    cfg = G.tiny_config()
    sd = G.trained_state_dict()
    cases = [dict(num_beams=3, num_gen=2), dict(num_beams=1, num_gen=1)]
    out = {"generated_by": "oracle/make_golden_reference_api.py: /root/reference/src/generation.py generate_text over the "
                           "oracle's generate on tests/golden/tiny_trained_fp16.npz", "cases": []}
    with _reference_modules():
        gen = importlib.import_module("src.generation")
        assert os.path.realpath(gen.__file__).startswith(REF), gen.__file__
        for c in cases:
            calls, lines = [], []
            model = OracleGenerateAdapter(cfg, sd, calls)
            args = types.SimpleNamespace(amp=False, **c)
            logger = types.SimpleNamespace(info=lambda m: lines.append(m))
            recs = gen.generate_text(model, gen_loader(), G.IdTokenizer(), args, torch.device("cpu"), logger=logger,
                                     log_interval=1)
            out["cases"].append({"args": c, "generate_kwargs": calls, "records": recs,
                                 "log_prefixes": [ln.split(", ETA")[0] for ln in lines]})
            print("[generate_text]", c, recs[0])
    json.dump(out, open(os.path.join(GOLD, "generate_text_reference.json"), "w"), indent=1)
    return out


def processors_batch():
    """copy-task inputs whose event text REPEATS tokens and bigrams (the model reverses the text, so the plain search
    repeats them too and the repetition penalty / n-gram ban have something to act on)"""
    b = copy_task_batch(21, 4)
    mask = b["input_ids"] == G.TINY["img_feat_id"]
    for i in range(4):
        r = int(mask[i].sum())
        ev = b["input_ids"][i, 4 + r:]
        e = int((ev < G.TINY_SPECIAL_BASE).long().cumprod(0).sum())     # event tokens up to </event>
        if e >= 4:
            ev[2] = ev[0]
            ev[3] = ev[1]            # a b a b ...
        if e >= 6:
            ev[5] = ev[0]
    return b


def dump_generate_processors():
    """Score post-processing of generate (repetition_penalty / no_repeat_ngram_size / bad_words_ids; reference
    src/model/mixins.py:150-235 validates them, transformers 3.0.2 postprocess_next_token_scores applies them): the oracle's
    restatement on the trained tiny fixture, cross-checked against transformers 5.15 `generate()` (same processors, the
    bad-words length quirk of 3.0.2 does not trigger in these cases) -> tests/golden/tiny_generate_processors.json."""
    from transformers.modeling_outputs import BaseModelOutput
    from .make_golden import hf_model
    cfg, sd = G.tiny_config(), G.trained_state_dict()
    b = processors_batch()
    hf = hf_model(cfg, sd)
    with torch.no_grad():
        enc = O.encoder_forward(sd, cfg, b["input_ids"], b["image_features"], b["attention_mask"])
    plain = O.generate(sd, cfg, b["input_ids"], b["image_features"], b["attention_mask"], num_beams=3, max_length=12, early_stopping=True)
    bad = [[int(plain[0, 2])], [int(plain[1, 2]), int(plain[1, 3])]]     # a token and a bigram the plain search produces
    cases = [dict(num_beams=3, max_length=12, early_stopping=True, repetition_penalty=200.0),
             dict(num_beams=3, max_length=12, early_stopping=True, no_repeat_ngram_size=2),
             dict(num_beams=3, max_length=12, early_stopping=True, bad_words_ids=bad),
             dict(num_beams=4, num_return_sequences=2, max_length=12, early_stopping=True, repetition_penalty=50.0,
                  no_repeat_ngram_size=3, bad_words_ids=bad),
             dict(num_beams=1, max_length=12, repetition_penalty=1.3, no_repeat_ngram_size=2, bad_words_ids=bad)]
    out = {"generated_by": "oracle/make_golden_reference_api.py::dump_generate_processors", "seed": 21, "batch": 4,
           "plain_ids": plain.tolist(), "cases": []}
    for kw in cases:
        beams = kw.get("num_beams", 1)
        r = O.generate(sd, cfg, b["input_ids"], b["image_features"], b["attention_mask"], return_scores=beams > 1, **kw)
        ids = r[0] if beams > 1 else r
        h = hf.generate(encoder_outputs=BaseModelOutput(last_hidden_state=enc.clone()), attention_mask=b["attention_mask"],
                        forced_bos_token_id=0 if beams > 1 else None, forced_eos_token_id=2 if beams > 1 else None,
                        decoder_start_token_id=0, do_sample=False, **kw)
        L = max(h.shape[1], ids.shape[1])
        same = bool((torch.nn.functional.pad(h, (0, L - h.shape[1]), value=1) ==
                     torch.nn.functional.pad(ids, (0, L - ids.shape[1]), value=1)).all())
        changed = (ids.shape != plain.shape or not bool((ids == plain).all())) if beams > 1 and kw.get("num_return_sequences", 1) == 1 else None
        if changed is not None:
            assert changed, ("the option does not change this search: the case would test nothing", kw)
        print("[processors]", kw, "== transformers 5.15" if same else "!= transformers 5.15", "| differs from the plain search:", changed)
        assert same, kw
        rec = {"kwargs": kw, "ids": ids.tolist(), "identical_to_transformers_5_15": same}
        if beams > 1:
            rec["scores"] = [float(x) for x in r[1]]
        out["cases"].append(rec)
    json.dump(out, open(os.path.join(GOLD, "tiny_generate_processors.json"), "w"))
    return out


if __name__ == "__main__":
    torch.set_num_threads(8)
    dump_config()
    dump_generate_text()
    dump_generate_processors()
