"""Pins the oracle and (re)generates tests/golden/*.  Run in the BUILD container only:

    python -m oracle.make_golden            # from /root/repo

It needs `/root/reference` (it imports the reference's own `src.training` -- `fine_tune` -- and reads
`config/vcg_base.json`; the reference's `src.model.config` and `src.generation` are imported by the sibling script
`oracle/make_golden_reference_api.py`, which pins rows a1 and a15 on them) and the installed transformers 5.15 BART
(independent cross-check).  Neither travels to the GPU box; only the small vectors written here do.

Steps
 1. cross-check `oracle.kmbart_oracle.forward` (+ autograd) against transformers 5.15
    `BartForConditionalGeneration` on a shared state-dict (tiny ragged batch and a vcg_base-shaped
    batch): logits / loss / every gradient must agree to <= 2e-5 relative.  The KM-BART-specific
    multimodal embedding on the transformers side is NOT the oracle's: it is a plain nn.Linear +
    nn.Embedding + masked index assignment written from reference src/model/modules.py:24-41,89-102.
 2. drive `OracleModel` + `HFAdamW` through the REFERENCE's `src.training.fine_tune`
    (reference src/training.py:96-171) for 3 steps; check that the loop order matches the
    oracle's own step loop bit-for-bit and record the loss sequence.
 3. train the tiny model on a reverse-the-event-text task (800 HF-AdamW steps on the oracle, CPU) so that
    generation is NOT degenerate (peaked, input-dependent distributions: no near-ties for bf16 to flip), store the
    weights as fp16 (tests/golden/tiny_trained_fp16.npz), and cross-check the oracle's beam search against
    transformers 5.15 `generate()` on them: token ids AND length-normalised scores must be identical for every
    early_stopping=True case (the setting of reference src/generation.py:22-32), min_length included.
    Known algorithm difference, printed not asserted: with early_stopping=False transformers >= 4.x bounds the
    attainable score with max_length (`is_done`), 3.0.2 with the current length.
 4. write tests/golden/tiny_train.npz, tiny_generate.json.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

from . import kmbart_oracle as O
from . import goldenlib as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
from src.data.synthetic import make_batch  # noqa: E402  (the product's synthetic generator)


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def tiny_batch(regions=(6, 3), event_lens=(8, 4), label_lens=(12, 7), enc_len=24, dec_len=12, seed=5):
    b = make_batch(len(regions), enc_len=enc_len, dec_len=dec_len, regions=regions, event_lens=event_lens,
                   label_lens=label_lens, seed=seed, vocab_hi=G.TINY_SPECIAL_BASE,
                   img_feat_id=G.TINY["img_feat_id"], special_base=G.TINY_SPECIAL_BASE)
    b["image_features"] = G.golden_features(regions)
    return b


def copy_task_batch(seed, bsz=32, train=False):
    """Reverse-copy task of the generation fixtures: encoder row = task, <img>, R region placeholders, </img>, <event>,
    e text tokens from a 61-token alphabet, </event>; target = <s>, the text reversed, </s> (the leading <s> is what the
    reference's forced-BOS step produces after decoder_start_token_id = 0, src/model/mixins.py:400-405)."""
    g = torch.Generator().manual_seed(seed)
    regions = torch.randint(0, 7, (bsz,), generator=g).tolist()
    ev = torch.randint(2, 9, (bsz,), generator=g).tolist()
    b = make_batch(bsz, enc_len=24, dec_len=12, regions=regions, event_lens=ev, label_lens=[e + 2 for e in ev],
                   seed=seed, vocab_hi=64, img_feat_id=G.TINY["img_feat_id"], special_base=G.TINY_SPECIAL_BASE)
    b["image_features"] = G.golden_features(regions, seed=seed)
    if train:
        for i in range(bsz):
            r, e = regions[i], ev[i]
            tgt = torch.flip(b["input_ids"][i, 4 + r: 4 + r + e], [0])
            b["decoder_input_ids"][i, 1] = 0
            b["decoder_input_ids"][i, 2: e + 2] = tgt
            b["labels"][i, 0] = 0
            b["labels"][i, 1: e + 1] = tgt
            b["labels"][i, e + 1] = 2
    return b


def train_generation_model(steps=800):
    """Step 3: the tiny oracle model trained on copy_task_batch (CPU, ~40 s); returns the fp16-rounded state dict."""
    cfg = G.tiny_config()
    model = O.OracleModel(cfg, state_dict=G.golden_state_dict(cfg, seed=3)).train()
    opt = O.HFAdamW(model.parameters(), lr=2e-3)
    for step in range(steps):
        b = copy_task_batch(1000 + step, train=True)
        loss = model(b["input_ids"], b["image_features"], b["attention_mask"], decoder_input_ids=b["decoder_input_ids"],
                     decoder_attention_mask=b["decoder_attention_mask"], labels=b["labels"])[0]
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 100 == 0 or step == steps - 1:
            print(f"[train tiny] step {step} loss {float(loss.detach()):.4f}", flush=True)
    return {k: v.detach().half().float() for k, v in model.sd().items()}


def hf_model(cfg, sd):
    from transformers import BartConfig, BartForConditionalGeneration
    hc = BartConfig(vocab_size=cfg.vocab_size, d_model=cfg.d_model, encoder_layers=cfg.encoder_layers,
                    decoder_layers=cfg.decoder_layers, encoder_attention_heads=cfg.encoder_attention_heads,
                    decoder_attention_heads=cfg.decoder_attention_heads, encoder_ffn_dim=cfg.encoder_ffn_dim,
                    decoder_ffn_dim=cfg.decoder_ffn_dim, max_position_embeddings=cfg.max_position_embeddings,
                    dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, activation_function="gelu",
                    scale_embedding=False, pad_token_id=1, bos_token_id=0, eos_token_id=2,
                    decoder_start_token_id=0, attn_implementation="eager")
    hf = BartForConditionalGeneration(hc).eval()
    hsd = {k: v for k, v in sd.items() if "embed_images" not in k}
    hsd["model.encoder.embed_tokens.weight"] = sd["model.shared.weight"]
    hsd["model.decoder.embed_tokens.weight"] = sd["model.shared.weight"]
    hsd["lm_head.weight"] = sd["model.shared.weight"]
    missing, unexpected = hf.load_state_dict(hsd, strict=False)
    assert not unexpected, unexpected
    assert all("embed_images" not in m for m in missing), missing
    return hf


def reference_style_multimodal_embedding(sd, cfg, input_ids, image_features):
    """The KM-BART-specific part of the encoder input, written from the reference source rather than taken from the
    oracle: ImageEmbedding.forward (src/model/modules.py:24-41: cat the non-empty feature lists, one nn.Linear, split
    back, torch.empty(0) for an empty sample) and _embed_multi_modal (:89-102: nn.Embedding lookup, then per sample
    `embedded[index, mask[index]] = value` where mask marks <img_feat> / <cls> ids)."""
    lin = torch.nn.Linear(cfg.image_feature_size, cfg.d_model)
    emb = torch.nn.Embedding(cfg.vocab_size, cfg.d_model, padding_idx=cfg.pad_token_id)
    with torch.no_grad():
        lin.weight.copy_(sd["model.encoder.embed_images.linear.weight"])
        lin.bias.copy_(sd["model.encoder.embed_images.linear.bias"])
        emb.weight.copy_(sd["model.shared.weight"])
    img_len = [len(x) for x in image_features]
    non_empty = [x for x in image_features if len(x) > 0]
    out = lin(torch.cat(non_empty, dim=0)) if non_empty else None
    per_sample, index = [], 0
    for n in img_len:
        per_sample.append(out[index: index + n] if n > 0 else torch.empty(0))
        index += n
    mask = (input_ids == cfg.img_feat_id) | (input_ids == cfg.cls_token_id)
    embedded = emb(input_ids)
    for index, value in enumerate(per_sample):
        if len(value) > 0:
            embedded[index, mask[index]] = value
    return embedded


def hf_generate_crosscheck(cfg, sd, batch, cases):
    """Step 3: oracle beam search vs transformers 5.15 generate() on the trained tiny model.  Returns, per case,
    whether ids and scores were identical."""
    from transformers.modeling_outputs import BaseModelOutput
    hf = hf_model(cfg, sd)
    with torch.no_grad():
        enc = O.encoder_forward(sd, cfg, batch["input_ids"], batch["image_features"], batch["attention_mask"])
    verdicts = []
    for kw in cases:
        if kw.get("num_beams", 1) == 1:
            verdicts.append(None)
            continue
        ref, rsc = O.generate(sd, cfg, batch["input_ids"], batch["image_features"], batch["attention_mask"],
                              return_scores=True, **kw)
        out = hf.generate(encoder_outputs=BaseModelOutput(last_hidden_state=enc.clone()),
                          attention_mask=batch["attention_mask"], forced_bos_token_id=0, forced_eos_token_id=2,
                          decoder_start_token_id=0, do_sample=False, output_scores=True, return_dict_in_generate=True,
                          **kw)
        hs = out.sequences
        L = max(hs.shape[1], ref.shape[1])
        same_ids = bool((torch.nn.functional.pad(hs, (0, L - hs.shape[1]), value=1) ==
                         torch.nn.functional.pad(ref, (0, L - ref.shape[1]), value=1)).all())
        same_sc = same_ids and bool(torch.allclose(out.sequences_scores, rsc, atol=1e-5))
        print(f"[generate crosscheck vs transformers 5.15] {kw}: ids {'==' if same_ids else '!='} "
              f"scores {'==' if same_sc else '!='}")
        if kw.get("early_stopping"):
            assert same_ids and same_sc, kw
        verdicts.append(same_ids and same_sc)
    return verdicts


def hf_crosscheck(cfg, sd, batch, tag):
    from transformers import BartConfig, BartForConditionalGeneration
    hc = BartConfig(vocab_size=cfg.vocab_size, d_model=cfg.d_model, encoder_layers=cfg.encoder_layers,
                    decoder_layers=cfg.decoder_layers, encoder_attention_heads=cfg.encoder_attention_heads,
                    decoder_attention_heads=cfg.decoder_attention_heads, encoder_ffn_dim=cfg.encoder_ffn_dim,
                    decoder_ffn_dim=cfg.decoder_ffn_dim, max_position_embeddings=cfg.max_position_embeddings,
                    dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, activation_function="gelu",
                    scale_embedding=False, pad_token_id=1, bos_token_id=0, eos_token_id=2,
                    decoder_start_token_id=0, attn_implementation="eager")
    hf = BartForConditionalGeneration(hc).eval()
    hsd = {k: v for k, v in sd.items() if "embed_images" not in k}
    hsd["model.encoder.embed_tokens.weight"] = sd["model.shared.weight"]
    hsd["model.decoder.embed_tokens.weight"] = sd["model.shared.weight"]
    hsd["lm_head.weight"] = sd["model.shared.weight"]
    missing, unexpected = hf.load_state_dict(hsd, strict=False)
    assert not unexpected, unexpected
    assert all("embed_images" not in m for m in missing), missing

    # oracle side
    osd = {k: v.clone().requires_grad_(k != "final_logits_bias") for k, v in sd.items()}
    loss, logits, enc = O.forward(osd, cfg, batch["input_ids"], batch["image_features"], batch["attention_mask"],
                                  batch["decoder_input_ids"], batch["decoder_attention_mask"], batch["labels"])
    loss.backward()

    # HF side: the multimodal embedding (KM-BART specific) is the reference-style nn.Linear + nn.Embedding + masked
    # assignment above, NOT the oracle's function; everything after it (positions, LN, 6+6 layers, tied head, CE) is
    # transformers' own code.  The oracle's embedding must equal it exactly (same fp32 operations).
    ref_emb = reference_style_multimodal_embedding(sd, cfg, batch["input_ids"], batch["image_features"])
    own_emb = O.embed_multi_modal(sd, cfg, batch["input_ids"], batch["image_features"])
    assert rel(own_emb, ref_emb) < 1e-6, rel(own_emb, ref_emb)
    emb = ref_emb.detach().requires_grad_(True)
    out = hf(inputs_embeds=emb, attention_mask=batch["attention_mask"],
             decoder_input_ids=batch["decoder_input_ids"],
             decoder_attention_mask=batch["decoder_attention_mask"], labels=batch["labels"])
    out.loss.backward()
    valid = batch["decoder_attention_mask"].bool()
    errs = {"loss": abs(float(out.loss) - float(loss)) / abs(float(out.loss)),
            "logits": rel(logits[valid], out.logits[valid]),
            "enc": rel(enc[batch["attention_mask"].bool()], out.encoder_last_hidden_state[batch["attention_mask"].bool()])}
    hp = dict(hf.named_parameters())
    for k, v in osd.items():
        if k in hp and hp[k].grad is not None and k != "model.shared.weight":
            errs["grad:" + k] = rel(v.grad, hp[k].grad)
    # tied matrix: HF grad lacks the encoder-side token-embedding contribution (inputs_embeds path);
    # add it back from d(loss)/d(inputs_embeds) scattered over non-image rows.
    mask = (batch["input_ids"] == cfg.img_feat_id) | (batch["input_ids"] == cfg.cls_token_id)
    g_sh = hp["model.shared.weight"].grad.clone()
    g_sh.index_add_(0, batch["input_ids"][~mask], emb.grad[~mask])
    errs["grad:model.shared.weight"] = rel(osd["model.shared.weight"].grad, g_sh)
    worst = max(errs.values())
    print(f"[crosscheck {tag}] worst rel err {worst:.3e} over {len(errs)} tensors "
          f"(loss {float(loss):.6f} vs hf {float(out.loss):.6f})")
    assert worst < 2e-5, {k: v for k, v in errs.items() if v >= 2e-5}
    return worst


def reference_harness(cfg, sd, batches):
    """Runs the REFERENCE's fine_tune (imported from /root/reference) on the oracle model."""
    ref = "/root/reference"
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "src" or k.startswith("src.")}
    sys.path.insert(0, ref)
    try:
        training = importlib.import_module("src.training")
        model = O.OracleModel(cfg, state_dict=sd)
        opt = O.HFAdamW(model.parameters(), lr=1e-3)
        losses = []

        class Log:
            def info(self, msg, pad=False):
                if "Loss:" in msg:
                    losses.append(float(msg.split("Loss:")[1].split(",")[0]))

        args = types.SimpleNamespace(amp=False, epochs=1)
        training.fine_tune(0, model, batches, opt, torch.device("cpu"), args, logger=Log())
        return losses, model
    finally:
        sys.path.remove(ref)
        for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def own_loop(cfg, sd, batches, lr=1e-3):
    model = O.OracleModel(cfg, state_dict=sd).train()
    opt = O.HFAdamW(model.parameters(), lr=lr)
    losses = []
    for b in batches:
        loss = model(b["input_ids"], b["image_features"], b["attention_mask"],
                     decoder_input_ids=b["decoder_input_ids"],
                     decoder_attention_mask=b["decoder_attention_mask"], labels=b["labels"])[0]
        losses.append(float(loss))
        opt.zero_grad()
        loss.backward()
        opt.step()
    return losses, model


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    cfg = G.tiny_config()
    sd = G.golden_state_dict(cfg)
    batch = tiny_batch()

    # 1. independent cross-check
    hf_crosscheck(cfg, sd, batch, "tiny-ragged")
    base = json.load(open("/root/reference/config/vcg_base.json"))
    bcfg = O.OracleConfig.from_dict({**base, "dropout": 0.0})
    bsd = O.init_state_dict(bcfg, seed=0)
    bb = make_batch(2, seed=1234)
    hf_crosscheck(bcfg, bsd, bb, "vcg_base-b2")

    # 2. the reference's own training loop on the oracle model
    batches = [tiny_batch(seed=5), tiny_batch(regions=(4, 0), event_lens=(10, 9), label_lens=(9, 12), seed=6),
               tiny_batch(regions=(6, 6), event_lens=(13, 13), label_lens=(12, 12), seed=7)]
    ref_losses, ref_model = reference_harness(cfg, sd, batches)
    own_losses, own_model = own_loop(cfg, sd, batches)
    print("[harness] reference fine_tune losses", ref_losses)
    print("[harness] oracle step loop losses   ", own_losses)
    assert all(abs(a - round(b, 4)) <= 1e-4 for a, b in zip(ref_losses, own_losses))
    for p, q in zip(ref_model.params, own_model.params):
        assert torch.equal(p, q), "fine_tune order != oracle loop order"

    # 3. fixtures
    osd = {k: v.clone().requires_grad_(k != "final_logits_bias") for k, v in sd.items()}
    loss, logits, enc = O.forward(osd, cfg, batch["input_ids"], batch["image_features"], batch["attention_mask"],
                                  batch["decoder_input_ids"], batch["decoder_attention_mask"], batch["labels"])
    loss.backward()
    names = [k for k in osd if k != "final_logits_bias"]
    fx = {
        "input_ids": batch["input_ids"].numpy(), "attention_mask": batch["attention_mask"].numpy(),
        "decoder_input_ids": batch["decoder_input_ids"].numpy(),
        "decoder_attention_mask": batch["decoder_attention_mask"].numpy(), "labels": batch["labels"].numpy(),
        "regions": np.array([len(f) for f in batch["image_features"]]),
        "loss": np.float32(float(loss)), "logits": logits.detach().numpy(), "encoder_out": enc.detach().numpy(),
        "grad_norms": np.array([float(osd[n].grad.norm()) for n in names], dtype=np.float64),
        "grad_img_bias": osd["model.encoder.embed_images.linear.bias"].grad.numpy(),
        "grad_dec_l1_fc2_bias": osd["model.decoder.layers.1.fc2.bias"].grad.numpy(),
        "grad_shared_rows": osd["model.shared.weight"].grad[[0, 2, int(batch["decoder_input_ids"][0, 1])]].numpy(),
        "step_losses": np.array(own_losses, dtype=np.float64),
        "step_param_sums": np.array([float(p.double().sum()) for p in own_model.params], dtype=np.float64),
    }
    for i, b in enumerate(batches):
        for k in ("input_ids", "attention_mask", "decoder_input_ids", "decoder_attention_mask", "labels"):
            fx[f"step{i}_{k}"] = b[k].numpy()
        fx[f"step{i}_regions"] = np.array([len(f) for f in b["image_features"]])
    np.savez_compressed(os.path.join(GOLD, "tiny_train.npz"), **fx)

    # generation: the tiny model TRAINED on the reverse-copy task (step 3): input-dependent, peaked distributions
    gcfg = G.tiny_config()
    gsd = train_generation_model()
    np.savez_compressed(os.path.join(GOLD, "tiny_trained_fp16.npz"),
                        **{k: v.half().numpy() for k, v in gsd.items()})
    assert all(torch.equal(v, G.trained_state_dict()[k]) for k, v in gsd.items())
    gb = copy_task_batch(5, 6)
    cases = (dict(num_beams=1, max_length=12),
             dict(num_beams=4, num_return_sequences=2, max_length=12, early_stopping=True),
             dict(num_beams=5, num_return_sequences=1, max_length=20, early_stopping=True),
             dict(num_beams=3, num_return_sequences=2, max_length=10, min_length=6, early_stopping=True),
             dict(num_beams=3, num_return_sequences=3, max_length=10, early_stopping=False, length_penalty=2.0))
    verdicts = hf_generate_crosscheck(gcfg, gsd, gb, cases)
    gen = {"seed": 5, "batch": 6, "input_ids": gb["input_ids"].tolist(), "attention_mask": gb["attention_mask"].tolist(),
           "regions": [len(f) for f in gb["image_features"]], "cases": []}
    for kw, v in zip(cases, verdicts):
        r = O.generate(gsd, gcfg, gb["input_ids"], gb["image_features"], gb["attention_mask"],
                       return_scores=kw.get("num_beams", 1) > 1, **kw)
        if isinstance(r, tuple):
            gen["cases"].append({"kwargs": kw, "ids": r[0].tolist(), "scores": [float(x) for x in r[1]],
                                 "identical_to_transformers_5_15": v})
        else:
            gen["cases"].append({"kwargs": kw, "ids": r.tolist()})
        print("[generate]", kw, gen["cases"][-1]["ids"])
    json.dump(gen, open(os.path.join(GOLD, "tiny_generate.json"), "w"))
    print("wrote", os.listdir(GOLD))


if __name__ == "__main__":
    main()
