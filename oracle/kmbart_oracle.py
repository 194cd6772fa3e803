"""CPU oracle for the KM-BART hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 restatement of the arithmetic the reference
(fomalhautb/KM-BART) runs for one training step and for generation.  It is the
checker the HIP path is compared against; it is never the thing shipped or
measured.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  The product (`km-bart_amd/`) must never import it.

Pinning status
--------------
The reference holds no tests and no golden vectors and its
arithmetic lives in the un-vendored dependency `transformers==3.0.2`
(reference `environment.yaml:159`), which cannot be imported in the build
container.  The oracle is therefore pinned two ways, both run by
`oracle/make_golden.py` in the build container:
  * forward/backward (encoder, decoder, tied head, CE): against the installed
    transformers 5.15 `BartForConditionalGeneration` on a shared state-dict
    (same architecture and parameter names) to <= 1e-5;
  * the training harness: by driving this model through the REFERENCE's own
    `src.training.fine_tune` (importable here) and committing the loss sequence.
  * the KM-BART-specific multimodal embedding: against a plain nn.Linear + nn.Embedding +
    masked index assignment written from reference src/model/modules.py:24-41,89-102
    (`make_golden.reference_style_multimodal_embedding`), which also feeds the transformers
    side of the cross-check (the oracle's own function no longer does);
  * generation: the beam search (HF 3.0.2 `_generate_beam_search` restated from its call
    sites in `src/model/mixins.py:33-434`) against transformers 5.15 `generate()` on a tiny
    model trained to a non-degenerate task: token ids AND length-normalised scores are
    identical for every early_stopping=True case (the reference's setting,
    src/generation.py:22-32), min_length included.  early_stopping=False differs by a
    documented algorithm change (4.x bounds the attainable score with max_length).
What stays **parity unpinned**: everything is pinned to transformers 5.15 and to the
reference's importable modules, not to transformers 3.0.2 itself (absent offline).  The
multinomial-sampling branch of beam search (do_sample with num_beams > 1) is pinned where it
can be made independent of the random stream: with top_k = 2 the 2 * num_beams draws without
replacement take every non-zero entry, and ids + scores are identical to transformers 5.15
`generate(do_sample=True, num_beams=k, top_k=2)` (oracle/make_golden_beam_sample.py, round 6);
arbitrary draws are not reproducible across implementations and stay compared through a
shared sampler only.

Reference map (file:line into /root/reference)
----------------------------------------------
  init_state_dict ............ src/model/model.py:27-37 (+ HF init_weights, std=init_std)
  image_embedding ............ src/model/modules.py:19-41
  embed_multi_modal .......... src/model/modules.py:89-102
  encoder_forward ............ src/model/modules.py:104-165 (+ HF3.0.2 EncoderLayer)
  decoder_forward ............ src/model/model.py:87-97     (+ HF3.0.2 BartDecoder/DecoderLayer)
  prepare_decoder_masks ...... src/model/model.py:63-70     (+ HF3.0.2 _prepare_bart_decoder_inputs)
  forward (logits, loss) ..... src/model/model.py:325-405
  HFAdamW .................... vcg_train.py:13,100 (transformers.AdamW 3.0.2 defaults)
  generate ................... src/model/mixins.py:33-434   (+ HF3.0.2 _generate_*_search)
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

NEG_INF = float("-inf")

DEFAULTS = dict(
    # src/model/config.py:4-47 defaults that config/vcg_base.json does not carry
    activation_dropout=0.0, extra_pos_embeddings=2, activation_function="gelu",
    vocab_size=50320, image_feature_size=2052, d_model=1024,
    encoder_ffn_dim=4096, encoder_layers=12, encoder_attention_heads=16,
    decoder_ffn_dim=4096, decoder_layers=12, decoder_attention_heads=16,
    attention_dropout=0.0, dropout=0.1, max_position_embeddings=1024,
    init_std=0.02, pad_token_id=1, bos_token_id=0, eos_token_id=2,
    img_feat_id=50273, cls_token_id=50276, scale_embedding=False,
    decoder_start_token_id=0, num_labels=0, num_attributes=0, num_relations=0,
    lm_loss_factor=1.0, mrm_loss_factor=1.0, attribute_loss_factor=1.0, relation_loss_factor=1.0,
    # transformers 3.0.2 PretrainedConfig generation defaults
    max_length=20, min_length=0, num_beams=1, length_penalty=1.0,
    early_stopping=False, num_return_sequences=1,
)


class OracleConfig:
    def __init__(self, **kw):
        d = dict(DEFAULTS)
        d.update(kw)
        for k, v in d.items():
            setattr(self, k, v)

    @classmethod
    def from_dict(cls, d):
        return cls(**d)


# --------------------------------------------------------------------------- #
# parameters
# --------------------------------------------------------------------------- #
def param_names(cfg):
    """State-dict keys (HF BART names + embed_images), one entry per distinct tensor."""
    names = ["model.shared.weight",
             "model.encoder.embed_images.linear.weight",
             "model.encoder.embed_images.linear.bias",
             "model.encoder.embed_positions.weight",
             "model.encoder.layernorm_embedding.weight",
             "model.encoder.layernorm_embedding.bias"]
    for i in range(cfg.encoder_layers):
        p = f"model.encoder.layers.{i}."
        for a in ("q_proj", "k_proj", "v_proj", "out_proj"):
            names += [p + f"self_attn.{a}.weight", p + f"self_attn.{a}.bias"]
        names += [p + "self_attn_layer_norm.weight", p + "self_attn_layer_norm.bias",
                  p + "fc1.weight", p + "fc1.bias", p + "fc2.weight", p + "fc2.bias",
                  p + "final_layer_norm.weight", p + "final_layer_norm.bias"]
    names += ["model.decoder.embed_positions.weight",
              "model.decoder.layernorm_embedding.weight",
              "model.decoder.layernorm_embedding.bias"]
    for i in range(cfg.decoder_layers):
        p = f"model.decoder.layers.{i}."
        for blk in ("self_attn", "encoder_attn"):
            for a in ("q_proj", "k_proj", "v_proj", "out_proj"):
                names += [p + f"{blk}.{a}.weight", p + f"{blk}.{a}.bias"]
            names += [p + f"{blk}_layer_norm.weight", p + f"{blk}_layer_norm.bias"]
        names += [p + "fc1.weight", p + "fc1.bias", p + "fc2.weight", p + "fc2.bias",
                  p + "final_layer_norm.weight", p + "final_layer_norm.bias"]
    return names


HEADS = (("mrm_head", "num_labels", 1), ("attribute_head", "num_attributes", 1), ("relation_head", "num_relations", 2))


def head_param_names(cfg):
    """BartClassificationHead parameters of MultiModalBartForPreTraining (src/model/model.py:133-158)."""
    out = []
    for name, attr, _ in HEADS:
        if getattr(cfg, attr, 0) > 0:
            out += [f"{name}.dense.weight", f"{name}.dense.bias", f"{name}.out_proj.weight", f"{name}.out_proj.bias"]
    return out


def param_shape(cfg, name):
    d = cfg.d_model
    for hname, attr, mult in HEADS:
        if name.startswith(hname + "."):
            C = getattr(cfg, attr)
            return {"dense.weight": (d, mult * d), "dense.bias": (d,), "out_proj.weight": (C, d),
                    "out_proj.bias": (C,)}[name[len(hname) + 1:]]
    if name == "model.shared.weight":
        return (cfg.vocab_size, d)
    if name.endswith("embed_images.linear.weight"):
        return (d, cfg.image_feature_size)
    if name.endswith("embed_positions.weight"):
        return (cfg.max_position_embeddings + cfg.extra_pos_embeddings, d)
    ffn = cfg.encoder_ffn_dim if ".encoder." in name else cfg.decoder_ffn_dim
    if name.endswith("fc1.weight"):
        return (ffn, d)
    if name.endswith("fc1.bias"):
        return (ffn,)
    if name.endswith("fc2.weight"):
        return (d, ffn)
    if name.endswith("proj.weight"):
        return (d, d)
    return (d,)  # biases, LayerNorm weight/bias


def init_state_dict(cfg, seed=0):
    """HF `init_weights`: Linear/Embedding ~ N(0, init_std), biases 0, padding rows 0,
    LayerNorm (1, 0).  `final_logits_bias` is a zero buffer (model.py:323)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for n in param_names(cfg):
        shp = param_shape(cfg, n)
        if "layer_norm" in n or "layernorm" in n:
            t = torch.ones(shp) if n.endswith("weight") else torch.zeros(shp)
        elif n.endswith("bias"):
            t = torch.zeros(shp)
        else:
            t = torch.randn(shp, generator=g) * cfg.init_std
            if n == "model.shared.weight" or n.endswith("embed_positions.weight"):
                t[cfg.pad_token_id].zero_()
        sd[n] = t
    sd["final_logits_bias"] = torch.zeros(1, cfg.vocab_size)
    return sd


# --------------------------------------------------------------------------- #
# model
# --------------------------------------------------------------------------- #
def _ln(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def _lin(x, sd, prefix):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def attention(sd, prefix, cfg, n_heads, query, key, key_padding_mask=None, causal_mask=None,
              cache=None, static_kv=False, p_attn_drop=0.0, training=False, probs_out=None):
    """HF3.0.2 `SelfAttention.forward`, batch-major.  query [B,Tq,D], key [B,Tk,D].
    q is scaled by head_dim**-0.5 BEFORE q.k^T; key padding via masked_fill(-inf).
    cache: dict with prev_key/prev_value [B,H,t,hd] (and prev_key_padding_mask)."""
    B, Tq, D = query.shape
    hd = D // n_heads
    q = _lin(query, sd, prefix + ".q_proj") * (hd ** -0.5)

    def split(t):
        return t.view(B, -1, n_heads, hd).transpose(1, 2)  # [B,H,T,hd]

    if cache is not None and static_kv and "prev_key" in cache:
        k, v = cache["prev_key"], cache["prev_value"]
    else:
        k = split(_lin(key, sd, prefix + ".k_proj"))
        v = split(_lin(key, sd, prefix + ".v_proj"))
        if cache is not None and not static_kv and "prev_key" in cache:
            k = torch.cat([cache["prev_key"], k], dim=2)
            v = torch.cat([cache["prev_value"], v], dim=2)
    if cache is not None:
        cache["prev_key"], cache["prev_value"] = k, v
    q = split(q)
    w = torch.matmul(q, k.transpose(-1, -2))  # [B,H,Tq,Tk]
    if causal_mask is not None:
        w = w + causal_mask
    if key_padding_mask is not None:  # True = pad
        w = w.masked_fill(key_padding_mask[:, None, None, :], NEG_INF)
    w = F.softmax(w, dim=-1)
    if probs_out is not None:   # output_attentions: HF3.0.2 returns the weights [B, H, Tq, Tk] (before attention dropout)
        probs_out.append(w)
    w = F.dropout(w, p=p_attn_drop, training=training)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, Tq, D)
    return _lin(o, sd, prefix + ".out_proj")


def image_embedding(sd, image_features):
    """src/model/modules.py:24-41: cat non-empty -> Linear(2052->d) -> split back."""
    lens = [len(x) for x in image_features]
    non_empty = [x for x in image_features if len(x) > 0]
    out, idx = [], 0
    emb = _lin(torch.cat(non_empty, 0), sd, "model.encoder.embed_images.linear") if non_empty else None
    for l in lens:
        out.append(emb[idx: idx + l] if l > 0 else torch.empty(0))
        idx += l
    return out


def embed_multi_modal(sd, cfg, input_ids, image_features):
    """src/model/modules.py:89-102: rows where id is <img_feat>/<cls> are replaced, in order,
    by the projected region features (count must equal R_i)."""
    mask = (input_ids == cfg.img_feat_id) | (input_ids == cfg.cls_token_id)
    emb_img = image_embedding(sd, image_features)
    embedded = F.embedding(input_ids, sd["model.shared.weight"])
    rows = []
    for i, value in enumerate(emb_img):
        row = embedded[i]
        if len(value) > 0:
            if int(mask[i].sum()) != len(value):
                raise RuntimeError("number of <img_feat>/<cls> ids != number of region features")
            row = row.index_put((mask[i].nonzero(as_tuple=True)[0],), value)
        rows.append(row)
    return torch.stack(rows, 0)


def _ffn_block(sd, p, cfg, x, drop, training):
    r = x
    h = F.gelu(_lin(x, sd, p + "fc1"))
    h = F.dropout(h, p=cfg.activation_dropout, training=training)
    h = F.dropout(_lin(h, sd, p + "fc2"), p=drop, training=training)
    return _ln(r + h, sd, p + "final_layer_norm")


def encoder_forward(sd, cfg, input_ids, image_features, attention_mask=None, training=False, taps=None):
    """src/model/modules.py:104-165 (batch-major; HF runs time-major, same arithmetic).  taps: optional dict that receives
    'hidden' (output_hidden_states: every layer's input and the final output, modules.py:143-160) and 'attn'
    (output_attentions: every layer's self-attention weights)."""
    hidden = taps.setdefault("hidden", []) if taps is not None else None
    attn = taps.setdefault("attn", []) if taps is not None else None
    pad = attention_mask.eq(0) if attention_mask is not None else None
    scale = math.sqrt(cfg.d_model) if cfg.scale_embedding else 1.0
    S = input_ids.shape[1]
    pos = sd["model.encoder.embed_positions.weight"][torch.arange(S) + cfg.extra_pos_embeddings]
    x = embed_multi_modal(sd, cfg, input_ids, image_features) * scale + pos
    x = _ln(x, sd, "model.encoder.layernorm_embedding")
    x = F.dropout(x, p=cfg.dropout, training=training)
    for i in range(cfg.encoder_layers):
        p = f"model.encoder.layers.{i}."
        if hidden is not None:
            hidden.append(x)
        a = attention(sd, p + "self_attn", cfg, cfg.encoder_attention_heads, x, x, key_padding_mask=pad,
                      p_attn_drop=cfg.attention_dropout, training=training, probs_out=attn)
        x = _ln(x + F.dropout(a, p=cfg.dropout, training=training), sd, p + "self_attn_layer_norm")
        x = _ffn_block(sd, p, cfg, x, cfg.dropout, training)
    if hidden is not None:
        hidden.append(x)
    return x


def prepare_decoder_masks(cfg, decoder_input_ids, decoder_attention_mask):
    """HF3.0.2 `_prepare_bart_decoder_inputs`: pad mask (True = pad, None if no pad) + triu(-inf,1)."""
    T = decoder_input_ids.shape[1]
    if decoder_attention_mask is None:
        pm = decoder_input_ids.eq(cfg.pad_token_id)
        pm = pm if bool(pm.any()) else None
    else:
        pm = decoder_attention_mask.eq(0)
    causal = torch.triu(torch.full((T, T), NEG_INF), 1)
    return pm, causal


def decoder_forward(sd, cfg, decoder_input_ids, enc_out, enc_attention_mask, dec_pad_mask, causal_mask,
                    cache=None, use_cache=False, training=False, taps=None):
    """HF3.0.2 `BartDecoder.forward`.  With use_cache only the last token is embedded, at
    learned position (len-1)+2, and per-layer self K/V are appended; cross K/V are reused."""
    enc_pad = enc_attention_mask.eq(0) if enc_attention_mask is not None else None
    scale = math.sqrt(cfg.d_model) if cfg.scale_embedding else 1.0
    T = decoder_input_ids.shape[1]
    P = sd["model.decoder.embed_positions.weight"]
    if use_cache:
        pos = P[torch.tensor([T - 1]) + cfg.extra_pos_embeddings]
        ids = decoder_input_ids[:, -1:]
    else:
        pos = P[torch.arange(T) + cfg.extra_pos_embeddings]
        ids = decoder_input_ids
    x = F.embedding(ids, sd["model.shared.weight"]) * scale + pos
    x = _ln(x, sd, "model.decoder.layernorm_embedding")
    x = F.dropout(x, p=cfg.dropout, training=training)
    if use_cache and cache is None:
        cache = [dict(self={}, encoder_decoder={}) for _ in range(cfg.decoder_layers)]
    # taps (HF3.0.2 BartDecoder): 'hidden' = every layer's INPUT (no final entry), 'attn' = the SELF-attention weights
    hidden = taps.setdefault("hidden", []) if taps is not None else None
    attn = taps.setdefault("attn", []) if taps is not None else None
    for i in range(cfg.decoder_layers):
        p = f"model.decoder.layers.{i}."
        lc = cache[i] if use_cache else None
        if hidden is not None:
            hidden.append(x)
        a = attention(sd, p + "self_attn", cfg, cfg.decoder_attention_heads, x, x,
                      key_padding_mask=dec_pad_mask, causal_mask=causal_mask,
                      cache=lc["self"] if lc is not None else None,
                      p_attn_drop=cfg.attention_dropout, training=training, probs_out=attn)
        x = _ln(x + F.dropout(a, p=cfg.dropout, training=training), sd, p + "self_attn_layer_norm")
        a = attention(sd, p + "encoder_attn", cfg, cfg.decoder_attention_heads, x, enc_out,
                      key_padding_mask=enc_pad,
                      cache=lc["encoder_decoder"] if lc is not None else None, static_kv=True,
                      p_attn_drop=cfg.attention_dropout, training=training)
        x = _ln(x + F.dropout(a, p=cfg.dropout, training=training), sd, p + "encoder_attn_layer_norm")
        x = _ffn_block(sd, p, cfg, x, cfg.dropout, training)
    return x, cache


def forward(sd, cfg, input_ids, image_features, attention_mask=None, decoder_input_ids=None,
            decoder_attention_mask=None, labels=None, training=False, encoder_out=None):
    """src/model/model.py:325-405 training/teacher-forced branch.
    Returns (loss or None, lm_logits [B,T,V], encoder_last_hidden [B,S,D])."""
    if encoder_out is None:
        encoder_out = encoder_forward(sd, cfg, input_ids, image_features, attention_mask, training)
    pm, causal = prepare_decoder_masks(cfg, decoder_input_ids, decoder_attention_mask)
    h, _ = decoder_forward(sd, cfg, decoder_input_ids, encoder_out, attention_mask, pm, causal,
                           training=training)
    logits = F.linear(h, sd["model.shared.weight"], sd["final_logits_bias"])
    loss = None
    if labels is not None:
        loss = F.cross_entropy(logits.view(-1, cfg.vocab_size), labels.view(-1))  # ignore_index=-100, mean
    return loss, logits, encoder_out


def classification_head(sd, name, x):
    """HF3.0.2 BartClassificationHead with classif_dropout = 0: out_proj(tanh(dense(x)))."""
    return _lin(torch.tanh(_lin(x, sd, name + ".dense")), sd, name + ".out_proj")


def pretrain_forward(sd, cfg, input_ids, image_features, attention_mask, decoder_input_ids, decoder_attention_mask,
                     labels=None, mrm_labels=None, mrm_mask=None, attribute_labels=None, attribute_mask=None,
                     relation_labels=None, training=False):
    """MultiModalBartForPreTraining.forward (src/model/model.py:162-309): LM loss with <cls> labels ignored,
    KL-div(batchmean) masked-region modelling, attribute CE, relation CE on cat(object, subject) rows, weighted sum.
    Returns (losses dict, lm_logits)."""
    enc = encoder_forward(sd, cfg, input_ids, image_features, attention_mask, training)
    pm, causal = prepare_decoder_masks(cfg, decoder_input_ids, decoder_attention_mask)
    h, _ = decoder_forward(sd, cfg, decoder_input_ids, enc, attention_mask, pm, causal, training=training)
    losses = {}
    mrm_loss = attribute_loss = relation_loss = lm_loss = 0
    if mrm_labels is not None:
        rep = h[mrm_mask.bool()]
        if len(rep) > 0:
            pred = F.log_softmax(classification_head(sd, "mrm_head", rep), dim=1)
            mrm_loss = F.kl_div(pred, torch.cat(mrm_labels, 0), reduction="batchmean") * cfg.mrm_loss_factor
            losses["mrm_loss"] = mrm_loss
    if attribute_labels is not None:
        rep = h[attribute_mask.bool()]
        if len(rep) > 0:
            pred = classification_head(sd, "attribute_head", rep)
            attribute_loss = F.cross_entropy(pred, torch.cat(attribute_labels, 0).reshape(-1)) * cfg.attribute_loss_factor
            losses["attribute_loss"] = attribute_loss
    if relation_labels is not None:
        reps, ids = [], []
        for i, rels in enumerate(relation_labels):
            for rel in rels:
                reps.append(torch.cat([h[i][rel["object_index"]], h[i][rel["subject_index"]]], 0))
                ids.append(rel["label"])
        if reps:
            pred = classification_head(sd, "relation_head", torch.stack(reps))
            relation_loss = F.cross_entropy(pred, torch.tensor(ids)) * cfg.relation_loss_factor
            losses["relation_loss"] = relation_loss
    logits = F.linear(h, sd["model.shared.weight"], sd["final_logits_bias"])
    if labels is not None:
        lab = labels.clone()
        lab[lab == cfg.cls_token_id] = -100
        lm_loss = F.cross_entropy(logits.view(-1, cfg.vocab_size), lab.reshape(-1)) * cfg.lm_loss_factor
        losses["lm_loss"] = lm_loss
    losses["loss"] = lm_loss + mrm_loss + attribute_loss + relation_loss
    return losses, logits


# --------------------------------------------------------------------------- #
# nn.Module wrapper with the reference's call surface (used with the REFERENCE's
# src.training.fine_tune to pin the harness, and by the CPU baseline)
# --------------------------------------------------------------------------- #
class OracleModel(torch.nn.Module):
    def __init__(self, cfg, seed=0, state_dict=None):
        super().__init__()
        self.config = cfg
        sd = state_dict if state_dict is not None else init_state_dict(cfg, seed)
        self._names = [n for n in sd if n != "final_logits_bias"]
        self.params = torch.nn.ParameterList([torch.nn.Parameter(sd[n].clone().float()) for n in self._names])
        self.register_buffer("final_logits_bias", sd["final_logits_bias"].clone().float())

    def sd(self):
        d = {n: p for n, p in zip(self._names, self.params)}
        d["final_logits_bias"] = self.final_logits_bias
        return d

    def forward(self, input_ids, image_features, attention_mask=None, encoder_outputs=None,
                decoder_input_ids=None, decoder_attention_mask=None, labels=None, **unused):
        loss, logits, enc = forward(self.sd(), self.config, input_ids, image_features, attention_mask,
                                    decoder_input_ids, decoder_attention_mask, labels, self.training)
        return (logits, enc) if loss is None else (loss, logits, enc)

    @torch.no_grad()
    def generate(self, input_ids, image_features=None, attention_mask=None, **kw):
        return generate(self.sd(), self.config, input_ids, image_features, attention_mask, **kw)


class HFAdamW(torch.optim.Optimizer):
    """transformers 3.0.2 `AdamW` (vcg_train.py:100): eps added OUTSIDE the bias correction,
    decoupled weight decay applied after the update.  Defaults lr 1e-3, betas (0.9, 0.999),
    eps 1e-6, weight_decay 0.0, correct_bias True."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      correct_bias=correct_bias))

    @torch.no_grad()
    def step(self):
        for g in self.param_groups:
            b1, b2 = g["betas"]
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                st["exp_avg"].mul_(b1).add_(p.grad, alpha=1.0 - b1)
                st["exp_avg_sq"].mul_(b2).addcmul_(p.grad, p.grad, value=1.0 - b2)
                denom = st["exp_avg_sq"].sqrt().add_(g["eps"])
                step_size = g["lr"]
                if g["correct_bias"]:
                    step_size = step_size * math.sqrt(1.0 - b2 ** st["step"]) / (1.0 - b1 ** st["step"])
                p.addcdiv_(st["exp_avg"], denom, value=-step_size)
                if g["weight_decay"] > 0.0:
                    p.add_(p, alpha=-g["lr"] * g["weight_decay"])


# --------------------------------------------------------------------------- #
# generation (HF3.0.2 generation_utils as reached from src/model/mixins.py:33-434)
# --------------------------------------------------------------------------- #
class BeamHypotheses:
    """HF3.0.2 `BeamHypotheses`: n-best list, score = sum_logprobs / len(hyp)**length_penalty."""

    def __init__(self, num_beams, max_length, length_penalty, early_stopping):
        self.max_length = max_length - 1
        self.length_penalty = length_penalty
        self.early_stopping = early_stopping
        self.num_beams = num_beams
        self.beams = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / len(hyp) ** self.length_penalty
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                sorted_scores = sorted([(s, idx) for idx, (s, _) in enumerate(self.beams)])
                del self.beams[sorted_scores[0][1]]
                self.worst_score = sorted_scores[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self) < self.num_beams:
            return False
        if self.early_stopping:
            return True
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


def _decode_logits(sd, cfg, dec_ids, enc_out, attention_mask, cache):
    """One cached decoder step -> logits of the last position (model.py:384-397 use_cache branch:
    no causal / decoder padding masks, model.py:71-72)."""
    h, cache = decoder_forward(sd, cfg, dec_ids, enc_out, attention_mask, None, None, cache=cache, use_cache=True)
    logits = F.linear(h, sd["model.shared.weight"], sd["final_logits_bias"])
    return logits[:, -1, :], cache


def _reorder_cache(cache, beam_idx):
    """mixins.py:419-434: index_select(0, beam_idx) on every cached tensor."""
    for lc in cache:
        for blk in lc.values():
            for k in list(blk.keys()):
                if blk[k] is not None:
                    blk[k] = blk[k].index_select(0, beam_idx)
    return cache


def top_k_top_p_filtering(logits, top_k=0, top_p=1.0, filter_value=NEG_INF, min_tokens_to_keep=1):
    """HF3.0.2 `top_k_top_p_filtering` (generation_utils): keep the top_k largest, then the smallest prefix of the
    sorted distribution whose cumulative probability exceeds top_p (at least min_tokens_to_keep)."""
    if top_k > 0:
        top_k = min(max(top_k, min_tokens_to_keep), logits.size(-1))
        logits = logits.masked_fill(logits < torch.topk(logits, top_k)[0][..., -1, None], filter_value)
    if top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        cum = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        rm = cum > top_p
        if min_tokens_to_keep > 1:
            rm[..., :min_tokens_to_keep] = False
        rm[..., 1:] = rm[..., :-1].clone()
        rm[..., 0] = False
        logits = logits.masked_fill(rm.scatter(1, sorted_indices, rm), filter_value)
    return logits


@torch.no_grad()
def calc_banned_ngram_tokens(prev_ids, no_repeat_ngram_size, cur_len):
    """HF3.0.2 generation_utils.calc_banned_ngram_tokens: per row, the tokens that would complete an n-gram that already
    occurs in the row."""
    if cur_len + 1 < no_repeat_ngram_size:
        return [[] for _ in prev_ids]
    out = []
    for row in prev_ids:
        seen = {}
        for ng in zip(*[row[i:] for i in range(no_repeat_ngram_size)]):
            seen.setdefault(tuple(ng[:-1]), []).append(ng[-1])
        out.append(seen.get(tuple(row[cur_len + 1 - no_repeat_ngram_size:cur_len]), []))
    return out


def calc_banned_bad_words_ids(prev_ids, bad_words_ids):
    """HF3.0.2 generation_utils.calc_banned_bad_words_ids, including its length test against the NUMBER OF ROWS
    (`len(tokens) > len(prev_input_ids)`, a quirk of that release)."""
    n_rows = len(prev_ids)
    out = []
    for row in prev_ids:
        banned = []
        for seq in bad_words_ids:
            assert len(seq) > 0, "Banned words token sequences {} cannot have an empty list".format(bad_words_ids)
            head = list(seq[:-1])
            if len(head) == 0:
                match = True
            elif len(head) > n_rows:
                match = False
            else:
                match = row[-len(head):] == head
            if match:
                banned.append(seq[-1])
        out.append(banned)
    return out


def postprocess_next_token_scores(scores, prev_ids, cur_len, min_length, eos, repetition_penalty=1.0,
                                  no_repeat_ngram_size=0, bad_words_ids=None):
    """HF3.0.2 GenerationMixin.postprocess_next_token_scores, in its order: repetition penalty (CTRL: a score < 0 is
    multiplied by the penalty, a score >= 0 divided), EOS banned below min_length, no-repeat n-grams, bad words.
    `scores` [rows, V] is modified in place (raw logits in the no-beam loop, log-probabilities in beam search);
    prev_ids: list of the rows' token lists so far."""
    if repetition_penalty != 1.0:
        for i, row in enumerate(prev_ids):
            for tok in set(row):
                if scores[i, tok] < 0:
                    scores[i, tok] *= repetition_penalty
                else:
                    scores[i, tok] /= repetition_penalty
    if eos is not None and cur_len < min_length:
        scores[:, eos] = NEG_INF
    if no_repeat_ngram_size > 0:
        for i, banned in enumerate(calc_banned_ngram_tokens(prev_ids, no_repeat_ngram_size, cur_len)):
            scores[i, banned] = NEG_INF
    if bad_words_ids is not None:
        for i, banned in enumerate(calc_banned_bad_words_ids(prev_ids, bad_words_ids)):
            scores[i, banned] = NEG_INF
    return scores


def generate(sd, cfg, input_ids, image_features, attention_mask=None, max_length=None, min_length=None,
             num_beams=None, num_return_sequences=None, early_stopping=None, length_penalty=None,
             do_sample=False, top_k=0, top_p=1.0, temperature=1.0, return_scores=False, sampler=None,
             repetition_penalty=1.0, no_repeat_ngram_size=0, bad_words_ids=None, **unused):
    """Greedy / sampling without beams (num_beams==1) and beam search with or without multinomial sampling.
    mixins.py:150-384 -> HF3.0.2 _generate_no_beam_search / _generate_beam_search.  `sampler(probs, n)` replaces
    torch.multinomial (tests feed both implementations the same draws)."""
    if sampler is None:
        sampler = lambda probs, n: torch.multinomial(probs, num_samples=n)  # noqa: E731
    max_length = cfg.max_length if max_length is None else max_length
    min_length = cfg.min_length if min_length is None else min_length
    num_beams = cfg.num_beams if num_beams is None else num_beams
    nret = cfg.num_return_sequences if num_return_sequences is None else num_return_sequences
    early_stopping = cfg.early_stopping if early_stopping is None else early_stopping
    length_penalty = cfg.length_penalty if length_penalty is None else length_penalty
    pad, eos, V = cfg.pad_token_id, cfg.eos_token_id, cfg.vocab_size
    B = input_ids.shape[0]
    if attention_mask is None:
        attention_mask = input_ids.ne(pad).long() if bool((input_ids == pad).any()) else torch.ones_like(input_ids)
    if do_sample:   # mixins.py:259-262: sampling replicates the batch, one sequence is returned per replica
        if nret > 1:
            rep_idx = torch.arange(B).repeat_interleave(nret)
            input_ids, attention_mask = input_ids[rep_idx], attention_mask[rep_idx]
            image_features = [image_features[i] for i in rep_idx.tolist()]
            B = B * nret
        nret = 1
    elif num_beams == 1:
        assert nret == 1
    else:
        assert num_beams >= nret
    enc_out = encoder_forward(sd, cfg, input_ids, image_features, attention_mask, False)  # mixins.py:281-283
    if num_beams > 1:
        idx = torch.arange(B).view(-1, 1).repeat(1, num_beams).view(-1)
        enc_out = enc_out.index_select(0, idx)
        attention_mask = attention_mask.index_select(0, idx)
    ids = torch.full((B * num_beams, 1), cfg.decoder_start_token_id, dtype=torch.long)
    cur_len = 1
    assert cur_len < max_length
    cache = None

    if num_beams == 1:  # HF3.0.2 _generate_no_beam_search, greedy branch
        unfinished = torch.ones(B, dtype=torch.long)
        while cur_len < max_length:
            logits, cache = _decode_logits(sd, cfg, ids, enc_out, attention_mask, cache)
            postprocess_next_token_scores(logits, ids.tolist(), cur_len, min_length, eos, repetition_penalty,
                                          no_repeat_ngram_size, bad_words_ids)
            if do_sample:
                lg = logits / temperature if temperature != 1.0 else logits
                lg = top_k_top_p_filtering(lg, top_k=top_k, top_p=top_p)
                nxt = sampler(F.softmax(lg, dim=-1), 1).squeeze(1)
            else:
                nxt = torch.argmax(logits, dim=-1)
            tok = nxt * unfinished + pad * (1 - unfinished)
            ids = torch.cat([ids, tok.unsqueeze(-1)], dim=-1)
            cur_len += 1
            eos_in = tok == eos
            unfinished = unfinished * (~eos_in).long()
            if int(unfinished.max()) == 0:
                break
        return ids

    # HF3.0.2 _generate_beam_search, do_sample=False
    hyps = [BeamHypotheses(num_beams, max_length, length_penalty, early_stopping) for _ in range(B)]
    beam_scores = torch.zeros((B, num_beams), dtype=torch.float)
    if not do_sample:   # greedy beam search starts from beam 0 only
        beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    done = [False] * B
    next_scores = next_tokens = None
    while cur_len < max_length:
        logits, cache = _decode_logits(sd, cfg, ids, enc_out, attention_mask, cache)
        # adjust_logits_during_generation (mixins.py:400-405): force BOS at len 1, EOS at max_length-1 -- only when
        # not sampling (HF3.0.2: `if self.config.is_encoder_decoder and do_sample is False`)
        if cur_len == 1 and not do_sample:
            keep = logits[:, cfg.bos_token_id].clone()
            logits.fill_(NEG_INF)
            logits[:, cfg.bos_token_id] = keep
        if cur_len == max_length - 1 and eos is not None and not do_sample:
            keep = logits[:, eos].clone()
            logits.fill_(NEG_INF)
            logits[:, eos] = keep
        scores = F.log_softmax(logits, dim=-1)
        postprocess_next_token_scores(scores, ids.tolist(), cur_len, min_length, eos, repetition_penalty,
                                      no_repeat_ngram_size, bad_words_ids)   # on the log-probabilities
        if do_sample:
            _scores = scores + beam_scores[:, None]
            if temperature != 1.0:
                _scores = _scores / temperature
            _scores = top_k_top_p_filtering(_scores, top_k=top_k, top_p=top_p, min_tokens_to_keep=2)
            _scores = _scores.contiguous().view(B, num_beams * V)
            next_tokens = sampler(F.softmax(_scores, dim=-1), 2 * num_beams)
            next_scores = torch.gather(_scores, -1, next_tokens)
            next_scores, order = torch.sort(next_scores, descending=True, dim=1)
            next_tokens = torch.gather(next_tokens, -1, order)
        else:
            next_scores = (scores + beam_scores[:, None]).view(B, num_beams * V)
            next_scores, next_tokens = torch.topk(next_scores, 2 * num_beams, dim=1, largest=True, sorted=True)
        next_batch_beam = []
        for b in range(B):
            if done[b]:
                next_batch_beam.extend([(0, pad, 0)] * num_beams)
                continue
            sent = []
            for rank, (tid, tscore) in enumerate(zip(next_tokens[b], next_scores[b])):
                beam_id = int(tid) // V
                token_id = int(tid) % V
                eff = b * num_beams + beam_id
                if eos is not None and token_id == eos:
                    if rank >= num_beams:
                        continue
                    hyps[b].add(ids[eff].clone(), float(tscore))
                else:
                    sent.append((float(tscore), token_id, eff))
                if len(sent) == num_beams:
                    break
            done[b] = done[b] or hyps[b].is_done(float(next_scores[b].max()), cur_len)
            assert len(sent) == num_beams
            next_batch_beam.extend(sent)
        if all(done):
            break
        beam_scores = torch.tensor([x[0] for x in next_batch_beam], dtype=torch.float)
        beam_tokens = torch.tensor([x[1] for x in next_batch_beam], dtype=torch.long)
        beam_idx = torch.tensor([x[2] for x in next_batch_beam], dtype=torch.long)
        ids = torch.cat([ids[beam_idx, :], beam_tokens.unsqueeze(1)], dim=-1)
        cur_len += 1
        cache = _reorder_cache(cache, beam_idx)
    for b in range(B):
        if done[b]:
            continue
        for beam_id in range(num_beams):
            eff = b * num_beams + beam_id
            hyps[b].add(ids[eff], float(beam_scores[eff]))
    best, best_scores, lens = [], [], []
    for h in hyps:
        sh = sorted(h.beams, key=lambda x: x[0])
        for _ in range(nret):
            s, hyp = sh.pop()
            best.append(hyp)
            best_scores.append(s)
            lens.append(len(hyp))
    if min(lens) != max(lens):
        L = min(max(lens) + 1, max_length)
        out = torch.full((len(best), L), pad, dtype=torch.long)
        for i, hyp in enumerate(best):
            out[i, : lens[i]] = hyp
            if lens[i] < max_length:
                out[i, lens[i]] = eos
    else:
        out = torch.stack(best).long()
    return (out, torch.tensor(best_scores)) if return_scores else out
