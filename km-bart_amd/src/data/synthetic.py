"""Synthetic VCG batches: the shape BASELINE.json's metric is quoted on.

Each sample is `[task, <img>, <img_feat> x R, </img>, <event>, text..., </event>]` on the encoder
side (ids follow reference `src/data/tokenization.py:36-57`: the 16 added special tokens start at
50265) and `[0, text...]` / shifted labels ending in `</s>` on the decoder side, the layout the
reference's `Collator` (`src/data/collation.py:68-213`) produces.  Region features are
`[R, 2048 + 4]` fp32: 2048 pooled (non-negative) features followed by a raw pixel box
(`src/data/dataset.py:44-47`).
"""
import torch

BEGIN_IMG, END_IMG, BEGIN_EVENT, END_EVENT = 50265, 50266, 50267, 50268
TASK_IDS = (50269, 50270, 50271)  # before, intent, after
IMG_FEAT = 50273
PAD, BOS, EOS = 1, 0, 2
TEXT_LO, TEXT_HI = 3, 50265


def make_batch(batch_size, enc_len=64, dec_len=32, num_regions=36, seed=1234, feat_dim=2048,
               regions=None, event_lens=None, label_lens=None, vocab_hi=TEXT_HI, img_feat_id=IMG_FEAT,
               special_base=BEGIN_IMG):
    """Returns the dict a reference DataLoader batch carries (CPU tensors).

    regions / event_lens / label_lens: optional per-sample lists for the ragged variant
    (right-padded with id 1, mask 0, label pads -100).
    """
    g = torch.Generator().manual_seed(seed)
    b = batch_size
    regions = [num_regions] * b if regions is None else list(regions)
    off = special_base - BEGIN_IMG
    fixed = 5  # task, <img>, </img>, <event>, </event>
    if event_lens is None:
        event_lens = [enc_len - fixed - r for r in regions]
    if label_lens is None:
        label_lens = [dec_len] * b
    input_ids = torch.full((b, enc_len), PAD, dtype=torch.long)
    attention_mask = torch.zeros((b, enc_len), dtype=torch.long)
    feats = []
    for i in range(b):
        r, e = regions[i], event_lens[i]
        assert fixed + r + e <= enc_len and e >= 0
        task = TASK_IDS[int(torch.randint(0, 3, (1,), generator=g))] + off
        text = torch.randint(TEXT_LO, vocab_hi, (e,), generator=g)
        row = [task, BEGIN_IMG + off] + [img_feat_id] * r + [END_IMG + off, BEGIN_EVENT + off] \
            + text.tolist() + [END_EVENT + off]
        input_ids[i, : len(row)] = torch.tensor(row)
        attention_mask[i, : len(row)] = 1
        f = torch.randn((r, feat_dim), generator=g).abs()
        xy = torch.rand((r, 2), generator=g) * 500.0
        wh = torch.rand((r, 2), generator=g) * 484.0 + 16.0
        feats.append(torch.cat([f, xy, xy + wh], dim=1) if r > 0 else torch.empty(0))
    decoder_input_ids = torch.full((b, dec_len), PAD, dtype=torch.long)
    decoder_attention_mask = torch.zeros((b, dec_len), dtype=torch.long)
    labels = torch.full((b, dec_len), -100, dtype=torch.long)
    for i in range(b):
        n = label_lens[i]
        assert 1 <= n <= dec_len
        text = torch.randint(TEXT_LO, vocab_hi, (n - 1,), generator=g)
        decoder_input_ids[i, :n] = torch.cat([torch.tensor([BOS]), text])
        decoder_attention_mask[i, :n] = 1
        labels[i, :n] = torch.cat([text, torch.tensor([EOS])])
    return {
        "input_ids": input_ids,
        "attention_mask": attention_mask,
        "image_features": feats,
        "decoder_input_ids": decoder_input_ids,
        "decoder_attention_mask": decoder_attention_mask,
        "labels": labels,
        "index": list(range(b)),
        "task_type": ["intent"] * b,
    }


def make_pretrain_batch(batch_size, enc_len=80, dec_len=48, num_regions=50, seed=1234, num_labels=1601,
                        num_attributes=129, num_relations=129, mrm_probability=0.2, n_attr=4, n_rel=3, cls_id=50276,
                        **kw):
    """Synthetic multi-task pre-training batch (BASELINE config 4): the VCG batch plus
    masked-region-modelling targets (soft labels over `num_labels` classes at decoder positions whose label is
    <cls>, reference src/data/collation.py:113-147), attribute labels and relation triples."""
    b = make_batch(batch_size, enc_len=enc_len, dec_len=dec_len, num_regions=num_regions, seed=seed, **kw)
    g = torch.Generator().manual_seed(seed + 7)
    labels = b["labels"]
    B, T = labels.shape
    valid = labels != -100
    mrm_mask = (torch.rand((B, T), generator=g) < mrm_probability) & valid
    mrm_mask[:, 0] = False
    labels[mrm_mask] = cls_id                      # collation.py: masked regions carry <cls> as their label
    b["mrm_mask"] = labels == cls_id
    b["mrm_labels"] = [torch.softmax(torch.randn((int(m.sum()), num_labels), generator=g), dim=-1) for m in b["mrm_mask"]]
    attr_mask = torch.zeros((B, T))
    attr_labels, rel_labels = [], []
    for i in range(B):
        pos = torch.nonzero(valid[i] & ~b["mrm_mask"][i]).reshape(-1)
        pick = pos[torch.randperm(len(pos), generator=g)[: min(n_attr, len(pos))]]
        attr_mask[i, pick] = 1
        attr_labels.append(torch.randint(0, num_attributes, (len(pick),), generator=g))
        rels = []
        for _ in range(n_rel):
            o, s = [int(x) for x in pos[torch.randint(0, len(pos), (2,), generator=g)]]
            rels.append({"object_index": o, "subject_index": s,
                         "label": int(torch.randint(0, num_relations, (1,), generator=g))})
        rel_labels.append(rels)
    # the reference orders attribute labels by position (boolean-mask order)
    b["attribute_mask"] = attr_mask
    b["attribute_labels"] = attr_labels
    b["relation_labels"] = rel_labels
    return b
