"""Condition / label encoding for KM-BART batches: the counterpart of the reference's ConditionTokenizer
(src/data/tokenization.py:6-268), same constructor keywords, attributes, method names and returned keys.

The reference glues marker strings and free text into one string per sample and lets the BART tokenizer split it
again (tokenization.py:118-171, 198-222).  Here every row is assembled directly from ids: the 16 markers have fixed
ids (tokenization.py:36-57 appends them after the base vocabulary), so only the free-text pieces (event, MLM text,
label) go through the BPE -- in one batched call -- and the masks are known from the piece boundaries instead of
being searched for.  The two agree id for id with a fast (Rust) BART tokenizer, which tokenizes the text between two
markers exactly as it tokenizes that text alone; tests/test_collation_cpu.py pins this against outputs of the
reference class itself.

The base vocabulary is pluggable (`base_tokenizer=`): anything with the HuggingFace tokenizer call convention.
`facebook/bart-large` is loaded when none is given and its files are on disk; offline use
`src.data.offline_tokenizer`.
"""
import numpy as np
import torch

from src.utils import TaskType

_MARKER_ORDER = ("begin_img", "end_img", "begin_event", "end_event", "before", "intent", "after", "caption",
                 "img_feat", "begin_mlm", "end_mlm", "cls_token", "token1", "token2", "token3", "region_caption")


def _as_list(x):
    return x if isinstance(x, list) else [x]


class ConditionTokenizer:
    """Encodes (task type, number of regions, event, MLM sentence) and labels; NOT a transformers tokenizer itself."""

    def __init__(self, pretrained_model_name="facebook/bart-large", begin_img="<img>", end_img="</img>",
                 begin_event="<event>", end_event="</event>", before="<before>", intent="<intent>", after="<after>",
                 caption="<caption>", img_feat="<img_feat>", begin_mlm="<mlm>", end_mlm="</mlm>", cls_token="<cls>",
                 token1="<token1>", token2="<token2>", token3="<token3>", region_caption="<region_caption>",
                 base_tokenizer=None):
        if base_tokenizer is None:
            from src.data.offline_tokenizer import load_base_tokenizer
            base_tokenizer = load_base_tokenizer(pretrained_model_name)
        self._base_tokenizer = base_tokenizer
        markers = dict(begin_img=begin_img, end_img=end_img, begin_event=begin_event, end_event=end_event,
                       before=before, intent=intent, after=after, caption=caption, img_feat=img_feat,
                       begin_mlm=begin_mlm, end_mlm=end_mlm, cls_token=cls_token, token1=token1, token2=token2,
                       token3=token3, region_caption=region_caption)
        self.additional_special_tokens = [markers[k] for k in _MARKER_ORDER]   # tokenization.py:36-53: this order
        base_tokenizer.add_special_tokens({"additional_special_tokens": self.additional_special_tokens})
        for name in ("begin_img", "end_img", "begin_event", "end_event", "before", "intent", "after", "img_feat",
                     "caption", "begin_mlm", "end_mlm", "cls_token", "region_caption"):
            setattr(self, name, markers[name])
            setattr(self, name + "_id", self.convert_tokens_to_ids(markers[name]))
        self.vocab_size = base_tokenizer.vocab_size
        for name in ("bos_token", "eos_token", "pad_token", "unk_token"):
            setattr(self, name, getattr(base_tokenizer, name))
            setattr(self, name + "_id", getattr(base_tokenizer, name + "_id"))
        self._task_ids = {TaskType.INTENT: self.intent_id, TaskType.BEFORE: self.before_id,
                          TaskType.AFTER: self.after_id, TaskType.CAPTION: self.caption_id,
                          TaskType.REGION_CAPTION: self.region_caption_id}

    # ---- plumbing ------------------------------------------------------------------------------------
    def encode(self, *args, **kwargs):
        return self._base_tokenizer(*args, **kwargs)

    def decode(self, token_ids, skip_special_tokens=False):
        return self._base_tokenizer.decode(token_ids, skip_special_tokens=skip_special_tokens)

    def convert_tokens_to_ids(self, tokens):
        return self._base_tokenizer.convert_tokens_to_ids(tokens)

    def convert_ids_to_tokens(self, ids):
        return self._base_tokenizer.convert_ids_to_tokens(ids)

    def get_base_tokenizer(self):
        return self._base_tokenizer

    def __len__(self):
        return len(self._base_tokenizer)

    def _text_ids(self, texts):
        """BPE ids of each free-text piece (no <s> / </s>); one batched call, '' -> []"""
        if not texts:
            return []
        return [list(x) for x in self._base_tokenizer(list(texts), add_special_tokens=False)["input_ids"]]

    def _pad(self, rows):
        """ragged id rows -> (ids [B, L] int64 padded with pad_token_id, attention_mask [B, L] int64)"""
        width = max((len(r) for r in rows), default=0)
        ids = np.full((len(rows), width), self.pad_token_id, dtype=np.int64)
        att = np.zeros((len(rows), width), dtype=np.int64)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = r
            att[i, : len(r)] = 1
        return ids, att

    # ---- encoder side --------------------------------------------------------------------------------
    def encode_condition(self, task_type, img_num=None, event=None, mlm=None):
        """Rows `task [<img> <img_feat>*n </img>] [<event> EVENT </event>] [<mlm> MLM </mlm>]` (tokenization.py:100-195).
        Returns input_ids / attention_mask and, for each given part, event_mask / mlm_mask (True on the free-text
        tokens between the markers) and img_mask (True on <img_feat>)."""
        task_type = _as_list(task_type)
        rows = []
        for t in task_type:
            if t not in self._task_ids:
                raise ValueError('Unexpected task type "{}"'.format(t))
            rows.append([self._task_ids[t]])
        n = len(rows)
        spans = {}
        if img_num is not None:
            for i, k in enumerate(_as_list(img_num)):
                rows[i] += [self.begin_img_id] + [self.img_feat_id] * int(k) + [self.end_img_id]
        for key, texts, lo, hi in (("event_mask", event, self.begin_event_id, self.end_event_id),
                                   ("mlm_mask", mlm, self.begin_mlm_id, self.end_mlm_id)):
            if texts is None:
                continue
            pieces = self._text_ids(_as_list(texts))
            where = []
            for i, ids in enumerate(pieces):
                first = len(rows[i]) + 1
                rows[i] += [lo] + ids + [hi]
                where.append((first, first + len(ids)))
            spans[key] = where
        ids, att = self._pad(rows)
        out = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(att)}
        for key, where in spans.items():
            m = np.zeros(ids.shape, dtype=bool)
            for i, (a, b) in enumerate(where):
                m[i, a:b] = True
            out[key] = torch.from_numpy(m)
        if img_num is not None:
            out["img_mask"] = out["input_ids"] == self.img_feat_id
        assert len(out["input_ids"]) == n
        return out

    # ---- decoder side --------------------------------------------------------------------------------
    def encode_label(self, label, img_num=None):
        """Target rows `[<img> <img_feat>*n </img>] <s> LABEL </s>` split into labels (without <s>) and
        decoder_input_ids / decoder_attention_mask (without </s>), tokenization.py:197-250."""
        pieces = self._text_ids(_as_list(label))
        prefix = [[] for _ in pieces]
        if img_num is not None:
            for i, k in enumerate(_as_list(img_num)):
                prefix[i] = [self.begin_img_id] + [self.img_feat_id] * int(k) + [self.end_img_id]
        lab, att = self._pad([p + ids + [self.eos_token_id] for p, ids in zip(prefix, pieces)])
        dec, _ = self._pad([p + [self.bos_token_id] + ids for p, ids in zip(prefix, pieces)])
        out = {"labels": torch.from_numpy(lab), "decoder_input_ids": torch.from_numpy(dec),
               "decoder_attention_mask": torch.from_numpy(att)}
        if img_num is not None:
            out["label_img_mask"] = out["labels"] == self.img_feat_id
            out["decoder_input_img_mask"] = out["decoder_input_ids"] == self.img_feat_id
        return out
