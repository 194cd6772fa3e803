"""Batch construction for KM-BART (fine-tuning, generation and multi-task pre-training): the counterpart of the
reference's Collator (src/data/collation.py:9-247) -- same constructor keywords, same keys and tensors out, same
random draws (so a seeded run masks the same tokens and regions) -- built around ONE packed region-feature buffer.

The reference returns `image_features` as a Python list of per-sample tensors and masks regions sample by sample
with a cat of fresh tensors (collation.py:73-76, 120-130); the training loop then copies the list to the GPU tensor
by tensor (training.py:121).  Here the ragged features are gathered once into a [Ntot, 2052] float32 buffer with CSR
offsets (`kmbart.data.PackedFeatures`, optionally pinned), masked-region rows are zeroed in that buffer with one
indexed write, and the batch reaches the device as two copies.  `PackedFeatures` still iterates / indexes like the
list the reference hands out.
"""
import warnings

import numpy as np
import torch

from kmbart.data import PackedFeatures
from src.utils import TaskType

_FEATURE_DIMS = 2048   # leading dims zeroed by MRM; the trailing 4 box coordinates are kept (collation.py:127-130)


class Collator:
    """collate_fn for every KM-BART dataset (see src/data/dataset.py for the entry format)."""

    def __init__(self, tokenizer, has_label=True, mlm_enabled=False, mrm_enabled=False, rp_enabled=False,
                 ap_enabled=False, mlm_probability=0.0, mrm_probability=0.0, event_max_len=20, lm_max_len=30,
                 max_img_num=30, max_rel_count=80, pin_memory=False, feature_size=2052):
        if mlm_enabled and not has_label:
            raise ValueError('mlm_enabled can not be true while has_label is false. MLM need labels.')
        if ap_enabled and not has_label:
            raise ValueError('ap_enabled can not be true while has_label is false. attribute prediction need labels.')
        if rp_enabled and not has_label:
            raise ValueError('rp_enabled can not be true while has_label is false. relation prediction need labels.')
        if (rp_enabled or ap_enabled) and not mrm_enabled:
            raise ValueError('if rp/ap is enabled, mrm must also be enabled')
        self._tokenizer = tokenizer
        self._has_label, self._mlm_enabled, self._mrm_enabled = has_label, mlm_enabled, mrm_enabled
        self._rp_enabled, self._ap_enabled = rp_enabled, ap_enabled
        self._mlm_probability, self._mrm_probability = mlm_probability, mrm_probability
        self._event_max_len, self._lm_max_len = event_max_len, lm_max_len
        self._max_img_num, self._max_rel_count = max_img_num, max_rel_count
        self._pin, self._feature_size = pin_memory, feature_size

    # ---- pieces ----------------------------------------------------------------------------------------
    def _clip_texts(self, texts, length):
        """first `length` BPE tokens of every text, back as text (collation.py:64-66); one batched tokenizer call"""
        base = self._tokenizer.get_base_tokenizer()
        if not texts:
            return []
        ids = base(list(texts), add_special_tokens=False)["input_ids"]
        return [base.decode(row[:length]) for row in ids]

    def _pack_regions(self, batch):
        """entries -> (PackedFeatures over the first max_img_num regions of every entry, regions per entry)"""
        counts, blocks = [], []
        for e in batch:
            f = e.get("image_features")
            if f is None:
                counts.append(0)
                continue
            f = np.asarray(f)[: self._max_img_num]
            counts.append(int(f.shape[0]))
            if f.shape[0]:
                blocks.append(f.astype(np.float32, copy=False))
        total = sum(counts)
        buf = torch.empty((max(total, 1), self._feature_size), dtype=torch.float32,
                          pin_memory=bool(self._pin and torch.cuda.is_available()))
        if total:
            np.concatenate(blocks, axis=0, out=buf.numpy()[:total])   # the only copy of the features on the host
        else:
            buf.zero_()
        offsets = torch.zeros(len(batch) + 1, dtype=torch.int32)
        offsets[1:] = torch.tensor(counts, dtype=torch.int32).cumsum(0)
        return PackedFeatures(buf, offsets, total), counts

    def _mask_tokens(self, inputs, input_mask):
        """BERT-style corruption of the MLM span (collation.py:216-247, after transformers' MLM collator): of the
        selected tokens 80 % -> <mask>, 10 % -> random id, 10 % kept.  The four draws are made over the whole
        [B, L] grid in the reference's order, which is what keeps seeded runs identical."""
        base = self._tokenizer.get_base_tokenizer()
        grid = inputs.shape
        special = torch.isin(inputs, torch.tensor(sorted(set(base.all_special_ids)), dtype=inputs.dtype))
        p = torch.full(grid, self._mlm_probability, dtype=torch.float)
        p.masked_fill_(special, 0.0)
        if base.pad_token is not None:
            p.masked_fill_(inputs.eq(base.pad_token_id), 0.0)
        chosen = torch.bernoulli(p).bool()
        to_mask = torch.bernoulli(torch.full(grid, 0.8)).bool() & chosen
        to_random = torch.bernoulli(torch.full(grid, 0.5)).bool() & chosen & ~to_mask
        random_ids = torch.randint(base.vocab_size, grid, dtype=torch.long)
        inputs[to_mask & input_mask] = base.mask_token_id
        swap = to_random & input_mask
        inputs[swap] = random_ids[swap]
        return inputs

    def _mask_regions(self, batch, input_ids, img_mask, packed, counts):
        """MRM (collation.py:113-132): draw over the whole grid, turn the hit <img_feat> ids into <cls>, zero the 2048
        feature dims of those regions in the packed buffer, and collect their soft labels."""
        hit = torch.bernoulli(torch.full(input_ids.shape, self._mrm_probability, dtype=torch.float)).bool() & img_mask
        input_ids[hit] = self._tokenizer.cls_token_id
        offs = packed.offsets.tolist()
        soft, rows = [], []
        for i, e in enumerate(batch):
            which = hit[i][img_mask[i]].nonzero(as_tuple=False).flatten()      # region indices of sample i
            src = e.get("mrm_labels")
            if src is None:
                soft.append(torch.zeros((0, 0)))
            else:
                src = torch.as_tensor(np.asarray(src)[: self._max_img_num], dtype=torch.float32)
                soft.append(src[which].clone())
            if counts[i]:
                rows.append(which + offs[i])
        if rows:
            rows = torch.cat(rows)
            packed.packed[rows, :_FEATURE_DIMS] = 0.0
        return soft

    def _vg_start(self, labels_row):
        """position of the first object's slot in a Visual Genome target row: `<img> whole-image obj0 obj1 ...`"""
        return int((labels_row == self._tokenizer.begin_img_id).nonzero(as_tuple=True)[0][0]) + 2

    def _attribute_targets(self, batch, labels):
        """(attribute_labels list[LongTensor], attribute_mask float [B, T]), collation.py:149-165"""
        mask = torch.zeros(labels.size())
        targets = []
        for i, e in enumerate(batch):
            picked = []
            if "object_ids" in e:                     # only Visual Genome entries carry objects
                start = self._vg_start(labels[i])
                by_id = {o["object_id"]: o for o in e["objects"]}
                for slot, oid in enumerate(e["object_ids"][: self._max_img_num - 2]):
                    attrs = by_id[oid].get("attribute_ids")
                    if attrs is not None:
                        mask[i, slot + start] = 1
                        picked.append(attrs[0])     # the first attribute only
            targets.append(torch.LongTensor(picked))
        return targets, mask

    def _relation_targets(self, batch, labels):
        """list[B] of [{'object_index', 'subject_index', 'label'}], at most max_rel_count each (collation.py:167-190)"""
        out = []
        for i, e in enumerate(batch):
            rels = []
            if "object_ids" in e:
                start = self._vg_start(labels[i])
                slot_of = {oid: start + k for k, oid in enumerate(e["object_ids"][: self._max_img_num - 2])}
                for r in e["relations"]:
                    if r["object_id"] in slot_of and r["subject_id"] in slot_of:
                        rels.append({"object_index": slot_of[r["object_id"]],
                                     "subject_index": slot_of[r["subject_id"]], "label": r["predicate_id"]})
                        if len(rels) >= self._max_rel_count:
                            break
            out.append(rels)
        return out

    # ---- the collate function --------------------------------------------------------------------------
    def __call__(self, batch):
        batch = [e for e in batch if e is not None]       # ReasonDataset yields None for a missing feature file
        if any(e["task_type"] not in TaskType.ALL_TYPES for e in batch):
            warnings.warn('Unexpected task type in batch')
        tok = self._tokenizer
        packed, counts = self._pack_regions(batch)
        task_type = [e["task_type"] for e in batch]
        has_event = [("event" in e) for e in batch]
        clipped = self._clip_texts([e["event"] for e in batch if "event" in e], self._event_max_len)
        it = iter(clipped)
        event = [next(it) if h else "" for h in has_event]
        target = self._clip_texts([e["labels"] for e in batch], self._lm_max_len) if self._has_label else None
        mlm = None
        if self._mlm_enabled:
            mlm = list(target)
            for i, t in enumerate(task_type):        # reasoning samples denoise their event instead of the target
                if t in (TaskType.BEFORE, TaskType.AFTER, TaskType.INTENT):
                    mlm[i], event[i] = event[i], ""

        cond = tok.encode_condition(task_type=task_type, img_num=counts, event=event, mlm=mlm)
        input_ids = cond["input_ids"]
        if self._mlm_enabled:
            input_ids = self._mask_tokens(input_ids, cond["mlm_mask"])
        out = {"input_ids": input_ids, "attention_mask": cond["attention_mask"], "image_features": packed,
               "index": [e.get("index") for e in batch], "task_type": task_type}
        img_mask = cond["img_mask"]
        if self._mrm_enabled:
            out["mrm_labels"] = self._mask_regions(batch, input_ids, img_mask, packed, counts)

        if self._has_label:
            enc = tok.encode_label(label=target, img_num=counts if self._mrm_enabled else None)
            labels, dec_in = enc["labels"], enc["decoder_input_ids"]
            if self._mrm_enabled:       # the decoder sees (and predicts) the same <img_feat> / <cls> pattern
                labels[enc["label_img_mask"]] = input_ids[img_mask]
                dec_in[enc["decoder_input_img_mask"]] = input_ids[img_mask]
            if self._ap_enabled:
                out["attribute_labels"], out["attribute_mask"] = self._attribute_targets(batch, labels)
            if self._rp_enabled:
                out["relation_labels"] = self._relation_targets(batch, labels)
            ignore = torch.tensor([tok.pad_token_id, tok.begin_img_id, tok.end_img_id, tok.img_feat_id])
            labels[torch.isin(labels, ignore)] = -100
            out["labels"], out["decoder_input_ids"] = labels, dec_in
            out["decoder_attention_mask"] = enc["decoder_attention_mask"]
            if self._mrm_enabled:
                out["mrm_mask"] = labels == tok.cls_token_id
        if batch and "question_id" in batch[0]:
            out["question_id"] = [e["question_id"] for e in batch]
        if batch and "dataset_index" in batch[0]:
            out["dataset_index"] = [e.get("dataset_index") for e in batch]
        if self._has_label:
            out["raw_labels"] = [e["labels"] for e in batch]
        return out
