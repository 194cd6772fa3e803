"""Readers for KM-BART's on-disk dataset format: the counterparts of the reference's dataset classes
(src/data/dataset.py:24-214; writers scripts/prepare_vcg.py:24-42,88-95 and the other scripts/prepare_*.py).

Layout of one dataset directory (what the prepare scripts leave behind):

    <data_dir>/<split>.json              list of records {img_id, task_type, event?, labels?, index?, ...}
    <data_dir>/<split>_eval.json         same, one record per image (eval_mode)
    <data_dir>/reason_<split>[_eval].json   records of the reasoning corpora (ReasonDataset)
    <data_dir>/<split>_region.json       Visual Genome: list of {img_id, region_id, description}; <split>.json is then a
                                         dict img_id -> record with objects / relations
    <image_dir>/<split>/<img_id>.pkl     pickle {image_features [R, 2048] f32, boxes [R, 4], mrm_labels [R, 1601]?}
                                         (Visual Genome: image_feature / image_box / image_score, object_* , region_*)

Every `__getitem__` returns the record plus `image_features` = [R, 2052] float32 (features ++ raw pixel box, the
reference does not normalise the boxes) and `mrm_labels` when the pickle has them -- the dict `Collator` consumes.
No image library is needed to READ the format (the reference module imports cv2 without using it here).
`write_synthetic_split` produces the same layout from closed-form data for tests and offline runs.
"""
import json
import os
import pickle

import numpy as np
from torch.utils.data import Dataset

from src.utils import TaskType


def _load_pickle(image_dir, split, img_id):
    with open(os.path.join(image_dir, split, "%s.pkl" % img_id), "rb") as f:
        return pickle.load(f)


def _regions(blob):
    """[R, 2048] features ++ [R, 4] boxes -> [R, 2052] float32 (dataset.py:44-47)"""
    return np.concatenate([blob["image_features"], blob["boxes"]], axis=1).astype(np.float32)


class _JsonRecords(Dataset):
    """a JSON list of records + one feature pickle per image"""

    def __init__(self, data_dir, file_name, image_dir, split, use_image):
        self._data_dir = data_dir
        self._image_dir = data_dir if image_dir is None else image_dir
        self._split = split
        self._use_image = use_image
        with open(os.path.join(data_dir, file_name), "r") as f:
            self._dataset = json.load(f)

    def __len__(self):
        return len(self._dataset)

    def _with_features(self, record):
        out = dict(record)
        if self._use_image:
            blob = _load_pickle(self._image_dir, self._split, record["img_id"])
            out["image_features"] = _regions(blob)
            if "mrm_labels" in blob:
                out["mrm_labels"] = blob["mrm_labels"]
        return out

    def __getitem__(self, index):
        return self._with_features(self._dataset[index])


class COCODataset(_JsonRecords):
    """captions; `eval_mode` loads "<split>_eval.json" where every image appears once (dataset.py:24-55)"""

    def __init__(self, data_dir, image_dir=None, split="train", eval_mode=False, use_image=True):
        super().__init__(data_dir, split + ("_eval.json" if eval_mode else ".json"), image_dir, split, use_image)


class VCGDataset(COCODataset):
    """Visual Commonsense Graphs (dataset.py:58-89).  `use_event=False` keeps only the first word of the event (the
    person tag); `pretrain=True` turns the record into a captioning sample whose target is the event."""

    def __init__(self, data_dir, image_dir=None, split="train", eval_mode=False, use_image=True, use_event=True,
                 pretrain=False):
        super().__init__(data_dir, image_dir=image_dir, split=split, eval_mode=eval_mode, use_image=use_image)
        self._use_event, self._pretrain = use_event, pretrain

    def __getitem__(self, index):
        out = super().__getitem__(index)
        if not self._use_event:
            out["event"] = out["event"].split()[0]
        if self._pretrain:
            out["labels"] = out.pop("event")
            out["task_type"] = TaskType.CAPTION
        return out


class SBUDataset(COCODataset):
    """SBU captions (dataset.py:92-106): always a captioning sample, target stripped of surrounding blanks"""

    def __init__(self, data_dir, image_dir=None, split="train", use_image=True):
        super().__init__(data_dir, image_dir=image_dir, split=split, eval_mode=False, use_image=use_image)

    def __getitem__(self, index):
        out = super().__getitem__(index)
        out["task_type"] = TaskType.CAPTION
        out["labels"] = out["labels"].strip()
        return out


class CCDataset(SBUDataset):
    """Conceptual Captions: same format as SBU (dataset.py:109-110)"""


class VGDataset(Dataset):
    """Visual Genome region captioning (dataset.py:113-167): one sample per REGION; its regions are
    [whole image, every object, the described region] and it carries the image's objects / relations for the
    attribute- and relation-prediction heads."""

    def __init__(self, data_dir, image_dir=None, split="train"):
        self._data_dir = data_dir
        self._image_dir = data_dir if image_dir is None else image_dir
        self._split = split
        with open(os.path.join(data_dir, split + ".json"), "r") as f:
            self._dataset = json.load(f)                 # img_id (str) -> record
        with open(os.path.join(data_dir, split + "_region.json"), "r") as f:
            self._region_dataset = json.load(f)

    def __len__(self):
        return len(self._region_dataset)

    def __getitem__(self, index):
        region = self._region_dataset[index]
        record = self._dataset[str(region["img_id"])]
        blob = _load_pickle(self._image_dir, self._split, record["img_id"])
        k = blob["region_ids"].index(region["region_id"])
        whole = np.concatenate([blob["image_feature"], blob["image_box"]], axis=0)[None, :]
        objects = np.concatenate([blob["object_features"], blob["object_boxes"]], axis=1)
        described = np.concatenate([blob["region_features"][k], blob["region_boxes"][k]], axis=0)[None, :]
        out = dict(record)
        out["image_features"] = np.concatenate([whole, objects, described], axis=0)
        out["mrm_labels"] = np.concatenate([blob["image_score"][None, :], blob["object_scores"],
                                            blob["region_scores"][k: k + 1]], axis=0)
        out["object_ids"] = blob["object_ids"]
        out["task_type"] = TaskType.REGION_CAPTION
        out["labels"] = region["description"]
        return out


class ReasonDataset(_JsonRecords):
    """"reason_<split>.json" corpora (dataset.py:170-214).  A record whose feature file is missing yields None (the
    collator drops it); every sample carries `dataset_index`."""

    def __init__(self, data_dir, image_dir=None, split="train", eval_mode=False, use_image=True, use_event=True):
        super().__init__(data_dir, "reason_" + split + ("_eval.json" if eval_mode else ".json"), image_dir, split,
                         use_image)
        self._use_event = use_event

    def get_raw_data(self, index):
        return self._dataset[index]

    def __getitem__(self, index):
        record = self._dataset[index]
        try:
            out = self._with_features(record)
        except FileNotFoundError:
            return None
        if not self._use_event:
            out["event"] = ""
        out["dataset_index"] = index
        return out


# ---- synthetic data in the same format ------------------------------------------------------------------
def _unit(n, seed):
    """n floats in [0, 1) from splitmix64(index, seed): exact on every machine"""
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) \
            + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return ((x >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


_WORDS = ("PersonX walks into the kitchen and opens fridge to get a cold drink man is holding red umbrella in rain "
          "near bus stop two dogs are playing with ball on green grass park wants ask woman behind counter for "
          "directions station tall building next river has many small windows blue door").split()


def _sentence(n, seed):
    pick = (_unit(n, seed) * len(_WORDS)).astype(np.int64)
    return " ".join(_WORDS[i] for i in pick)


def write_synthetic_split(data_dir, split="train", n_images=8, records_per_image=3, regions=36, seed=0,
                          with_mrm=True, reason=False):
    """Writes `<split>.json` (+ `<split>_eval.json`, or `reason_<split>.json`) and one pickle per image under
    data_dir, in the format above: ReLU-like features, raw pixel boxes, soft class labels.  Returns the records."""
    os.makedirs(os.path.join(data_dir, split), exist_ok=True)
    records, eval_records = [], []
    tasks = (TaskType.INTENT, TaskType.BEFORE, TaskType.AFTER)
    for img in range(n_images):
        img_id = "img%04d" % img
        r = int(regions if np.isscalar(regions) else regions[img % len(regions)])
        feats = np.abs(_unit(r * 2048, seed * 977 + img).reshape(r, 2048) * 2.0 - 0.5)
        u = _unit(r * 4, seed * 977 + img + 500).reshape(r, 4)
        xy = u[:, :2] * 500.0
        boxes = np.concatenate([xy, xy + u[:, 2:] * 484.0 + 16.0], axis=1).astype(np.float32)
        blob = {"image_features": feats.astype(np.float32), "boxes": boxes}
        if with_mrm:
            s = _unit(r * 1601, seed * 977 + img + 900).reshape(r, 1601).astype(np.float64)
            blob["mrm_labels"] = (s / s.sum(1, keepdims=True)).astype(np.float32)
        with open(os.path.join(data_dir, split, img_id + ".pkl"), "wb") as f:
            pickle.dump(blob, f)
        for k in range(records_per_image):
            rec = {"img_id": img_id, "task_type": tasks[(img + k) % 3], "index": len(records),
                   "event": "1 " + _sentence(6 + (img + k) % 5, seed + 31 * img + k),
                   "labels": _sentence(4 + (img * 3 + k) % 6, seed + 31 * img + k + 7)}
            records.append(rec)
            if k == 0:
                eval_records.append(rec)
    name = ("reason_" if reason else "") + split
    with open(os.path.join(data_dir, name + ".json"), "w") as f:
        json.dump(records, f)
    with open(os.path.join(data_dir, name + "_eval.json"), "w") as f:
        json.dump(eval_records, f)
    return records


def write_synthetic_vg(data_dir, split="train", n_images=4, objects=5, regions_per_image=2, seed=0,
                       num_attributes=129, num_relations=201):
    """Visual Genome flavour of the format: `<split>.json` (dict img_id -> record with objects / relations),
    `<split>_region.json` and one pickle per image with whole-image / object / region features and scores."""
    os.makedirs(os.path.join(data_dir, split), exist_ok=True)
    images, region_list = {}, []

    def feat(n, s):
        return np.abs(_unit(n * 2048, s).reshape(n, 2048) * 2.0 - 0.5).astype(np.float32)

    def box(n, s):
        u = _unit(n * 4, s).reshape(n, 4)
        xy = u[:, :2] * 500.0
        return np.concatenate([xy, xy + u[:, 2:] * 484.0 + 16.0], axis=1).astype(np.float32)

    def score(n, s):
        p = _unit(n * 1601, s).reshape(n, 1601).astype(np.float64)
        return (p / p.sum(1, keepdims=True)).astype(np.float32)

    for img in range(n_images):
        img_id = 1000 + img
        base = seed * 7919 + img * 13
        n_obj = objects + img % 2
        object_ids = [img_id * 100 + j for j in range(n_obj)]
        region_ids = [img_id * 10 + j for j in range(regions_per_image)]
        blob = {"image_feature": feat(1, base)[0], "image_box": box(1, base + 1)[0], "image_score": score(1, base + 2)[0],
                "object_features": feat(n_obj, base + 3), "object_boxes": box(n_obj, base + 4),
                "object_scores": score(n_obj, base + 5), "object_ids": object_ids,
                "region_features": feat(regions_per_image, base + 6), "region_boxes": box(regions_per_image, base + 7),
                "region_scores": score(regions_per_image, base + 8), "region_ids": region_ids}
        with open(os.path.join(data_dir, split, "%d.pkl" % img_id), "wb") as f:
            pickle.dump(blob, f)
        objs = [{"object_id": oid, **({"attribute_ids": [(3 * j + img) % num_attributes]} if j % 3 != 2 else {})}
                for j, oid in enumerate(object_ids)]
        rels = [{"object_id": object_ids[j % n_obj], "subject_id": object_ids[(2 * j + 1) % n_obj],
                 "predicate_id": (5 * j + img) % num_relations} for j in range(n_obj)]
        images[str(img_id)] = {"img_id": img_id, "objects": objs, "relations": rels}
        for j, rid in enumerate(region_ids):
            region_list.append({"img_id": img_id, "region_id": rid, "description": _sentence(5 + j, base + 20 + j)})
    with open(os.path.join(data_dir, split + ".json"), "w") as f:
        json.dump(images, f)
    with open(os.path.join(data_dir, split + "_region.json"), "w") as f:
        json.dump(region_list, f)
    return region_list
