"""On-disk dataset format: write the format, read it back through every dataset class and
run the result through the collator and a DataLoader -- the path a `vcg_train.py` / `pretrain.py` run takes before
the batch reaches the GPU.  (The reference's dataset module cannot be imported here -- it needs cv2 -- so this row
is pinned on the documented format, reference src/data/dataset.py:24-214, not on a reference run.)"""
import os
import pickle
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "km-bart_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from src.data.collation import Collator  # noqa: E402
from src.data.dataset import (CCDataset, COCODataset, ReasonDataset, SBUDataset, VCGDataset, VGDataset,  # noqa: E402
                              write_synthetic_split, write_synthetic_vg)
from src.data.offline_tokenizer import load_base_tokenizer  # noqa: E402
from src.data.tokenization import ConditionTokenizer  # noqa: E402
from src.utils import TaskType  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _tok():
    return ConditionTokenizer(base_tokenizer=load_base_tokenizer(os.path.join(GOLD, "tiny_bpe_tokenizer.json")))


def test_vcg_format_round_trip(tmp_path):
    d = str(tmp_path)
    recs = write_synthetic_split(d, "train", n_images=3, records_per_image=2, regions=[5, 0, 36], seed=3)
    ds = VCGDataset(d, split="train")
    assert len(ds) == len(recs) == 6
    e = ds[5]
    blob = pickle.load(open(os.path.join(d, "train", e["img_id"] + ".pkl"), "rb"))
    assert e["image_features"].dtype == np.float32 and e["image_features"].shape == (36, 2052)
    assert np.array_equal(e["image_features"][:, :2048], blob["image_features"])
    assert np.array_equal(e["image_features"][:, 2048:], blob["boxes"])          # raw pixel boxes, not normalised
    assert e["mrm_labels"].shape == (36, 1601) and e["event"] == recs[5]["event"] and e["index"] == 5
    assert ds[2]["image_features"].shape == (0, 2052)                            # an image without regions
    # eval split: one record per image
    assert len(VCGDataset(d, split="train", eval_mode=True)) == 3
    # --no_event keeps the person tag only; pretrain=True turns the event into a caption target
    assert VCGDataset(d, split="train", use_event=False)[0]["event"] == recs[0]["event"].split()[0]
    p = VCGDataset(d, split="train", pretrain=True)[0]
    assert "event" not in p and p["labels"] == recs[0]["event"] and p["task_type"] == TaskType.CAPTION
    # --no_image never opens a pickle
    t = VCGDataset(d, split="train", use_image=False)[0]
    assert "image_features" not in t and "mrm_labels" not in t
    # image_dir separate from data_dir
    os.rename(os.path.join(d, "train"), os.path.join(d, "feats_train"))
    os.makedirs(os.path.join(d, "elsewhere"))
    os.rename(os.path.join(d, "feats_train"), os.path.join(d, "elsewhere", "train"))
    assert COCODataset(d, image_dir=os.path.join(d, "elsewhere"), split="train")[0]["image_features"].shape == (5, 2052)


def test_caption_and_reason_datasets(tmp_path):
    d = str(tmp_path)
    recs = write_synthetic_split(d, "val", n_images=2, records_per_image=1, regions=4, seed=5)
    for cls in (SBUDataset, CCDataset):
        e = cls(d, split="val")[1]
        assert e["task_type"] == TaskType.CAPTION and e["labels"] == recs[1]["labels"].strip()
    write_synthetic_split(d, "val", n_images=2, records_per_image=1, regions=4, seed=5, reason=True)
    ds = ReasonDataset(d, split="val")
    e = ds[1]
    assert e["dataset_index"] == 1 and e["image_features"].shape == (4, 2052) and ds.get_raw_data(1)["index"] == 1
    assert ReasonDataset(d, split="val", use_event=False)[0]["event"] == ""
    os.remove(os.path.join(d, "val", e["img_id"] + ".pkl"))
    assert ds[1] is None                                                         # missing feature file -> dropped
    out = Collator(_tok())([ds[0], ds[1]])
    assert out["input_ids"].shape[0] == 1 and out["dataset_index"] == [0]


def test_visual_genome_dataset_feeds_attribute_and_relation_heads(tmp_path):
    d = str(tmp_path)
    regions = write_synthetic_vg(d, "train", n_images=3, objects=4, regions_per_image=2, seed=2)
    ds = VGDataset(d, split="train")
    assert len(ds) == len(regions) == 6
    e = ds[3]
    n_obj = len(e["object_ids"])
    assert e["task_type"] == TaskType.REGION_CAPTION and e["labels"] == regions[3]["description"]
    assert e["image_features"].shape == (n_obj + 2, 2052) and e["mrm_labels"].shape == (n_obj + 2, 1601)
    blob = pickle.load(open(os.path.join(d, "train", "%d.pkl" % e["img_id"]), "rb"))
    k = blob["region_ids"].index(regions[3]["region_id"])
    assert np.array_equal(e["image_features"][0, :2048], blob["image_feature"])          # whole image first
    assert np.array_equal(e["image_features"][1:-1, 2048:], blob["object_boxes"])        # then the objects
    assert np.array_equal(e["image_features"][-1, :2048], blob["region_features"][k])    # the described region last
    tok = _tok()
    torch.manual_seed(0)
    out = Collator(tok, mlm_enabled=True, mrm_enabled=True, ap_enabled=True, rp_enabled=True, mlm_probability=0.2,
                   mrm_probability=0.2)([ds[0], ds[3]])
    for i, ent in enumerate((ds[0], ds[3])):
        with_attr = [o for o in ent["objects"] if "attribute_ids" in o]
        assert out["attribute_labels"][i].tolist() == [o["attribute_ids"][0] for o in with_attr]
        assert int(out["attribute_mask"][i].sum()) == len(with_attr)
        first_obj = int((out["decoder_input_ids"][i] == tok.begin_img_id).nonzero()[0]) + 2 - 0
        for r in out["relation_labels"][i]:
            assert first_obj <= r["object_index"] < first_obj + len(ent["object_ids"])
    assert sum(t.shape[0] for t in out["mrm_labels"]) == int(out["mrm_mask"].sum())


def test_dataloader_pipeline(tmp_path):
    d = str(tmp_path)
    write_synthetic_split(d, "train", n_images=4, records_per_image=2, regions=[36, 20, 7, 30], seed=9)
    tok = _tok()
    loader = torch.utils.data.DataLoader(VCGDataset(d, split="train"), batch_size=4, shuffle=False, num_workers=0,
                                         collate_fn=Collator(tok, event_max_len=20, lm_max_len=30, max_img_num=30))
    batches = list(loader)
    assert len(batches) == 2
    for b in batches:
        counts = [int(x) for x in (b["input_ids"] == tok.img_feat_id).sum(1)]
        assert counts == [f.shape[0] if f.dim() == 2 else 0 for f in b["image_features"]]
        assert max(counts) <= 30                                                  # max_img_num clips 36 -> 30
        assert b["labels"].shape == b["decoder_input_ids"].shape == b["decoder_attention_mask"].shape
        assert int((b["labels"] == tok.eos_token_id).sum()) == 4
        assert torch.equal(b["decoder_input_ids"][:, 0], torch.full((4,), tok.bos_token_id))


def test_pretrain_cli_arguments_and_dataset_list(tmp_path):
    """pretrain.py flag checks (reference pretrain.py:414-434) and the NAME -> dataset class mapping (:128-247)"""
    import importlib.util
    import pytest
    spec = importlib.util.spec_from_file_location("kmb_pretrain_cli", os.path.join(ROOT, "km-bart_amd", "pretrain.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    base = ["--checkpoint_dir", str(tmp_path), "--model_config", "x.json"]
    with pytest.raises(ValueError, match="repeated"):
        cli.parse_args(base + ["--dataset", "coco_train", "a", "--dataset", "coco_train", "b"])
    with pytest.raises(ValueError, match="not a valid dataset"):
        cli.parse_args(base + ["--dataset", "imagenet", "a"])
    with pytest.raises(ValueError, match="VG"):
        cli.parse_args(base + ["--dataset", "vg_train", "a", "--no_image"])
    with pytest.raises(ValueError, match="cannot be empty"):
        cli.parse_args(["--checkpoint_dir", str(tmp_path), "--synthetic", "1"])
    with pytest.raises(ValueError):
        cli.parse_args(base)                                   # neither --dataset nor --synthetic
    d = str(tmp_path)
    write_synthetic_split(os.path.join(d, "c"), "train", n_images=2, records_per_image=1, regions=3)
    write_synthetic_split(os.path.join(d, "r"), "val", n_images=2, records_per_image=1, regions=3, reason=True)
    write_synthetic_vg(os.path.join(d, "g"), "train", n_images=1)
    args = cli.parse_args(base + ["--dataset", "vg_train", os.path.join(d, "g"), "--dataset", "vcg_train",
                                  os.path.join(d, "c"), "--dataset", "sbu_reason_val", os.path.join(d, "r"),
                                  "--dataset", "cc_train", os.path.join(d, "c")])
    assert args.mrm_enabled and args.ap_enabled and args.rp_enabled and args.mlm_probability == 0.2
    kinds = [type(x).__name__ for x in cli.build_datasets(args)]
    assert kinds == ["ReasonDataset", "VGDataset", "CCDataset", "VCGDataset"]   # the reference's fixed order
    assert cli.build_datasets(args)[3][0]["task_type"] == TaskType.CAPTION       # vcg_train is used with pretrain=True
