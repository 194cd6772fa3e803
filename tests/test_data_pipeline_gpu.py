"""The whole path a `vcg_train.py --data_dir` / `pretrain.py --dataset` run takes:
files in the reference's format -> dataset -> Collator (packed, pinned) -> DevicePrefetcher -> HIP engine, with the
losses of the collated batch checked against the CPU oracle on the same batch."""
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "km-bart_amd"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu

from oracle import goldenlib as G  # noqa: E402
from oracle import kmbart_oracle as O  # noqa: E402

DEV = "cuda:0"
GOLD = os.path.join(ROOT, "tests", "golden")


def _setup(tmp_path):
    from src.data.dataset import VCGDataset, VGDataset, write_synthetic_split, write_synthetic_vg
    from src.data.offline_tokenizer import load_base_tokenizer
    from src.data.tokenization import ConditionTokenizer
    tok = ConditionTokenizer(base_tokenizer=load_base_tokenizer(os.path.join(GOLD, "tiny_bpe_tokenizer.json")))
    d = str(tmp_path)
    write_synthetic_split(os.path.join(d, "vcg"), "train", n_images=4, records_per_image=2, regions=[6, 3, 0, 9], seed=1)
    write_synthetic_vg(os.path.join(d, "vg"), "train", n_images=2, objects=3, regions_per_image=2, seed=2,
                       num_attributes=11, num_relations=9)
    return tok, VCGDataset(os.path.join(d, "vcg"), split="train"), VGDataset(os.path.join(d, "vg"), split="train")


def _config(tok, **over):
    from test_model_gpu import cfg_from_oracle
    ocfg = G.tiny_config(img_feat_id=tok.img_feat_id, cls_token_id=tok.cls_token_id, **over)
    assert len(tok) <= ocfg.vocab_size
    return ocfg, cfg_from_oracle(ocfg, **over)


def test_fine_tune_from_files(tmp_path):
    from kmbart.data import DevicePrefetcher, PackedFeatures
    from kmbart.optim import AdamW
    from src.data.collation import Collator
    from src.model import MultiModalBartForConditionalGeneration
    from src.training import fine_tune
    tok, vcg, _ = _setup(tmp_path)
    ocfg, cfg = _config(tok)
    sd = G.golden_state_dict(ocfg, seed=5)
    model = MultiModalBartForConditionalGeneration(cfg)
    model.load_state_dict({k: v for k, v in sd.items() if k in model.state_dict()}, strict=False)
    model.to(DEV).eval()
    loader = torch.utils.data.DataLoader(vcg, batch_size=4, shuffle=False,
                                         collate_fn=Collator(tok, has_label=True, pin_memory=True, max_img_num=8))
    batches = list(loader)
    assert isinstance(batches[0]["image_features"], PackedFeatures)
    for b in batches:   # loss of every collated batch against the oracle on the very same tensors
        ref, _, _ = O.forward(sd, ocfg, b["input_ids"], b["image_features"].as_list(), b["attention_mask"],
                              b["decoder_input_ids"], b["decoder_attention_mask"], b["labels"])
        got = model(input_ids=b["input_ids"].to(DEV), image_features=b["image_features"].to(DEV),
                    attention_mask=b["attention_mask"].to(DEV), decoder_input_ids=b["decoder_input_ids"].to(DEV),
                    decoder_attention_mask=b["decoder_attention_mask"].to(DEV), labels=b["labels"].to(DEV))[0]
        assert abs(float(got) - float(ref)) <= 2e-3 * abs(float(ref)), (float(got), float(ref))
    # validation loop (reference src/validation.py:62-121): mean of the per-batch losses, eval mode
    from src.validation import validate_fine_tune_loss
    refs = [float(O.forward(sd, ocfg, b["input_ids"], b["image_features"].as_list(), b["attention_mask"],
                            b["decoder_input_ids"], b["decoder_attention_mask"], b["labels"])[0]) for b in batches]
    val = validate_fine_tune_loss(0, model, loader, DEV, types.SimpleNamespace(amp=False))
    assert abs(val - sum(refs) / len(refs)) <= 2e-3 * abs(val)
    # the training loop over the prefetcher (copies of batch i+1 overlap step i)
    args = types.SimpleNamespace(epochs=1, amp=False)
    model.train()
    opt = AdamW(model.parameters(), lr=1e-3)
    first = float(model(**{k: (v.to(DEV) if hasattr(v, "to") else v) for k, v in batches[0].items()
                           if k in ("input_ids", "image_features", "attention_mask", "decoder_input_ids",
                                    "decoder_attention_mask", "labels")})[0])
    for _ in range(3):
        fine_tune(0, model, DevicePrefetcher(loader, DEV), opt, DEV, args)
    model.eval()
    after = float(model(input_ids=batches[0]["input_ids"].to(DEV), image_features=batches[0]["image_features"].to(DEV),
                        attention_mask=batches[0]["attention_mask"].to(DEV),
                        decoder_input_ids=batches[0]["decoder_input_ids"].to(DEV),
                        decoder_attention_mask=batches[0]["decoder_attention_mask"].to(DEV),
                        labels=batches[0]["labels"].to(DEV))[0])
    assert after < first


def test_pretrain_batch_from_files_against_oracle(tmp_path):
    from kmbart.optim import AdamW
    from src.data.collation import Collator
    from src.model import MultiModalBartForPreTraining
    from src.training import pretrain
    tok, vcg, vg = _setup(tmp_path)
    heads = dict(num_labels=1601, num_attributes=11, num_relations=9)
    ocfg, cfg = _config(tok, **heads)
    sd = G.golden_state_dict(ocfg, seed=6)
    collate = Collator(tok, mlm_enabled=True, mrm_enabled=True, ap_enabled=True, rp_enabled=True, mlm_probability=0.3,
                       mrm_probability=0.4, max_img_num=8, lm_max_len=10)
    torch.manual_seed(3)
    entries = [vg[0], dict(vcg[0], mrm_labels=vcg[0]["mrm_labels"]), vg[3], vcg[6]]
    b = collate(entries)
    assert sum(t.shape[0] for t in b["mrm_labels"]) > 0 and sum(len(r) for r in b["relation_labels"]) > 0
    ref, _ = O.pretrain_forward(sd, ocfg, b["input_ids"], b["image_features"].as_list(), b["attention_mask"],
                                b["decoder_input_ids"], b["decoder_attention_mask"], b["labels"], b["mrm_labels"],
                                b["mrm_mask"], b["attribute_labels"], b["attribute_mask"], b["relation_labels"])
    model = MultiModalBartForPreTraining(cfg)
    model.load_state_dict(sd, strict=False)
    model.to(DEV).eval()
    out = model(input_ids=b["input_ids"].to(DEV), image_features=b["image_features"].to(DEV),
                attention_mask=b["attention_mask"].to(DEV), decoder_input_ids=b["decoder_input_ids"].to(DEV),
                decoder_attention_mask=b["decoder_attention_mask"].to(DEV), labels=b["labels"].to(DEV),
                mrm_labels=b["mrm_labels"], mrm_mask=b["mrm_mask"], attribute_labels=b["attribute_labels"],
                attribute_mask=b["attribute_mask"], relation_labels=b["relation_labels"])[0]
    for k in ("loss", "lm_loss", "mrm_loss", "attribute_loss", "relation_loss"):
        assert abs(float(out[k]) - float(ref[k])) <= 3e-3 * abs(float(ref[k])) + 1e-4, (k, float(out[k]), float(ref[k]))
    # and the loop itself runs on collated batches
    loader = torch.utils.data.DataLoader(torch.utils.data.ConcatDataset([vg]), batch_size=2, shuffle=False,
                                         collate_fn=collate)
    model.train()
    mean_loss = pretrain(0, model, loader, AdamW(model.parameters(), lr=1e-4), DEV,
                         types.SimpleNamespace(epochs=1, amp=False))
    assert mean_loss == mean_loss and mean_loss > 0


def test_feature_list_is_packed_in_one_launch_like_a_concatenation():
    """The reference's batch carries region features as a LIST of per-sample [R_i, 2052] tensors (src/data/collation.py:73-76; the model
    concatenates the non-empty ones, src/model/modules.py:24-41).  kmbart.engine.pack_features gathers them with kmb_pack_features -- one
    launch per 128 tensors instead of a concatenation that costs ~one blit per tensor on this stack -- and must return exactly the
    concatenation: ragged counts, empty samples (torch.empty(0)), more than 128 tensors, a non-contiguous and a CPU-resident entry."""
    from kmbart.engine import pack_features
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    for counts in ((36, 20, 0, 7, 36), (5,), (0, 0), tuple(int(x) for x in torch.randint(0, 9, (300,), generator=g))):
        feats = []
        for i, n in enumerate(counts):
            if n == 0:
                feats.append(torch.empty(0))
                continue
            t = torch.randn((n, 2052), generator=g)
            if i % 5 == 1:
                t = torch.randn((n, 4104), generator=g)[:, ::2]          # non-contiguous view
            feats.append(t if i % 7 == 3 else t.to(dev))                # some stay on the host
        packed, offsets, ntot = pack_features(feats, 2052, dev)
        torch.cuda.synchronize()
        non_empty = [f.to(dev).float() for f in feats if f.dim() == 2 and f.shape[0] > 0]
        assert ntot == sum(counts)
        assert offsets.cpu().tolist() == [sum(counts[:i]) for i in range(len(counts) + 1)]
        if non_empty:
            want = torch.cat(non_empty, 0)
            assert packed.shape == want.shape and torch.equal(packed, want), counts[:8]
    with pytest.raises(ValueError):
        pack_features([torch.randn(3, 2048).to(dev), torch.randn(2, 2052).to(dev)], 2052, dev)
