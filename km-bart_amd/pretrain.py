"""Multi-task pre-training driver: the counterpart of the reference's pretrain.py (flags of pretrain.py:336-434) on the
MI355X engine -- denoising LM + masked-region modelling + attribute / relation prediction (BASELINE config 4).

    python pretrain.py --model_config pretrain_base.json --checkpoint_dir CKPT \
        --dataset coco_train DIR --dataset vg_train DIR [--tokenizer_json vocab.json]
    python pretrain.py --model_config pretrain_base.json --checkpoint_dir CKPT --synthetic 100

`--dataset NAME PATH` takes the reference's dataset names (pretrain.py:23-28) and on-disk format; the corpora are
concatenated, sharded by a DistributedSampler, collated into packed pinned batches and copied one step ahead.
`--synthetic N` trains on N synthetic batches per epoch (no files; SURVEY.md section 8d, config 4: 50 regions).
One process per GPU (`--gpu_num N`), RCCL gradient all-reduce, fused AdamW; there is no CPU path.
"""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import torch  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from kmbart.data import DevicePrefetcher  # noqa: E402
from kmbart.optim import AdamW  # noqa: E402
from kmbart.parallel import DistributedDataParallel  # noqa: E402
from src.data.synthetic import make_pretrain_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForPreTraining  # noqa: E402
from src.training import pretrain  # noqa: E402
from src.utils import Logger, cleanup_process, load_training_data, save_training_data, setup_process  # noqa: E402

DATASET_NAMES = (
    "coco_train", "coco_val", "coco_reason_train", "coco_reason_val", "sbu_train", "sbu_val", "sbu_reason_train",
    "sbu_reason_val", "vg_train", "vg_val", "cc_train", "cc_val", "cc_reason_train", "cc_reason_val", "vcg_train",
    "vcg_reason_train")


class SyntheticPretrainLoader:
    def __init__(self, n_batches, batch_size, rank, regions):
        self.n, self.bs, self.rank, self.regions = n_batches, batch_size, rank, regions

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield make_pretrain_batch(self.bs, seed=(1234 + self.rank) * 100003 + i, num_regions=self.regions)


def build_datasets(args):
    """one dataset object per --dataset NAME PATH, by the reference's naming scheme (pretrain.py:128-247)"""
    from src.data.dataset import CCDataset, COCODataset, ReasonDataset, SBUDataset, VCGDataset, VGDataset
    plain = {"sbu": SBUDataset, "coco": COCODataset, "cc": CCDataset}
    out = []
    for name in DATASET_NAMES:              # fixed order, as the reference's chain of ifs
        if name not in args.dataset:
            continue
        path = args.dataset[name]
        corpus, _, split = name.rpartition("_")
        if corpus.endswith("_reason"):
            out.append(ReasonDataset(path, split=split, use_image=args.use_image, use_event=args.use_event))
        elif corpus == "vg":
            out.append(VGDataset(path, split=split))
        elif corpus == "vcg":
            out.append(VCGDataset(path, split=split, use_image=args.use_image, pretrain=True))
        else:
            out.append(plain[corpus](path, split=split, use_image=args.use_image))
    return out


def build_loader(args, rank, device):
    from torch.utils.data import ConcatDataset, DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from src.data.collation import Collator
    from src.data.offline_tokenizer import load_base_tokenizer
    from src.data.tokenization import ConditionTokenizer
    tokenizer = ConditionTokenizer(base_tokenizer=load_base_tokenizer(args.tokenizer_json or "facebook/bart-large"))
    collate = Collator(tokenizer, mlm_enabled=True, mlm_probability=args.mlm_probability,
                       mrm_enabled=args.mrm_enabled, mrm_probability=args.mrm_probability, ap_enabled=args.ap_enabled,
                       rp_enabled=args.rp_enabled, lm_max_len=args.lm_max_len, max_img_num=args.max_img_num,
                       pin_memory=args.num_workers == 0)   # worker processes must not touch the GPU runtime
    dataset = ConcatDataset(build_datasets(args))
    sampler = DistributedSampler(dataset, num_replicas=args.gpu_num, rank=rank)
    loader = DataLoader(dataset, batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers,
                        sampler=sampler, collate_fn=collate)
    return DevicePrefetcher(loader, device)


def main(rank, args):
    distributed = args.gpu_num > 1
    if distributed:
        setup_process(rank, args.gpu_num, master_port=args.master_port)
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    logger = Logger(args.log_dir, enabled=(rank == 0))
    logger.info("Loading model...")
    if args.model_config is not None:
        with open(args.model_config) as f:
            config = MultiModalBartConfig.from_dict(json.load(f))
    else:
        config = MultiModalBartConfig.from_pretrained(args.checkpoint)
    for k in ("dropout", "classif_dropout", "attention_dropout", "activation_dropout"):
        if getattr(args, k) is not None:
            setattr(config, k, getattr(args, k))
    if args.checkpoint:
        model = MultiModalBartForPreTraining.from_pretrained(args.checkpoint, config=config, error_on_mismatch=False)
    else:
        model = MultiModalBartForPreTraining(config)
    model.to(device)
    if distributed:
        model = DistributedDataParallel(model, device_ids=[rank], find_unused_parameters=True)
    optimizer = AdamW(model.parameters(), lr=args.lr)
    epoch = 0
    if args.continue_training:
        epoch = load_training_data(args.checkpoint, optimizer=optimizer, map_location="cpu")["epoch"] + 1
    logger.info("Loading data...")
    if args.synthetic > 0:
        loader = SyntheticPretrainLoader(args.synthetic, args.batch_size, rank, args.max_img_num)
    else:
        loader = build_loader(args, rank, device)
    logger.info("Start training", pad=True)
    while epoch < args.epochs:
        logger.info("Epoch {}".format(epoch + 1), pad=True)
        pretrain(epoch=epoch, model=model, train_loader=loader, optimizer=optimizer, args=args, device=device,
                 logger=logger, log_interval=args.log_interval)
        if rank == 0:
            out = os.path.join(args.checkpoint_dir, "model{}".format(epoch))
            (model.module if distributed else model).save_pretrained(out)
            save_training_data(out, optimizer=optimizer, epoch=epoch)
            logger.info('Saved checkpoint at "{}"'.format(out))
        epoch += 1
    if distributed:
        cleanup_process()


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--dataset", action="append", nargs=2, metavar=("DATASET_NAME", "DATASET_PATH"), default=None,
                   help='append a dataset, one of "{}"'.format('", "'.join(DATASET_NAMES)))
    p.add_argument("--checkpoint_dir", required=True, type=str)
    p.add_argument("--log_dir", default=None, type=str)
    p.add_argument("--model_config", default=None, type=str)
    p.add_argument("--checkpoint", default=None, type=str)
    p.add_argument("--no_event", dest="use_event", action="store_false")
    p.add_argument("--no_image", dest="use_image", action="store_false")
    p.add_argument("--no_mrm", dest="mrm_enabled", action="store_false")
    p.add_argument("--no_ap", dest="ap_enabled", action="store_false")
    p.add_argument("--no_rp", dest="rp_enabled", action="store_false")
    p.add_argument("--epochs", default=40, type=int)
    p.add_argument("--lr", default=1e-5, type=float)
    p.add_argument("--num_gen", default=1, type=int)
    p.add_argument("--num_beams", default=1, type=int)
    p.add_argument("--continue_training", action="store_true")
    p.add_argument("--validate_loss", action="store_true")
    p.add_argument("--validate_score", action="store_true")
    p.add_argument("--max_img_num", type=int, default=30)
    p.add_argument("--lm_max_len", type=int, default=30)
    p.add_argument("--mrm_probability", type=float, default=0.2)
    p.add_argument("--mlm_probability", type=float, default=0.2)
    p.add_argument("--dropout", default=None, type=float)
    p.add_argument("--classif_dropout", default=None, type=float)
    p.add_argument("--attention_dropout", default=None, type=float)
    p.add_argument("--activation_dropout", default=None, type=float)
    p.add_argument("--gpu_num", default=1, type=int)
    p.add_argument("--cpu", action="store_true")
    p.add_argument("--amp", action="store_true")
    p.add_argument("--master_port", type=str, default="12355")
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--num_workers", type=int, default=0)
    p.add_argument("--synthetic", type=int, default=0, help="train on N synthetic pre-training batches per epoch")
    p.add_argument("--tokenizer_json", default=None, type=str,
                   help="a `tokenizers` JSON to use instead of the facebook/bart-large vocabulary files")
    p.add_argument("--log_interval", type=int, default=1)
    p.set_defaults(use_event=True, use_image=True, mrm_enabled=True, rp_enabled=True, ap_enabled=True)
    args = p.parse_args(argv)
    if args.cpu:
        raise ValueError("--cpu: this build has no CPU path (the hot path runs on MI355X only)")
    if args.checkpoint is None and args.model_config is None:
        raise ValueError("--model_config and --checkpoint cannot be empty at the same time")
    if args.synthetic <= 0 and not args.dataset:
        raise ValueError("give at least one --dataset NAME PATH, or --synthetic N")
    pairs = args.dataset or []
    names = [k for k, _ in pairs]
    if len(names) != len(set(names)):
        raise ValueError("repeated datasets")
    for name in names:
        if name not in DATASET_NAMES:
            raise ValueError('"{}" is not a valid dataset'.format(name))
    args.dataset = dict(pairs)
    if ("vg_val" in args.dataset or "vg_train" in args.dataset) and not args.use_image:
        raise ValueError("--no_image can not be set while using VG dataset")
    return args


if __name__ == "__main__":
    a = parse_args()
    if a.gpu_num > 1:
        mp.spawn(main, args=(a,), nprocs=a.gpu_num, join=True)
    else:
        main(0, a)
