"""VCG fine-tuning driver: the counterpart of the reference's vcg_train.py (flags of vcg_train.py:272-344)
on the MI355X engine.  One process per GPU (`--gpu_num N` spawns N ranks, reference vcg_train.py:350-355),
RCCL gradient all-reduce, fused AdamW.

`--data_dir DIR` reads the reference's on-disk format (src/data/dataset.py) through the Collator into packed, pinned
batches that a side stream copies one step ahead; the BART-large vocabulary is needed for real data
(`--tokenizer_json FILE` loads any `tokenizers` JSON instead -- there is no network here).  `--synthetic N` trains on
N synthetic VCG batches per epoch with no files at all.
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

from kmbart.optim import AdamW  # noqa: E402
from kmbart.parallel import DistributedDataParallel  # noqa: E402
from kmbart.data import DevicePrefetcher  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402
from src.training import fine_tune  # noqa: E402
from src.utils import Logger, cleanup_process, load_training_data, save_training_data, setup_process  # noqa: E402


class SyntheticLoader:
    """len() batches per epoch; rank r draws seed 1234 + r (DistributedSampler-style disjoint shards)."""

    def __init__(self, n_batches, batch_size, rank, use_image=True):
        self.n, self.bs, self.rank, self.use_image = n_batches, batch_size, rank, use_image

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            b = make_batch(self.bs, seed=(1234 + self.rank) * 100003 + i, num_regions=36 if self.use_image else 0,
                           event_lens=None if self.use_image else [59] * self.bs)
            yield b


def build_vcg_loader(args, rank, device, split="train"):
    """VCGDataset -> DistributedSampler -> DataLoader(Collator) -> DevicePrefetcher (vcg_train.py:115-142); the
    validation split is not sharded (vcg_train.py:144-160: rank 0 validates alone)"""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from src.data.collation import Collator
    from src.data.dataset import VCGDataset
    from src.data.offline_tokenizer import load_base_tokenizer
    from src.data.tokenization import ConditionTokenizer
    base = load_base_tokenizer(args.tokenizer_json or "facebook/bart-large")
    tokenizer = ConditionTokenizer(base_tokenizer=base)
    dataset = VCGDataset(args.data_dir, split=split, use_image=args.use_image, use_event=args.use_event)
    sampler = DistributedSampler(dataset, num_replicas=args.gpu_num, rank=rank) if split == "train" else None
    loader = DataLoader(dataset, batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers,
                        sampler=sampler, collate_fn=Collator(tokenizer, has_label=True, pin_memory=args.num_workers == 0))
    return DevicePrefetcher(loader, device)


def main(rank, args):
    distributed = args.gpu_num > 1
    if distributed:
        setup_process(rank, args.gpu_num, master_port=args.master_port)
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    logger = Logger(args.log_dir, enabled=(rank == 0))
    logger.info("Loading model...")
    config = None
    if args.model_config is not None:
        with open(args.model_config) as f:
            config = MultiModalBartConfig.from_dict(json.load(f))
        for k in ("dropout", "classif_dropout", "attention_dropout", "activation_dropout"):
            if getattr(args, k) is not None:
                setattr(config, k, getattr(args, k))
    if args.checkpoint:
        model = MultiModalBartForConditionalGeneration.from_pretrained(args.checkpoint, config=config,
                                                                       error_on_mismatch=False)
    else:
        model = MultiModalBartForConditionalGeneration(config=config)
    model.to(device)
    if distributed:
        model = DistributedDataParallel(model, device_ids=[rank], find_unused_parameters=True)
    optimizer = AdamW(model.parameters(), lr=args.lr)
    start_epoch = 0
    if args.continue_training:
        start_epoch = load_training_data(args.checkpoint, optimizer=optimizer, map_location="cpu")["epoch"] + 1
    if args.synthetic > 0:
        loader = SyntheticLoader(args.synthetic, args.batch_size, rank, use_image=args.use_image)
    else:
        loader = build_vcg_loader(args, rank, device)
    for epoch in range(start_epoch, args.epochs):
        logger.info("Epoch {}".format(epoch + 1), pad=True)
        fine_tune(epoch, model, loader, optimizer, device, args, logger=logger, log_interval=args.log_interval)
        if rank == 0:
            inner = model.module if distributed else model
            if args.validate_loss and args.synthetic <= 0:
                from src.validation import validate_fine_tune_loss
                validate_fine_tune_loss(epoch, inner, build_vcg_loader(args, rank, device, split="val"), device, args,
                                        logger=logger, log_interval=args.log_interval)
            out = os.path.join(args.checkpoint_dir, "epoch{}".format(epoch + 1))
            inner.save_pretrained(out)
            save_training_data(out, optimizer=optimizer, epoch=epoch)
            logger.info("Saved checkpoint to " + out)
    if distributed:
        cleanup_process()


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--data_dir", default=None, type=str)
    p.add_argument("--checkpoint_dir", required=True, type=str)
    p.add_argument("--log_dir", default=None, type=str)
    p.add_argument("--model_config", default=None, type=str)
    p.add_argument("--checkpoint", default=None, type=str)
    p.add_argument("--no_event", dest="use_event", action="store_false")
    p.add_argument("--no_image", dest="use_image", action="store_false")
    p.add_argument("--epochs", default=40, type=int)
    p.add_argument("--lr", default=1e-5, type=float)
    p.add_argument("--num_gen", default=1, type=int)
    p.add_argument("--num_beams", default=1, type=int)
    p.add_argument("--continue_training", action="store_true")
    p.add_argument("--validate_loss", action="store_true")
    p.add_argument("--validate_score", action="store_true")
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
    p.add_argument("--synthetic", type=int, default=0, help="train on N synthetic VCG batches per epoch")
    p.add_argument("--tokenizer_json", default=None, type=str,
                   help="a `tokenizers` JSON to use instead of the facebook/bart-large vocabulary files")
    p.add_argument("--log_interval", type=int, default=1)
    p.set_defaults(use_event=True, use_image=True)
    args = p.parse_args(argv)
    if args.cpu:
        raise ValueError("--cpu: this build has no CPU path (the hot path runs on MI355X only)")
    if args.checkpoint is None and args.model_config is None:
        raise ValueError("--model_config and --checkpoint cannot be empty at the same time")
    if args.synthetic <= 0 and args.data_dir is None:
        raise ValueError("give --data_dir (the reference's dataset format) or --synthetic N")
    return args


if __name__ == "__main__":
    a = parse_args()
    if a.gpu_num > 1:
        mp.spawn(main, args=(a,), nprocs=a.gpu_num, join=True)
    else:
        main(0, a)
