"""Generation driver: counterpart of the reference's vcg_generate.py (flags of vcg_generate.py:71-123).
Loads a checkpoint, runs generate_text over a loader and writes the `{index, task_type, generations}` JSON.
`--data_dir DIR` reads "<split>_eval.json" + feature pickles in the reference's format (tokenizer: the BART-large files
on disk or `--tokenizer_json`); `--synthetic N` needs no files and writes the generated ids un-decoded."""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import torch  # noqa: E402

from src.data.synthetic import make_batch  # noqa: E402
from src.generation import generate_text  # noqa: E402
from src.model import MultiModalBartForConditionalGeneration  # noqa: E402
from src.utils import Logger  # noqa: E402


class IdTokenizer:
    """Stand-in when the BART vocabulary files are not on disk: 'decodes' to space-separated ids."""

    def decode(self, seq, skip_special_tokens=True):
        keep = [int(t) for t in seq if not (skip_special_tokens and int(t) in (0, 1, 2))]
        return " ".join(map(str, keep))


def main(args):
    device = torch.device("cuda", 0)
    logger = Logger(args.log_dir)
    model = MultiModalBartForConditionalGeneration.from_pretrained(args.checkpoint)
    model.to(device)
    tokenizer = IdTokenizer()
    loader = []
    if args.synthetic > 0:
        for i in range(args.synthetic):
            b = make_batch(args.batch_size, seed=4321 + i, num_regions=36 if args.use_image else 0,
                           event_lens=None if args.use_image else [59] * args.batch_size)
            b["index"] = [i * args.batch_size + j for j in range(args.batch_size)]
            loader.append(b)
    else:   # vcg_generate.py:37-57: the eval split (one record per image), no labels
        from torch.utils.data import DataLoader
        from src.data.collation import Collator
        from src.data.dataset import VCGDataset
        from src.data.offline_tokenizer import load_base_tokenizer
        from src.data.tokenization import ConditionTokenizer
        tokenizer = ConditionTokenizer(base_tokenizer=load_base_tokenizer(args.tokenizer_json or "facebook/bart-large"))
        dataset = VCGDataset(args.data_dir, split=args.split, use_image=args.use_image, use_event=args.use_event,
                             eval_mode=True)
        loader = DataLoader(dataset, batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers,
                            collate_fn=Collator(tokenizer, has_label=False, pin_memory=args.num_workers == 0))
    generated = generate_text(model, loader, tokenizer, args, device, logger=logger)
    with open(args.output_file, "w") as f:
        json.dump(generated, f)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--data_dir", default=None, type=str)
    p.add_argument("--output_file", required=True, type=str)
    p.add_argument("--checkpoint", required=True, type=str)
    p.add_argument("--log_dir", default=None, type=str)
    p.add_argument("--split", default="val", type=str)
    p.add_argument("--no_event", dest="use_event", action="store_false")
    p.add_argument("--no_image", dest="use_image", action="store_false")
    p.add_argument("--model", type=str, default="base")
    p.add_argument("--num_gen", default=1, type=int)
    p.add_argument("--num_beams", default=1, type=int)
    p.add_argument("--do_sample", action="store_true")
    p.add_argument("--top_p", default=1.0, type=float)
    p.add_argument("--top_k", default=0, type=int)
    p.add_argument("--gpu_num", default=1, type=int)
    p.add_argument("--cpu", action="store_true")
    p.add_argument("--amp", action="store_true")
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--num_workers", type=int, default=0)
    p.add_argument("--synthetic", type=int, default=0)
    p.add_argument("--tokenizer_json", default=None, type=str)
    p.set_defaults(use_event=True, use_image=True)
    args = p.parse_args(argv)
    if args.cpu:
        raise ValueError("--cpu: this build has no CPU path")
    if args.synthetic <= 0 and args.data_dir is None:
        raise ValueError("give --data_dir or --synthetic N")
    return args


if __name__ == "__main__":
    main(parse_args())
