"""`generate_text` with the reference's contract (reference src/generation.py:6-52): run
model.generate over a loader and return `{index, task_type, generations}` records."""
from datetime import datetime


def _features(feats, device):
    """list of per-sample tensors (reference collator) or a kmbart.data.PackedFeatures (two copies, not B)"""
    return feats.to(device) if hasattr(feats, "packed") else [f.to(device) for f in feats]


def generate_text(model, gen_loader, tokenizer, args, device, logger=None, log_interval=1):
    n_steps = len(gen_loader)
    model.eval()
    t0 = datetime.now()
    records = []
    num_gen = args.num_gen
    for i, batch in enumerate(gen_loader):
        out = model.generate(
            input_ids=batch["input_ids"].to(device),
            image_features=_features(batch["image_features"], device),
            attention_mask=batch["attention_mask"].to(device),
            num_beams=args.num_beams,
            num_return_sequences=num_gen,
            do_sample=getattr(args, "do_sample", False),
            top_p=getattr(args, "top_p", 1.0),
            top_k=getattr(args, "top_k", 0),
            early_stopping=True,
        )
        for j, index in enumerate(batch["index"]):
            texts = [tokenizer.decode(seq, skip_special_tokens=True) for seq in out[j * num_gen:(j + 1) * num_gen]]
            records.append({"index": index, "task_type": batch["task_type"][j], "generations": texts})
        if logger is not None and (i + 1) % log_interval == 0:
            eta = (n_steps - (i + 1)) / (i + 1) * (datetime.now() - t0)
            logger.info("Generating, Step [{}/{}], ETA: {}".format(i + 1, n_steps, str(eta)))
    return records
