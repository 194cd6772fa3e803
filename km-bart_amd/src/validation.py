"""Validation-loss loops with the reference's contract (reference src/validation.py:10-121): eval mode, forward only,
running mean logged as `Computing validation loss, Step [i/N], Loss: x.xxxx, ETA: ...`, epoch summary and an optional
TensorBoard scalar.  The generation-score validation (BLEU / METEOR / CIDEr, validation.py:124-165) depends on the
evaluation package and Java tools and is outside the hot path; `generate_text` is what it calls."""
from datetime import datetime

import torch

from src.generation import generate_text
from src.training import _features, _on


def _run(epoch, model, val_loader, device, args, logger, log_interval, tb_writer, forward):
    n_steps = len(val_loader)
    model.eval()
    total = 0.0
    t0 = datetime.now()
    with torch.no_grad():   # the engine keeps no backward state when grad mode is off
        for i, batch in enumerate(val_loader):
            total += forward(batch)
            if logger is not None and i % log_interval == 0:
                eta = (n_steps - (i + 1)) / (i + 1) * (datetime.now() - t0)
                logger.info("Computing validation loss, Step [{}/{}], Loss: {:.4f}, ETA: {}".format(
                    i + 1, n_steps, total / (i + 1), str(eta)))
    mean = total / max(n_steps, 1)
    if logger is not None:
        logger.info("Validation loss", pad=True)
        logger.info("Epoch: {}, Val loss: {}".format(epoch + 1, mean))
        logger.line()
    if tb_writer is not None:
        tb_writer.add_scalars("loss/epoch", {"val": mean}, epoch + 1)
    return mean


def validate_fine_tune_loss(epoch, model, val_loader, device, args, logger=None, log_interval=1, tb_writer=None):
    def forward(batch):
        out = model.forward(
            input_ids=batch["input_ids"].to(device), image_features=_features(batch["image_features"], device),
            attention_mask=batch["attention_mask"].to(device), decoder_input_ids=_on(batch, "decoder_input_ids", device),
            decoder_attention_mask=_on(batch, "decoder_attention_mask", device), labels=_on(batch, "labels", device),
            answer_ids=_on(batch, "answer_ids", device), answer_attention_mask=_on(batch, "answer_attention_mask", device))
        return out[0].item()
    return _run(epoch, model, val_loader, device, args, logger, log_interval, tb_writer, forward)


def validate_pretraining_loss(epoch, model, val_loader, device, args, logger=None, log_interval=1, tb_writer=None):
    def forward(batch):
        out = model.forward(
            input_ids=batch["input_ids"].to(device), image_features=_features(batch["image_features"], device),
            attention_mask=batch["attention_mask"].to(device), decoder_input_ids=_on(batch, "decoder_input_ids", device),
            decoder_attention_mask=_on(batch, "decoder_attention_mask", device), labels=_on(batch, "labels", device),
            mrm_labels=[t.to(device) for t in batch["mrm_labels"]] if "mrm_labels" in batch else None,
            mrm_mask=_on(batch, "mrm_mask", device))   # the reference omits the mask here (validation.py:36); the rows need it
        return out[0]["loss"].item()
    return _run(epoch, model, val_loader, device, args, logger, log_interval, tb_writer, forward)


def validate_generation_score(epoch, model, gen_loader, reference, tokenizer, device, args, logger=None, log_interval=1,
                              tb_writer=None):
    """Reference src/validation.py:124-165: generate with `generate_text`, score the generations against `reference` with
    `src.evaluation.compute_metric_inference` (BLEU / METEOR / CIDEr through the pycocoevalcap Java tools), log and
    record `score/<name>` scalars.  The generation half is the hot path and runs here; the metric package is outside it
     and is used when an importable `src.evaluation` is on the path, otherwise this
    raises NotImplementedError BEFORE spending the generation time."""
    try:
        import src.evaluation as vcg   # provided by the user (the reference's src/evaluation.py + its Java dependencies)
    except ImportError as e:
        raise NotImplementedError(
            "validate_generation_score needs the reference's evaluation metrics (src/evaluation.py: BLEU / METEOR / CIDEr "
            "via pycocoevalcap and Java), which are outside the MI355X hot path and not shipped; put that module on the "
            "path as `src.evaluation`, or use --validate_loss") from e
    if not getattr(args, "cpu", False) and hasattr(model, "module"):
        model = model.module
    generated = generate_text(model=model, gen_loader=gen_loader, tokenizer=tokenizer, device=device, args=args,
                              logger=logger, log_interval=log_interval)
    scores = vcg.compute_metric_inference(gens_list=generated, refs_list=reference)
    if logger is not None:
        logger.info("Validation scores", pad=True)
        logger.info("Epoch: {}, BLEU2: {}, METEOR: {}, CIDEr: {}".format(epoch + 1, scores["BLEU2"], scores["METEOR"],
                                                                         scores["CIDEr"]))
        logger.line()
    if tb_writer is not None:
        for k, v in scores.items():
            tb_writer.add_scalar("score/{}".format(k), v, epoch + 1)
    return scores
