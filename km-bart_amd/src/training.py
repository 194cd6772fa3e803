"""Training loop with the reference's `fine_tune` contract (reference src/training.py:96-171):
forward -> loss.item() -> zero_grad -> (scaled) backward -> optimizer step, per-step log line
`Epoch [e/E], Step [i/N], Loss: x.xxxx, ETA: ...`, optional TensorBoard scalars and a callback."""
from datetime import datetime

import torch


def _on(batch, key, device):
    return batch[key].to(device) if key in batch and batch[key] is not None else None


def _features(feats, device):
    """list of per-sample tensors (reference collator) or a kmbart.data.PackedFeatures"""
    return feats.to(device) if hasattr(feats, "packed") else [f.to(device) for f in feats]


def fine_tune(epoch, model, train_loader, optimizer, device, args, logger=None, callback=None, log_interval=1,
              tb_writer=None, tb_interval=1, scaler=None):
    n_steps = len(train_loader)
    model.train()
    loss_sum = 0.0
    t0 = datetime.now()
    use_amp = bool(getattr(args, "amp", False))
    for i, batch in enumerate(train_loader):
        # the engine computes in bf16 with fp32 accumulation regardless of `amp`; autocast is kept so
        # that torch ops a caller adds around the model behave as in the reference
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=use_amp and torch.cuda.is_available()):
            outputs = model.forward(
                input_ids=batch["input_ids"].to(device),
                image_features=_features(batch["image_features"], device),
                attention_mask=batch["attention_mask"].to(device),
                decoder_input_ids=_on(batch, "decoder_input_ids", device),
                decoder_attention_mask=_on(batch, "decoder_attention_mask", device),
                labels=batch["labels"].to(device),
                answer_ids=_on(batch, "answer_ids", device),
                answer_attention_mask=_on(batch, "answer_attention_mask", device),
            )
            loss = outputs[0]
        loss_value = loss.item()
        loss_sum += loss_value
        optimizer.zero_grad()
        if use_amp and scaler is not None:
            scaler.scale(loss).backward()
            scaler.step(optimizer)
            scaler.update()
        else:
            loss.backward()
            optimizer.step()
        if logger is not None and i % log_interval == 0:
            eta = (n_steps - (i + 1)) / (i + 1) * (datetime.now() - t0)
            logger.info("Epoch [{}/{}], Step [{}/{}], Loss: {:.4f}, ETA: {}".format(
                epoch + 1, args.epochs, i + 1, n_steps, loss_value, str(eta)))
        if tb_writer is not None and i % tb_interval == 0:
            tb_writer.add_scalars("loss/step", {"loss": loss_value}, epoch * n_steps + i + 1)
        if callback is not None:
            callback(step=i, epoch=epoch, model=model, train_loader=train_loader, optimizer=optimizer, args=args,
                     logger=logger)
    if tb_writer is not None:
        tb_writer.add_scalars("loss/epoch", {"train": loss_sum / max(n_steps, 1)}, epoch + 1)
    return loss_sum / max(n_steps, 1)


def pretrain(epoch, model, train_loader, optimizer, device, args, logger=None, callback=None, log_interval=1,
             tb_writer=None, tb_interval=1, scaler=None):
    """Multi-task pre-training loop with the reference's contract (reference src/training.py:9-93): the model
    returns a dict of losses as outputs[0]; `loss` drives backward, the others are logged."""
    n_steps = len(train_loader)
    model.train()
    loss_sum = 0.0
    t0 = datetime.now()
    use_amp = bool(getattr(args, "amp", False))
    for i, batch in enumerate(train_loader):
        def opt(key):
            return batch[key].to(device) if key in batch and batch[key] is not None else None

        def opt_list(key):
            return [t.to(device) for t in batch[key]] if key in batch else None

        outputs = model.forward(
            input_ids=batch["input_ids"].to(device),
            image_features=_features(batch["image_features"], device),
            attention_mask=batch["attention_mask"].to(device),
            decoder_input_ids=opt("decoder_input_ids"),
            decoder_attention_mask=opt("decoder_attention_mask"),
            labels=opt("labels"),
            mrm_labels=opt_list("mrm_labels"),
            mrm_mask=opt("mrm_mask"),
            attribute_labels=opt_list("attribute_labels"),
            attribute_mask=opt("attribute_mask"),
            relation_labels=batch.get("relation_labels"),
        )
        losses = outputs[0]
        loss = losses["loss"]
        loss_value = loss.item()
        loss_sum += loss_value
        optimizer.zero_grad()
        if use_amp and scaler is not None:
            scaler.scale(loss).backward()
            scaler.step(optimizer)
            scaler.update()
        else:
            loss.backward()
            optimizer.step()
        if logger is not None and i % log_interval == 0:
            eta = (n_steps - (i + 1)) / (i + 1) * (datetime.now() - t0)
            logger.info("Epoch [{}/{}], Step [{}/{}], Loss: {:.4f}, ETA: {}".format(
                epoch + 1, args.epochs, i + 1, n_steps, loss_value, str(eta)))
        if tb_writer is not None and i % tb_interval == 0:
            step = epoch * n_steps + i + 1
            tb_writer.add_scalars("loss/step", {"total loss": loss_value}, step)
            for name, value in losses.items():
                if name != "loss":
                    tb_writer.add_scalars("loss/step", {name.replace("_", " "): value.item()}, step)
        if callback is not None:
            callback(step=i, epoch=epoch, model=model, train_loader=train_loader, optimizer=optimizer, args=args,
                     logger=logger)
    if tb_writer is not None:
        tb_writer.add_scalars("loss/epoch", {"train": loss_sum / max(n_steps, 1)}, epoch + 1)
    return loss_sum / max(n_steps, 1)
