"""Training loop with the reference's `fine_tune` contract (reference src/training.py:96-171):
forward -> zero_grad -> (scaled) backward -> optimizer step, per-step log line
`Epoch [e/E], Step [i/N], Loss: x.xxxx, ETA: ...`, optional TensorBoard scalars and a callback.

One deliberate difference in timing, none in content: the reference reads `loss.item()` right after forward
(training.py:134), which stalls the host until the GPU has drained and leaves the GPU idle while the host enqueues
the next ~600 launches.  Here step i's loss is read AFTER step i+1 has been enqueued (the last one after the loop), so
log lines / TensorBoard points / the returned epoch mean carry the same values, one step later.  A `callback` gets
called in step order as in the reference; it forces the pending loss to be read first."""
from datetime import datetime

import torch


def _on(batch, key, device):
    return batch[key].to(device) if key in batch and batch[key] is not None else None


def _allow_overlap(optimizer, ok):
    """These loops run `loss.backward()` and `optimizer.step()` back to back with nothing touching the gradients in
    between: kmbart.optim.AdamW may then step each gradient bucket beside the rest of backward (see its __init__)."""
    if hasattr(optimizer, "allow_overlap"):
        optimizer.allow_overlap(ok)


def _attach(model, optimizer, ok):
    """Data-parallel wrapper: chain each gradient piece's optimizer update behind its all-reduce (kmbart.parallel)."""
    if ok and hasattr(model, "attach_optimizer"):
        model.attach_optimizer(optimizer)


def _release(model, optimizer):
    """The fusions above hold only inside these loops: afterwards `loss.backward()` is a plain backward again (gradient
    accumulation, clip_grad_norm_, a diagnostic backward) and `optimizer.step()` a plain step."""
    _allow_overlap(optimizer, False)
    if hasattr(model, "detach_optimizer"):
        model.detach_optimizer()


def _features(feats, device):
    """list of per-sample tensors (reference collator) or a kmbart.data.PackedFeatures"""
    return feats.to(device) if hasattr(feats, "packed") else [f.to(device) for f in feats]


def fine_tune(epoch, model, train_loader, optimizer, device, args, logger=None, callback=None, log_interval=1,
              tb_writer=None, tb_interval=1, scaler=None):
    try:
        return _fine_tune(epoch, model, train_loader, optimizer, device, args, logger, callback, log_interval, tb_writer,
                          tb_interval, scaler)
    finally:
        _release(model, optimizer)


def _fine_tune(epoch, model, train_loader, optimizer, device, args, logger, callback, log_interval, tb_writer, tb_interval,
               scaler):
    n_steps = len(train_loader)
    model.train()
    t0 = datetime.now()
    use_amp = bool(getattr(args, "amp", False))
    _allow_overlap(optimizer, not (use_amp and scaler is not None))
    _attach(model, optimizer, not (use_amp and scaler is not None))
    pending = None
    state = {"sum": 0.0}

    def report(item):
        if item is None:
            return
        j, dev_loss, _ = item
        loss_value = dev_loss.item()
        state["sum"] += loss_value
        if logger is not None and j % log_interval == 0:
            eta = (n_steps - (j + 1)) / (j + 1) * (datetime.now() - t0)
            logger.info("Epoch [{}/{}], Step [{}/{}], Loss: {:.4f}, ETA: {}".format(
                epoch + 1, args.epochs, j + 1, n_steps, loss_value, str(eta)))
        if tb_writer is not None and j % tb_interval == 0:
            tb_writer.add_scalars("loss/step", {"loss": loss_value}, epoch * n_steps + j + 1)

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
        optimizer.zero_grad()
        if use_amp and scaler is not None:
            scaler.scale(loss).backward()
            scaler.step(optimizer)
            scaler.update()
        else:
            loss.backward()
            optimizer.step()
        report(pending)                       # step i-1, while step i runs on the GPU
        pending = (i, loss.detach(), None)
        if callback is not None:
            report(pending)
            pending = None
            callback(step=i, epoch=epoch, model=model, train_loader=train_loader, optimizer=optimizer, args=args,
                     logger=logger)
    report(pending)
    loss_sum = state["sum"]
    if tb_writer is not None:
        tb_writer.add_scalars("loss/epoch", {"train": loss_sum / max(n_steps, 1)}, epoch + 1)
    return loss_sum / max(n_steps, 1)


def pretrain(epoch, model, train_loader, optimizer, device, args, logger=None, callback=None, log_interval=1,
             tb_writer=None, tb_interval=1, scaler=None):
    """Multi-task pre-training loop with the reference's contract (reference src/training.py:9-93): the model
    returns a dict of losses as outputs[0]; `loss` drives backward, the others are logged."""
    try:
        return _pretrain(epoch, model, train_loader, optimizer, device, args, logger, callback, log_interval, tb_writer,
                         tb_interval, scaler)
    finally:
        _release(model, optimizer)


def _pretrain(epoch, model, train_loader, optimizer, device, args, logger, callback, log_interval, tb_writer, tb_interval,
              scaler):
    n_steps = len(train_loader)
    model.train()
    t0 = datetime.now()
    use_amp = bool(getattr(args, "amp", False))
    _allow_overlap(optimizer, not (use_amp and scaler is not None))
    _attach(model, optimizer, not (use_amp and scaler is not None))
    pending = None
    state = {"sum": 0.0}

    def report(item):
        if item is None:
            return
        j, dev_loss, parts = item
        loss_value = dev_loss.item()
        state["sum"] += loss_value
        if logger is not None and j % log_interval == 0:
            eta = (n_steps - (j + 1)) / (j + 1) * (datetime.now() - t0)
            logger.info("Epoch [{}/{}], Step [{}/{}], Loss: {:.4f}, ETA: {}".format(
                epoch + 1, args.epochs, j + 1, n_steps, loss_value, str(eta)))
        if tb_writer is not None and j % tb_interval == 0:
            step = epoch * n_steps + j + 1
            tb_writer.add_scalars("loss/step", {"total loss": loss_value}, step)
            for name, value in parts.items():
                tb_writer.add_scalars("loss/step", {name.replace("_", " "): value.item()}, step)

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
        optimizer.zero_grad()
        if use_amp and scaler is not None:
            scaler.scale(loss).backward()
            scaler.step(optimizer)
            scaler.update()
        else:
            loss.backward()
            optimizer.step()
        report(pending)
        pending = (i, loss.detach(), {k: v.detach() for k, v in losses.items() if k != "loss"})
        if callback is not None:
            report(pending)
            pending = None
            callback(step=i, epoch=epoch, model=model, train_loader=train_loader, optimizer=optimizer, args=args,
                     logger=logger)
    report(pending)
    loss_sum = state["sum"]
    if tb_writer is not None:
        tb_writer.add_scalars("loss/epoch", {"train": loss_sum / max(n_steps, 1)}, epoch + 1)
    return loss_sum / max(n_steps, 1)
