"""transformers-3.0.2-style AdamW (reference vcg_train.py:13,100: lr given, betas (0.9, 0.999), eps 1e-6,
weight_decay 0.0, correct_bias True) whose step() is ONE fused HIP kernel over the engine's flat arena
(28 B/param of HBM traffic + the bf16 mirror), instead of ~260 per-tensor launches."""
import torch


def reference_parameter_order(names):
    """`names` (the engine's parameter names) in the order the REFERENCE model's `parameters()` yields them, which is the
    order a torch-format optimizer state of the reference indexes its per-parameter entries by (reference
    vcg_train.py:100 `AdamW(model.parameters(), ...)`, saved by src/utils.py:20-39).  Module registration order:
    src/model/model.py:29-33 (shared, encoder, decoder), src/model/modules.py:73-88 (embed_tokens = shared, embed_images,
    embed_positions, layers, layernorm_embedding), transformers 3.0.2 EncoderLayer / DecoderLayer (self_attn,
    self_attn_layer_norm, [encoder_attn, encoder_attn_layer_norm,] fc1, fc2, final_layer_norm) with SelfAttention
    registering k_proj, v_proj, q_proj, out_proj in that order, BartDecoder (embed_positions, layers,
    layernorm_embedding), then src/model/model.py:133-158 (mrm_head, attribute_head, relation_head: dense, out_proj)."""
    have = set(names)
    out = []

    def add(n):
        if n in have:
            out.append(n)

    def lin(prefix):
        add(prefix + ".weight")
        add(prefix + ".bias")

    def attn(prefix):
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            lin(prefix + "." + proj)

    def n_layers(side):
        k = 0
        while "model.%s.layers.%d.fc1.weight" % (side, k) in have:
            k += 1
        return k

    add("model.shared.weight")
    lin("model.encoder.embed_images.linear")
    add("model.encoder.embed_positions.weight")
    for l in range(n_layers("encoder")):
        p = "model.encoder.layers.%d" % l
        attn(p + ".self_attn")
        lin(p + ".self_attn_layer_norm")
        lin(p + ".fc1")
        lin(p + ".fc2")
        lin(p + ".final_layer_norm")
    lin("model.encoder.layernorm_embedding")
    add("model.decoder.embed_positions.weight")
    for l in range(n_layers("decoder")):
        p = "model.decoder.layers.%d" % l
        attn(p + ".self_attn")
        lin(p + ".self_attn_layer_norm")
        attn(p + ".encoder_attn")
        lin(p + ".encoder_attn_layer_norm")
        lin(p + ".fc1")
        lin(p + ".fc2")
        lin(p + ".final_layer_norm")
    lin("model.decoder.layernorm_embedding")
    for head in ("mrm_head", "attribute_head", "relation_head"):
        lin(head + ".dense")
        lin(head + ".out_proj")
    if len(out) != len(have):
        raise ValueError("parameters without a place in the reference order: %s" % sorted(have - set(out))[:4])
    return out


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameters: {} - should be in [0.0, 1.0[".format(betas))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      correct_bias=correct_bias))
        self.grad_scale = 1.0  # data-parallel wrapper may fold 1/world here
        import os
        # Overlapped stepping (Engine.adamw_step_overlapped) issues each bucket's update on a second stream behind that
        # bucket's completion event of the backward pass still running on the GPU.  It is only correct when NOTHING
        # touches the gradients between `loss.backward()` and `step()` (clip_grad_norm_, GradScaler.unscale_, manual
        # scaling all enqueue on the main stream and would race with the side-stream update), so it is opt-in: the
        # loops that guarantee the adjacency (src.training.fine_tune / pretrain without a scaler, bench.py) set
        # `optimizer.overlap = True`; KMB_ADAMW_OVERLAP=1 forces it on, =0 forbids it.
        env = os.environ.get("KMB_ADAMW_OVERLAP")
        self.overlap = env == "1"
        self._overlap_locked = env is not None

    def _engine_ranges(self, group):
        """[(engine, offset, count)] with adjacent parameters coalesced."""
        items = []
        for p in group["params"]:
            eng = getattr(p, "_kmb_engine", None)
            if eng is None:
                raise RuntimeError("kmbart.optim.AdamW only steps parameters that live in a kmbart engine arena "
                                   "(move the model to the GPU before building the optimizer)")
            off, cnt = p._kmb_range
            items.append((eng, off, cnt))
        items.sort(key=lambda x: (id(x[0]), x[1]))
        out = []
        for eng, off, cnt in items:
            if out and out[-1][0] is eng and out[-1][1] + out[-1][2] == off:
                out[-1] = (eng, out[-1][1], out[-1][2] + cnt)
            else:
                out.append((eng, off, cnt))
        return out

    def allow_overlap(self, on=True):
        """Called by a training loop that runs `loss.backward()` and `step()` back to back (see __init__)."""
        if not self._overlap_locked:
            self.overlap = bool(on)

    # ---- data-parallel fused tail (kmbart.parallel.DistributedDataParallel.attach_optimizer) -------------------------
    def begin_fused_step(self, engine):
        """Called by the data-parallel wrapper before it launches a step's all-reduces: True if this optimizer will
        update each gradient piece right behind its all-reduce (one parameter group covering the whole engine arena,
        overlap allowed, no gradient scaling)."""
        if not (self.overlap and self.grad_scale == 1.0 and len(self.param_groups) == 1):
            return False
        params = self.param_groups[0]["params"]
        # every parameter of this engine and nothing else: the pieces then tile the whole arena (alignment gaps between
        # parameters hold zeros in all four arenas: their update is exactly zero)
        if len(params) != len(engine.index) or any(getattr(q, "_kmb_engine", None) is not engine for q in params):
            return False
        engine.step_count += 1
        self._fused_done = set()
        return True

    def fused_hyperparams(self, engine):
        """The KmbAdamW record of this step for the native exchange (kmb_allreduce_grads chains the update of every
        piece behind its collective itself); begin_fused_step has already advanced the step count."""
        from ._lib import KmbAdamW
        g = self.param_groups[0]
        return KmbAdamW(lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"],
                        step=engine.step_count, correct_bias=1 if g["correct_bias"] else 0, grad_scale=1.0)

    def fused_piece_step(self, engine, off, cnt):
        g = self.param_groups[0]
        engine.adamw_step(g["lr"], g["betas"], g["eps"], g["weight_decay"], g["correct_bias"], 1.0, offset=off,
                          count=cnt, bump=False)
        self._fused_done.add(id(engine))

    def end_fused_step(self, engine):
        self._fused_pending = True

    def _engines(self):
        engines = {}
        for g in self.param_groups:
            for e, _, _ in self._engine_ranges(g):
                engines[id(e)] = e
        return list(engines.values())

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if getattr(self, "_fused_pending", False):
            # the data-parallel wrapper already enqueued every piece's update behind its all-reduce and made the compute
            # stream wait for the communication stream (BucketedAllReducer.finish)
            self._fused_pending = False
            return loss
        all_ranges = [self._engine_ranges(group) for group in self.param_groups]
        for e in self._engines():
            # a reduce-scatter exchange with the fused optimizer left exp_avg / exp_avg_sq current on their owning rank only
            # (kmbart.parallel, algo "rsag"); this whole-arena step needs them everywhere.  step() runs on every rank, so the
            # collective is matched.
            if getattr(e, "moments_sharded", False):
                e.comm_gather_moments()
        # ONE bias-correction step per optimizer.step(), whatever the number of parameter groups
        for e in self._engines():
            e.step_count += 1
        overlap = self.overlap and self.grad_scale == 1.0
        for group, ranges in zip(self.param_groups, all_ranges):
            for eng, off, cnt in ranges:
                args = (group["lr"], group["betas"], group["eps"], group["weight_decay"], group["correct_bias"],
                        self.grad_scale)
                # per gradient bucket, beside the backward pass still running on the GPU (Engine.adamw_step_overlapped)
                if not (overlap and eng.adamw_step_overlapped(*args, offset=off, count=cnt)):
                    eng.adamw_step(*args, offset=off, count=cnt, bump=False)
        return loss

    def zero_grad(self, set_to_none=False):
        pass  # gradients are views into the arena and are overwritten by the next backward

    def state_dict(self):
        eng = self._engines()
        for e in eng:
            if getattr(e, "moments_sharded", False):
                raise RuntimeError("exp_avg / exp_avg_sq are sharded over the data-parallel ranks (reduce-scatter exchange with "
                                   "the fused optimizer): call DistributedDataParallel.gather_optimizer_state() on EVERY rank "
                                   "(it is a collective) before optimizer.state_dict()")
        return {
            "kmbart_adamw": True,
            "step": [e.step_count for e in eng],
            "exp_avg": [e.exp_avg.detach().cpu() for e in eng],
            "exp_avg_sq": [e.exp_avg_sq.detach().cpu() for e in eng],
            # the arena layout the two moment arenas were dumped in: name -> (offset, element count).  The arena order is an
            # implementation detail of the engine (round 3 moved every decoder layer's cross-attention k | v block), so a
            # raw dump is only meaningful together with it: load_state_dict maps by NAME.
            "layout": [{n: (int(off), int(rows) * int(cols)) for n, (off, rows, cols) in e.index.items()} for e in eng],
            "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
        }

    def load_state_dict(self, state, assume_layout=False):
        """Accepts this class's own format and the torch / transformers.AdamW format the reference saves in
        `training_data.pt` (reference src/utils.py:20-39: {'state': {index: {'step', 'exp_avg', 'exp_avg_sq'}},
        'param_groups': [{..., 'params': [indices]}]}): per-parameter moments are copied into the arena slices.
        assume_layout=True: a kmbart state saved before the arena layout was recorded is copied raw when its arena size
        equals this engine's (correct only if the arena order has not changed since it was saved: the caller's claim)."""
        if state.get("kmbart_adamw"):
            layouts = state.get("layout")
            if layouts is None:
                engines = self._engines()
                if assume_layout and all(state["exp_avg"][i].numel() == e.exp_avg.numel() for i, e in enumerate(engines)):
                    for i, e in enumerate(engines):
                        e.step_count = int(state["step"][i])
                        e.exp_avg.copy_(state["exp_avg"][i])
                        e.exp_avg_sq.copy_(state["exp_avg_sq"][i])
                    for g, s in zip(self.param_groups, state["param_groups"]):
                        g.update(s)
                    return
                raise ValueError("this kmbart AdamW state has no arena layout (saved before the layout was recorded): the "
                                 "moment arenas cannot be assigned to parameters safely; save the optimizer again with this "
                                 "build, load a torch-format {state, param_groups} dict, or -- if the arena order is known to "
                                 "be unchanged and the sizes match -- pass assume_layout=True")
            for i, e in enumerate(self._engines()):
                saved = layouts[i]
                here = {n: (int(off), int(rows) * int(cols)) for n, (off, rows, cols) in e.index.items()}
                if set(saved) != set(here):
                    raise ValueError("optimizer state was saved for a different parameter set: %s"
                                     % sorted(set(saved) ^ set(here))[:4])
                e.step_count = int(state["step"][i])
                if all(tuple(saved[n]) == here[n] for n in here) and state["exp_avg"][i].numel() == e.exp_avg.numel():
                    e.exp_avg.copy_(state["exp_avg"][i])
                    e.exp_avg_sq.copy_(state["exp_avg_sq"][i])
                    continue
                # a different arena order: copy every parameter's moments by name (alignment gaps stay zero)
                m, v = state["exp_avg"][i].reshape(-1), state["exp_avg_sq"][i].reshape(-1)
                e.exp_avg.zero_()
                e.exp_avg_sq.zero_()
                for n, (off, cnt) in here.items():
                    soff, scnt = (int(x) for x in saved[n])
                    if scnt != cnt:
                        raise ValueError("optimizer state of %s has %d elements, expected %d" % (n, scnt, cnt))
                    e.exp_avg[off: off + cnt].copy_(m[soff: soff + cnt])
                    e.exp_avg_sq[off: off + cnt].copy_(v[soff: soff + cnt])
            for g, s in zip(self.param_groups, state["param_groups"]):
                g.update(s)
            return
        if "state" not in state or "param_groups" not in state:
            raise ValueError("optimizer state is neither a kmbart AdamW state nor a torch-format {state, param_groups} dict")
        if len(state["param_groups"]) != len(self.param_groups):
            raise ValueError("loaded optimizer state has %d parameter groups, this optimizer has %d"
                             % (len(state["param_groups"]), len(self.param_groups)))
        steps = []
        for g, sg in zip(self.param_groups, state["param_groups"]):
            if len(sg["params"]) != len(g["params"]):
                raise ValueError("loaded optimizer state has a parameter group of a different size")
            # The saved indices follow the REFERENCE model's parameters() order (shared first, k/v/q with weight and bias
            # interleaved, layernorm_embedding after the layers), not this engine's arena order: map by NAME.
            by_name = {getattr(p, "_kmb_name", None): p for p in g["params"]}
            if None in by_name:
                raise RuntimeError("a parameter of this optimizer does not live in a kmbart engine arena")
            order = reference_parameter_order(list(by_name))
            for name, idx in zip(order, sg["params"]):
                p = by_name[name]
                st = state["state"].get(idx)
                if st is None:
                    st = state["state"].get(str(idx))
                if not st:
                    continue
                eng = p._kmb_engine
                off, cnt = p._kmb_range
                if st["exp_avg"].numel() != cnt:
                    raise ValueError("optimizer state %s (%s in the reference's parameter order) has %d elements, expected %d"
                                     % (idx, name, st["exp_avg"].numel(), cnt))
                eng.exp_avg[off: off + cnt].copy_(st["exp_avg"].reshape(-1))
                eng.exp_avg_sq[off: off + cnt].copy_(st["exp_avg_sq"].reshape(-1))
                steps.append(int(st["step"]))
            g.update({k: v for k, v in sg.items() if k != "params"})
        if steps:
            if min(steps) != max(steps):
                raise ValueError("per-parameter step counts differ (%d..%d): one fused step count is kept per engine"
                                 % (min(steps), max(steps)))
            for e in self._engines():
                e.step_count = steps[0]
