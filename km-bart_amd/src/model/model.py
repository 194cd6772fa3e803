"""MultiModalBartForConditionalGeneration on the MI355X engine.

Same call surface as the reference class (reference src/model/model.py:317-405, generation
src/model/mixins.py:33-434, checkpoint glue src/model/mixins.py:458-883), but the arithmetic runs in
libkmbart_hip.so: this module only marshals tensors, keeps the reference's return conventions and
does the beam bookkeeping on the host.  There is no CPU forward: calling the model before it has
been moved to a HIP device raises.
"""
import math
import os

import torch
from torch import nn

from kmbart.engine import Engine
from src.model.config import MultiModalBartConfig

WEIGHTS_NAME = "pytorch_model.bin"
_TIED = ("model.encoder.embed_tokens.weight", "model.decoder.embed_tokens.weight")


def _param_names(cfg, with_heads=False):
    names = ["model.encoder.embed_images.linear.weight", "model.encoder.embed_images.linear.bias",
             "model.encoder.embed_positions.weight", "model.encoder.layernorm_embedding.weight",
             "model.encoder.layernorm_embedding.bias"]

    def attn(p, blk):
        out = []
        for a in ("q_proj", "k_proj", "v_proj"):
            out.append(p + f"{blk}.{a}.weight")
        for a in ("q_proj", "k_proj", "v_proj"):
            out.append(p + f"{blk}.{a}.bias")
        out += [p + f"{blk}.out_proj.weight", p + f"{blk}.out_proj.bias",
                p + f"{blk}_layer_norm.weight", p + f"{blk}_layer_norm.bias"]
        return out

    def ffn(p):
        return [p + "fc1.weight", p + "fc1.bias", p + "fc2.weight", p + "fc2.bias",
                p + "final_layer_norm.weight", p + "final_layer_norm.bias"]

    for i in range(cfg.encoder_layers):
        p = f"model.encoder.layers.{i}."
        names += attn(p, "self_attn") + ffn(p)
    names += ["model.decoder.embed_positions.weight", "model.decoder.layernorm_embedding.weight",
              "model.decoder.layernorm_embedding.bias"]
    for i in range(cfg.decoder_layers):
        p = f"model.decoder.layers.{i}."
        names += attn(p, "self_attn") + attn(p, "encoder_attn") + ffn(p)
    for head, attr in (("mrm_head", "num_labels"), ("attribute_head", "num_attributes"), ("relation_head", "num_relations")):
        if with_heads and getattr(cfg, attr, 0) > 0:
            names += [f"{head}.dense.weight", f"{head}.dense.bias", f"{head}.out_proj.weight", f"{head}.out_proj.bias"]
    names.append("model.shared.weight")
    return names


def _param_shape(cfg, name):
    d = cfg.d_model
    for head, attr, mult in (("mrm_head", "num_labels", 1), ("attribute_head", "num_attributes", 1),
                             ("relation_head", "num_relations", 2)):
        if name.startswith(head + "."):
            C = getattr(cfg, attr)
            return {"dense.weight": (d, mult * d), "dense.bias": (d,), "out_proj.weight": (C, d),
                    "out_proj.bias": (C,)}[name[len(head) + 1:]]
    if name == "model.shared.weight":
        return (cfg.vocab_size, d)
    if name.endswith("embed_images.linear.weight"):
        return (d, cfg.image_feature_size)
    if name.endswith("embed_positions.weight"):
        return (cfg.max_position_embeddings + cfg.extra_pos_embeddings, d)
    ffn = cfg.encoder_ffn_dim if ".encoder." in name else cfg.decoder_ffn_dim
    if name.endswith("fc1.weight"):
        return (ffn, d)
    if name.endswith("fc1.bias"):
        return (ffn,)
    if name.endswith("fc2.weight"):
        return (d, ffn)
    if name.endswith("proj.weight"):
        return (d, d)
    return (d,)


class _LossFn(torch.autograd.Function):
    """Connects the engine's loss scalar to autograd so `loss.backward()` / `scaler.scale(loss).backward()`
    (reference src/training.py:137-142) run kmb_backward."""

    @staticmethod
    def forward(ctx, anchor, model, loss, enc_states=None):
        ctx.model = model
        ctx.enc_meta = None if enc_states is None else (enc_states.dtype, enc_states.device, tuple(enc_states.shape))
        return loss.reshape(()).clone()

    @staticmethod
    def backward(ctx, grad_out):
        # the upstream gradient (ones, or the GradScaler's scale) stays on the device: no host synchronisation
        ctx.model._backward(grad_out if grad_out.is_cuda else float(grad_out))
        g_enc = None
        if ctx.enc_meta is not None and ctx.needs_input_grad[3]:
            # forward(encoder_outputs=...) with a tensor that requires grad (src/model/model.py:76-83): the engine's backward
            # stopped at the given states; their gradient is the cross-attention key / value data gradient, summed over layers
            dtype, device, shape = ctx.enc_meta
            g_enc = ctx.model._engine.encoder_states_grad().to(device=device, dtype=dtype).reshape(shape)
        return None, None, None, g_enc


class LazyLogits:
    """outputs[1] of a training forward: the fp32 logits are produced on first use -- ONE head GEMM on the decoder
    states that forward left in the engine's workspace (the actual training logits, dropout included, as the reference
    returns them) -- instead of 1.6 GB written every step that the training loop never reads.  They can only be
    produced while those states and the weights that made them are still there: after `optimizer.step()`, another
    forward or a `generate()` the access raises; pass `return_logits=True` to forward to get them eagerly."""

    def __init__(self, engine):
        self._eng, self._serial, self._t = engine, engine.fwd_serial, None

    def tensor(self):
        if self._t is None:
            if self._eng.fwd_serial != self._serial:
                raise RuntimeError("the logits of this forward are gone: the model has run another forward / generate or "
                                   "an optimizer step since (call forward(..., return_logits=True) to keep them)")
            self._t = self._eng.last_logits()
        return self._t

    def __getattr__(self, name):
        return getattr(self.tensor(), name)

    def __getitem__(self, idx):
        return self.tensor()[idx]


class LazyEncoderStates(LazyLogits):
    """The encoder_last_hidden_state entry of a TRAINING forward's tuple (reference src/model/model.py:100-103 returns it; no
    training loop reads it): copied out of the engine's workspace on first use (kmb_hidden_state) instead of a [B, S, d]
    allocation + copy in every step (~100 MB at 1024 samples).  Same lifetime as LazyLogits."""

    def tensor(self):
        if self._t is None:
            if self._eng.fwd_serial != self._serial:
                raise RuntimeError("the encoder states of this forward are gone: the model has run another forward / generate "
                                   "or an optimizer step since (read outputs[2] before optimizer.step())")
            self._t = self._eng.encoder_last_state()
        return self._t


class BeamHypotheses:
    """transformers 3.0.2 BeamHypotheses (n-best list, score = sum_logprobs / len ** length_penalty)."""

    def __init__(self, num_beams, max_length, length_penalty, early_stopping):
        self.max_length = max_length - 1
        self.length_penalty = length_penalty
        self.early_stopping = early_stopping
        self.num_beams = num_beams
        self.beams = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / len(hyp) ** self.length_penalty
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                sorted_scores = sorted([(s, idx) for idx, (s, _) in enumerate(self.beams)])
                del self.beams[sorted_scores[0][1]]
                self.worst_score = sorted_scores[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self) < self.num_beams:
            return False
        if self.early_stopping:
            return True
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


class DecoderCache:
    """`decoder_cached_states` of the cached forward (reference src/model/model.py:384-397 returns the per-layer dicts of
    transformers 3.0.2's BartDecoder; src/model/mixins.py:386-398 hands them back through `past`).  Here the keys / values
    live in the engine's generation workspace (kmb_gen_begin / kmb_gen_step); this handle names them: the engine, the
    number of decoder positions fed so far, and the engine serial they were built under (any other forward, generate or
    optimizer step on the same model re-uses that workspace and invalidates the handle)."""

    def __init__(self, engine, rows, max_length):
        self.engine, self.rows, self.max_length, self.length = engine, rows, max_length, 0
        self.serial = engine.fwd_serial

    def check(self):
        if self.serial != self.engine.fwd_serial:
            raise RuntimeError("decoder_cached_states is stale: another forward / generate / optimizer step ran on this "
                               "model since the cache was created (the KV cache lives in the engine's workspace)")

    def __len__(self):
        return self.length


class _EncoderHandle:
    """`model.get_encoder()` (reference src/model/mixins.py:436-437).  Calling it runs the multimodal encoder once and
    returns the reference's filtered encoder tuple `(encoder_states,)`; the states are also left in the engine's generation
    workspace together with every decoder layer's cross-attention keys / values, so a cached forward that receives this
    tuple as `encoder_outputs` (prepare_inputs_for_generation, mixins.py:386-398) continues from it without input_ids."""

    def __init__(self, model):
        self._model = model

    def __call__(self, input_ids, image_features, attention_mask=None, max_cache_length=None, **unused):
        m = self._model
        eng = m._need_engine()
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        cap = int(max_cache_length or max(int(m.config.max_length), 64))
        eng.gen_begin(input_ids, image_features, attention_mask, 1, cap)
        enc = eng.gen_encoder_states()
        # The cache handle is kept BY ENGINE, not on the returned tensor: the reference's generate loop expands the encoder
        # states with index_select before the first cached step (src/model/mixins.py:286-324), which makes a new tensor.
        # The inputs stay with it so that rows expanded num_beams-fold can re-create the cross-attention tables.
        eng._gen_pending = {"cache": DecoderCache(eng, input_ids.shape[0], cap), "inputs": (input_ids, image_features, attention_mask, cap)}
        return (enc,)

    forward = __call__


class MultiModalBartForConditionalGeneration(nn.Module):
    base_model_prefix = "model"
    config_class = MultiModalBartConfig
    _with_heads = False

    def __init__(self, config: MultiModalBartConfig):
        super().__init__()
        config.check_supported()
        self.config = config
        self._engine = None
        self._names = _param_names(config, self._with_heads)
        g = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
        self._p = {}
        # transformers `init_weights` (model.py:37): N(0, init_std), zero biases / pad rows, LayerNorm (1, 0)
        for n in self._names:
            shp = _param_shape(config, n)
            if "layer_norm" in n or "layernorm" in n:
                t = torch.ones(shp) if n.endswith("weight") else torch.zeros(shp)
            elif n.endswith("bias"):
                t = torch.zeros(shp)
            else:
                t = torch.randn(shp, generator=g) * config.init_std
                if n == "model.shared.weight" or n.endswith("embed_positions.weight"):
                    t[config.pad_token_id].zero_()
            self._p[n] = nn.Parameter(t)
        self._flb = torch.zeros((1, config.vocab_size))  # buffer final_logits_bias (model.py:323)
        self._anchor = None
        self._post_backward = None  # set by the data-parallel wrapper
        self._dec_hidden_for_logits = None

    # ------------------------------------------------------------------ nn.Module surface
    def named_parameters(self, prefix="", recurse=True, remove_duplicate=True):
        for n in self._names:
            yield (prefix + ("." if prefix else "") + n, self._p[n])

    def parameters(self, recurse=True):
        for _, p in self.named_parameters():
            yield p

    @property
    def final_logits_bias(self):
        return self._engine.final_logits_bias.view(1, -1) if self._engine is not None else self._flb

    def state_dict(self, *args, **kwargs):
        sd = {"final_logits_bias": self.final_logits_bias.detach().clone().cpu()}
        for n in self._names:
            sd[n] = self._p[n].detach().clone().cpu()
        for t in _TIED:
            sd[t] = sd["model.shared.weight"]
        return sd

    def load_state_dict(self, state_dict, strict=True):
        missing, unexpected = [], []
        with torch.no_grad():
            for n in self._names:
                if n in state_dict:
                    self._p[n].copy_(state_dict[n].to(self._p[n].dtype))
                else:
                    missing.append(n)
            if "final_logits_bias" in state_dict:
                self.final_logits_bias.copy_(state_dict["final_logits_bias"].view(1, -1))
            for k in state_dict:
                if k not in self._p and k not in _TIED and k != "final_logits_bias":
                    unexpected.append(k)
        if strict and (missing or unexpected):
            raise RuntimeError("load_state_dict: missing %s unexpected %s" % (missing, unexpected))
        if self._engine is not None:
            self._engine.sync_params()
        return missing, unexpected

    def to(self, *args, **kwargs):
        device = None
        for a in args:
            if isinstance(a, (str, torch.device)):
                device = torch.device(a)
            elif isinstance(a, int):
                device = torch.device("cuda", a)
        device = torch.device(kwargs["device"]) if "device" in kwargs else device
        if device is None:
            return self
        if device.type != "cuda":
            if self._engine is not None:
                raise RuntimeError("moving a device model back to CPU is not supported; use state_dict()")
            return self
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self._engine is not None:
            if self._engine.device != device:
                raise RuntimeError("model already lives on %s" % self._engine.device)
            return self
        eng = Engine(self.config, device, with_heads=self._with_heads)
        with torch.no_grad():
            for n in self._names:
                eng.view(eng.params, n).copy_(self._p[n].detach().to(device))
            eng.final_logits_bias.copy_(self._flb.view(-1).to(device))
        for n in self._names:
            p = nn.Parameter(eng.view(eng.params, n))
            p.grad = eng.view(eng.grads, n)
            p._kmb_engine = eng
            p._kmb_name = n
            off, rows, cols = eng.index[n]
            p._kmb_range = (off, rows * cols)
            self._p[n] = p
        self._engine = eng
        self._anchor = torch.zeros((), device=device, requires_grad=True)
        eng.sync_params()
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def zero_grad(self, set_to_none=False):
        pass  # every gradient is overwritten by the next backward (reference zeroes after forward, training.py:136)

    @property
    def device(self):
        return self._engine.device if self._engine is not None else torch.device("cpu")

    def _need_engine(self):
        if self._engine is None:
            raise RuntimeError("MultiModalBartForConditionalGeneration runs on an MI355X only: call .to('cuda:N') "
                               "first (there is no CPU forward path)")
        return self._engine

    # ------------------------------------------------------------------ forward / backward
    def forward(self, input_ids, image_features, attention_mask=None, encoder_outputs=None, decoder_input_ids=None,
                decoder_attention_mask=None, decoder_cached_states=None, labels=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_logits=None, **unused):
        """Reference src/model/model.py:325-405.  Returns (loss, logits, encoder_last_hidden) with labels,
        (logits, encoder_last_hidden) without; with `use_cache=True` (and no labels) the KV-cached step of
        src/model/model.py:384-397: (logits of the LAST decoder position [rows, 1, V], decoder_cached_states,
        encoder_last_hidden), see _forward_cached.  Deliberate difference: `use_cache=None` means False here -- the
        reference falls back to config.use_cache (True), i.e. an eval forward without labels returns only the last
        position there; no caller of the reference relies on that (fine_tune / validation pass labels, generate passes
        use_cache), and the teacher-forced logits of every position are what the parity tests compare."""
        eng = self._need_engine()
        use_cache = self._resolve_use_cache(use_cache, labels is not None, decoder_input_ids, decoder_cached_states,
                                            output_attentions, output_hidden_states)
        if use_cache or decoder_cached_states is not None:
            if output_attentions or output_hidden_states:
                raise NotImplementedError("attention / hidden-state outputs are not materialised by the fused kernels")
            return self._forward_cached(input_ids, image_features, attention_mask, encoder_outputs, decoder_input_ids,
                                        decoder_cached_states, unused.get("max_cache_length"))
        output_attentions = self.config.output_attentions if output_attentions is None else output_attentions
        output_hidden_states = self.config.output_hidden_states if output_hidden_states is None else output_hidden_states
        enc_states = None
        if encoder_outputs is not None:   # src/model/model.py:76-83: a tuple whose first element is the encoder output
            enc_states = encoder_outputs[0] if isinstance(encoder_outputs, (tuple, list)) else encoder_outputs
        if decoder_input_ids is None:
            # transformers shift_tokens_right(input_ids, pad) (HF3.0.2 _prepare_bart_decoder_inputs)
            pad = self.config.pad_token_id
            prev = input_ids.clone()
            idx_eos = (input_ids.ne(pad).sum(dim=1) - 1).unsqueeze(-1)
            prev[:, 0] = input_ids.gather(1, idx_eos).squeeze()
            prev[:, 1:] = input_ids[:, :-1]
            decoder_input_ids = prev
        if decoder_attention_mask is None and bool((decoder_input_ids == self.config.pad_token_id).any()):
            decoder_attention_mask = decoder_input_ids.ne(self.config.pad_token_id).long()
        need_grad = labels is not None and torch.is_grad_enabled()
        want_logits = (labels is None) if return_logits is None else bool(return_logits)
        # a training forward's encoder states leave the workspace only when somebody reads them (LazyEncoderStates)
        lazy_enc = need_grad and enc_states is None and not eng.fp32_mode
        loss, logits, enc = eng.forward(input_ids, image_features, attention_mask, decoder_input_ids,
                                        decoder_attention_mask, labels, train=self.training, need_grad=need_grad,
                                        want_logits=want_logits, encoder_states=enc_states, want_encoder=not lazy_enc)
        if lazy_enc:
            enc = LazyEncoderStates(eng)
        # output_hidden_states / output_attentions (src/model/modules.py:143-165, transformers 3.0.2 BartDecoder; the tuple
        # is decoder_outputs + encoder_outputs with empty entries filtered, src/model/model.py:100-103): decoder = the layers'
        # INPUTS and their self-attention weights, encoder = the layers' inputs + the final output and the attention weights.
        # Read back from the workspace of this forward (kmb_hidden_state / kmb_attention_probs).
        dec_extra, enc_extra = (), ()
        if output_hidden_states:
            dec_extra += (eng.hidden_states(1)[:-1],)
            if enc_states is None:
                enc_extra += (list(eng.hidden_states(0)),)
        if output_attentions:
            dec_extra += (eng.attention_probs(1),)
            if enc_states is None:
                enc_extra += (list(eng.attention_probs(0)),)
        if enc_states is not None and isinstance(encoder_outputs, (tuple, list)):
            # a precomputed encoder tuple is passed through, empty entries filtered (_filter_out_falsey_values, model.py:102)
            enc_extra = tuple(x for x in encoder_outputs[1:] if isinstance(x, torch.Tensor) or x)
        if labels is None:
            return (logits,) + dec_extra + (enc,) + enc_extra
        if need_grad:
            # given encoder states take part in autograd: backward stops at them and hands their gradient on (the encoder's
            # own parameters get zero gradients from this loss, as in the reference where they are not in the graph)
            loss = _LossFn.apply(self._anchor, self, loss, enc_states if torch.is_tensor(enc_states) else None)
        else:
            loss = loss.view(())
        if logits is None:
            logits = LazyLogits(eng)
        return (loss, logits) + dec_extra + (enc,) + enc_extra

    def _resolve_use_cache(self, use_cache, has_labels, decoder_input_ids, decoder_cached_states, output_attentions,
                           output_hidden_states):
        """Reference src/model/model.py:52-59 and :381-382: labels (any loss) or missing decoder_input_ids switch the cache
        off, an explicit value is kept, None means config.use_cache.  Two cases keep the DEFAULT off here (an explicit
        use_cache=True still raises there): training mode (the cached decode blocks are an inference path without dropout)
        and requested attention / hidden-state outputs (the decode blocks do not materialise them)."""
        if has_labels or decoder_input_ids is None:
            return False
        if use_cache is None:
            if self.training or output_attentions or output_hidden_states:
                return False
            return bool(getattr(self.config, "use_cache", True))
        return bool(use_cache)

    def _forward_cached(self, input_ids, image_features, attention_mask, encoder_outputs, decoder_input_ids,
                        decoder_cached_states, max_cache_length=None, want_hidden=False):
        """One KV-cached decoder step over kmb_gen_step (reference src/model/model.py:384-397, mixins.py:386-398;
        transformers 3.0.2 BartDecoder with use_cache: only the last column of decoder_input_ids is embedded, at position
        len - 1).  First call (decoder_cached_states None): the encoder runs from input_ids / image_features -- or is
        taken from `encoder_outputs` when that is what get_encoder()(...) returned -- and every column of
        decoder_input_ids is fed in order; later calls feed the last column.  Rows are independent (a beam search
        passes already-expanded rows and permutes them with _reorder_cache)."""
        eng = self._need_engine()
        if self.training:
            raise RuntimeError("the cached decoder step is an inference path: call model.eval() first")
        if decoder_input_ids is None:
            raise ValueError("use_cache=True needs decoder_input_ids (src/model/model.py:52-53 turns the cache off without them)")
        cache = decoder_cached_states
        enc = encoder_outputs[0] if isinstance(encoder_outputs, (tuple, list)) else encoder_outputs
        pend = getattr(eng, "_gen_pending", None)
        if cache is None and enc is not None:
            # the reference's flow (mixins.py:281-324, :386-398): get_encoder()(...) ran, its states -- possibly expanded
            # num_beams-fold with index_select(0, arange(B).repeat_interleave(k)) -- come back as encoder_outputs.  When input_ids
            # are passed TOO the reference uses encoder_outputs as given and never runs its encoder (src/model/model.py:76-83): so
            # does this path.  Encoder states this model did not just compute cannot be honoured (the cross-attention keys / values
            # are the library's): that raises instead of silently re-running the encoder on input_ids.
            if pend is None or pend["cache"].serial != eng.fwd_serial or pend["cache"].length != 0:
                raise ValueError("the first cached step needs input_ids / image_features, or encoder_outputs made by "
                                 "model.get_encoder()(...) of THIS model with no other forward in between"
                                 + ("" if input_ids is None else " (input_ids were passed together with encoder_outputs: the "
                                    "reference uses the given encoder_outputs, which are not this model's pending ones)"))
            cache = pend["cache"]
            rows = decoder_input_ids.shape[0]
            ids0, feats0, mask0, cap0 = pend["inputs"]
            if rows % ids0.shape[0] != 0 or enc.shape[0] != rows:
                raise ValueError("decoder_input_ids has %d rows; the encoder ran on %d items" % (rows, ids0.shape[0]))
            k = rows // ids0.shape[0]
            # the given states must BE the pending ones, row i = item i // k (the reference's expansion order): states expanded in
            # another order or edited by the caller would otherwise be silently replaced by the library's own
            mine = eng.gen_encoder_states()
            if k > 1:
                mine = mine.repeat_interleave(k, dim=0)
            given = enc.to(device=mine.device)
            if given.shape != mine.shape or not torch.equal(given.to(mine.dtype), mine):
                raise ValueError("encoder_outputs are not the states model.get_encoder()(...) returned (repeated item-major %d-fold): "
                                 "the cached decoder step cannot take edited or re-ordered encoder states" % k)
            if rows != cache.rows:
                # rows expanded k-fold: row i belongs to item i // k; the encoder side is rebuilt with k rows per item (one more
                # encoder pass; generate() itself never takes this route)
                eng.gen_begin(ids0, feats0, mask0, k, cap0)
                cache = DecoderCache(eng, rows, cap0)
            eng._gen_pending = None
        new_cache = cache is None
        if new_cache:
            if input_ids is None:
                raise ValueError("the first cached step needs input_ids / image_features, or the tuple returned by "
                                 "model.get_encoder()(...) as encoder_outputs")
            enc = self.get_encoder()(input_ids, image_features, attention_mask, max_cache_length=max_cache_length)[0]
            cache = eng._gen_pending["cache"]
            eng._gen_pending = None
        elif not isinstance(cache, DecoderCache) or cache.engine is not eng:
            raise TypeError("decoder_cached_states must be the DecoderCache a previous cached forward of this model returned")
        cache.check()
        R, t = decoder_input_ids.shape
        if R != cache.rows:
            raise ValueError("decoder_input_ids has %d rows, the cache was created for %d" % (R, cache.rows))
        first = cache.length if t == cache.length + 1 else (0 if cache.length == 0 else None)
        if first is None:
            raise ValueError("decoder_input_ids must hold the %d cached positions plus the new one (got %d columns)"
                             % (cache.length, t))
        if t > cache.max_length:
            raise ValueError("decoder position %d exceeds the cache (max_cache_length=%d)" % (t, cache.max_length))
        logits = None
        for pos in range(first, t):
            logits = eng.gen_step(decoder_input_ids[:, pos], pos, want_logits=(pos == t - 1) and not want_hidden)
        cache.length = t
        if want_hidden:   # the bare model: the decoder's last hidden states of the new position (src/model/model.py:87-103)
            out = eng.gen_last_hidden().view(R, 1, int(self.config.d_model))
        else:
            V = int(self.config.vocab_size)
            out = logits[:, :V].clone().view(R, 1, V)
        if enc is None:
            enc = eng.gen_encoder_states()
        return (out, cache, enc)

    def _reorder_cache(self, past, beam_idx):
        """Reference src/model/mixins.py:419-434: `past` = ((encoder_states, encoder_mask), decoder_cached_states): the
        self-attention caches AND the rows' cross-attention keys / values / masks follow beam_idx (kmb_gen_reorder: the
        cached forward's rows are independent sequences, so the row -> encoder-item table is permuted with them), as do
        the encoder states and mask handed back to the caller."""
        (enc_out, enc_mask), cache = past
        cache.check()
        if cache.length > 0:
            cache.engine.gen_reorder(beam_idx, cache.length - 1)
        new_enc = enc_out if enc_out is None else enc_out.index_select(0, beam_idx.to(enc_out.device))
        new_mask = enc_mask if enc_mask is None else enc_mask.index_select(0, beam_idx.to(enc_mask.device))
        return ((new_enc, new_mask), cache)

    def prepare_inputs_for_generation(self, decoder_input_ids, past, attention_mask, use_cache, **kwargs):
        """Reference src/model/mixins.py:386-398."""
        assert past is not None, "past has to be defined for encoder_outputs"
        encoder_outputs, decoder_cached_states = past
        return {"input_ids": None, "image_features": None, "encoder_outputs": encoder_outputs,
                "decoder_cached_states": decoder_cached_states, "decoder_input_ids": decoder_input_ids,
                "attention_mask": attention_mask, "use_cache": use_cache}

    # ------------------------------------------------------------------ embeddings surface
    def get_encoder(self):
        """Reference src/model/mixins.py:436-437."""
        return _EncoderHandle(self)

    def get_input_embeddings(self):
        """nn.Embedding view of model.shared (reference src/model/model.py:105-106): shares the parameter's storage."""
        emb = nn.Embedding(self.config.vocab_size, self.config.d_model, padding_idx=self.config.pad_token_id,
                           _weight=self._p["model.shared.weight"].detach())
        return emb

    def get_output_embeddings(self):
        """The tied head as an nn.Linear made on the fly from model.shared, without bias (reference
        src/model/mixins.py:439-440, model.py:113-114: _make_linear_from_emb); shares the parameter's storage."""
        w = self._p["model.shared.weight"]
        lin = nn.Linear(w.shape[1], w.shape[0], bias=False, device=w.device, dtype=w.dtype)
        lin.weight.data = w.detach()
        return lin

    def resize_token_embeddings(self, new_num_tokens=None):
        """Reference src/model/mixins.py:442-455 (+ transformers 3.0.2 _get_resized_embeddings): model.shared becomes
        [new_num_tokens, d] -- the first min(old, new) rows are kept, new rows are N(0, init_std) -- and
        final_logits_bias is cut or zero-extended.  Only before the model is moved to the device (the engine's arenas
        are sized at .to(device); the reference's callers resize right after construction / from_pretrained)."""
        if new_num_tokens is None:
            return self.get_input_embeddings()
        if self._engine is not None:
            raise RuntimeError("resize_token_embeddings after .to(device) is not supported: resize first, then move the model")
        old = self._p["model.shared.weight"].detach()
        new_num_tokens = int(new_num_tokens)
        if new_num_tokens != old.shape[0]:
            g = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
            w = torch.randn((new_num_tokens, old.shape[1]), generator=g) * self.config.init_std
            n = min(old.shape[0], new_num_tokens)
            w[:n] = old[:n]
            self._p["model.shared.weight"] = nn.Parameter(w)
            flb = torch.zeros((1, new_num_tokens))
            flb[:, :n] = self._flb[:, :n]
            self._flb = flb
            self.config.vocab_size = new_num_tokens
        return self.get_input_embeddings()

    def _backward(self, loss_scale=1.0):
        self._engine.backward(loss_scale)
        if self._post_backward is not None:
            self._post_backward()

    def train_step_fwd_bwd(self, batch, loss_scale=1.0):
        """forward + backward without any host synchronisation; returns the device loss tensor [1]."""
        eng = self._need_engine()
        loss, _, _ = eng.forward(batch["input_ids"], batch["image_features"], batch.get("attention_mask"),
                                 batch.get("decoder_input_ids"), batch.get("decoder_attention_mask"), batch["labels"],
                                 train=self.training, need_grad=True, want_logits=False, want_encoder=False)
        self._backward(loss_scale)
        return loss

    # ------------------------------------------------------------------ generation
    @torch.no_grad()
    def generate(self, input_ids=None, image_features=None, max_length=None, min_length=None, do_sample=None,
                 early_stopping=None, num_beams=None, temperature=None, top_k=None, top_p=None,
                 repetition_penalty=None, bad_words_ids=None, bos_token_id=None, pad_token_id=None,
                 eos_token_id=None, length_penalty=None, no_repeat_ngram_size=None, num_return_sequences=None,
                 attention_mask=None, decoder_start_token_id=None, use_cache=None, return_scores=False,
                 **model_specific_kwargs):
        """Reference src/model/mixins.py:33-384 (+ transformers 3.0.2 _generate_beam_search /
        _generate_no_beam_search).  Encoder once, KV-cached decoder steps on the device, beam
        bookkeeping on the host exactly as the reference does it."""
        eng = self._need_engine()
        cfg = self.config

        def dflt(v, name):
            return v if v is not None else getattr(cfg, name)

        max_length = dflt(max_length, "max_length")
        min_length = dflt(min_length, "min_length")
        do_sample = dflt(do_sample, "do_sample")
        early_stopping = dflt(early_stopping, "early_stopping")
        num_beams = dflt(num_beams, "num_beams")
        temperature = dflt(temperature, "temperature")
        top_k = dflt(top_k, "top_k")
        top_p = dflt(top_p, "top_p")
        repetition_penalty = dflt(repetition_penalty, "repetition_penalty")
        pad_token_id = dflt(pad_token_id, "pad_token_id")
        eos_token_id = dflt(eos_token_id, "eos_token_id")
        length_penalty = dflt(length_penalty, "length_penalty")
        no_repeat_ngram_size = dflt(no_repeat_ngram_size, "no_repeat_ngram_size")
        bad_words_ids = dflt(bad_words_ids, "bad_words_ids")
        num_return_sequences = dflt(num_return_sequences, "num_return_sequences")
        decoder_start_token_id = dflt(decoder_start_token_id, "decoder_start_token_id")
        # argument validation of mixins.py:176-208
        assert input_ids is not None and input_ids.dim() == 2, "Input prompt should be of shape (batch_size, sequence length)."
        assert isinstance(max_length, int) and max_length > 0, "`max_length` should be a strictly positive integer."
        assert isinstance(min_length, int) and min_length >= 0, "`min_length` should be a positive integer."
        assert isinstance(do_sample, bool), "`do_sample` should be a boolean."
        assert isinstance(early_stopping, bool), "`early_stopping` should be a boolean."
        assert isinstance(num_beams, int) and num_beams > 0, "`num_beams` should be a strictly positive integer."
        assert temperature > 0, "`temperature` should be strictly positive."
        assert isinstance(top_k, int) and top_k >= 0, "`top_k` should be a positive integer."
        assert 0 <= top_p <= 1, "`top_p` should be between 0 and 1."
        assert length_penalty > 0, "`length_penalty` should be strictly positive."
        assert isinstance(num_return_sequences, int) and num_return_sequences > 0
        assert repetition_penalty >= 1.0, "`repetition_penalty` should be >= 1."
        assert isinstance(no_repeat_ngram_size, int) and no_repeat_ngram_size >= 0, "`no_repeat_ngram_size` should be a positive integer."
        assert bad_words_ids is None or (isinstance(bad_words_ids, list) and isinstance(bad_words_ids[0], list)), \
            "`bad_words_ids` is either `None` or a list of lists of tokens that should not be generated"
        # the score post-processing of transformers 3.0.2 (postprocess_next_token_scores: mixins.py:150-235 validates these
        # arguments, the loops apply them) needs every row's tokens on the host at every step: such searches take the
        # step-by-step host loop (like beam sampling), not the pipelined one
        processors_on = repetition_penalty != 1.0 or no_repeat_ngram_size > 0 or bad_words_ids is not None
        if not do_sample:
            if num_beams == 1:
                assert num_return_sequences == 1, "Greedy decoding will always produce the same output"
            else:
                assert num_beams >= num_return_sequences
        B = input_ids.shape[0]
        dev = eng.device
        if attention_mask is None:
            if pad_token_id is not None and bool((input_ids == pad_token_id).any()):
                attention_mask = input_ids.ne(pad_token_id).long()
            else:
                attention_mask = torch.ones_like(input_ids)
        eff_mult = num_return_sequences if do_sample else 1
        assert 1 < max_length, "The context has 1 number of tokens, but `max_length` is only %d" % max_length
        if do_sample and eff_mult > 1:  # sampling replicates the batch (mixins.py:259-262)
            rep = torch.arange(B).repeat_interleave(eff_mult)
            input_ids = input_ids[rep]
            attention_mask = attention_mask[rep]
            image_features = [image_features[i] for i in rep.tolist()]
            B = B * eff_mult
        V = cfg.vocab_size
        R = B * num_beams
        # fp32 validation mode (Engine.set_precision(True)): parity evidence, not a product path -- no KV cache, no fused decode
        # blocks; every step is an eval forward on the exact-fp32 kernels over the rows' tokens so far (the encoder runs once,
        # its fp32 states are handed back in), and the host loops below do the reference's bookkeeping on those logits
        fp32 = bool(getattr(eng, "fp32_mode", False))
        step_logits = None
        if fp32:
            step_logits = self._fp32_step_fn(eng, input_ids, image_features, attention_mask, num_beams)
        else:
            eng.gen_begin(input_ids, image_features, attention_mask, num_beams, max_length)
            # the device-side input validation is read back without waiting (the decode steps are enqueued while the encoder
            # still runs); a flagged batch raises at the first point where the host waits for the device anyway
            if os.environ.get("KMB_GEN_SYNC_CHECK") == "1":   # A/B knob: wait for the encoder before the first decode step
                eng.check_inputs()
            else:
                eng.check_inputs_begin()
        cur_len = 1

        if num_beams == 1:
            # Greedy / sampling without beams (transformers 3.0.2 _generate_no_beam_search): everything stays on the
            # device; the "every sentence has finished" test is read one step late from a pinned flag, so step t+1 is
            # enqueued before step t's flag is looked at, and the extra (all-pad) column of a late stop is dropped.
            unfinished = torch.ones(B, dtype=torch.long, device=dev)
            cols = [torch.full((B,), decoder_start_token_id, dtype=torch.long, device=dev)]
            flags = eng.pinned((max_length + 1,), torch.long)
            pending, keep = None, None
            while cur_len < max_length:
                logits = step_logits(torch.stack(cols, dim=1)) if fp32 else eng.gen_step(cols[-1], cur_len - 1)[:, :V]
                if processors_on:
                    _postprocess_next_token_scores(logits, torch.stack(cols, dim=1).tolist(), cur_len, min_length,
                                                   eos_token_id, repetition_penalty, no_repeat_ngram_size, bad_words_ids)
                elif eos_token_id is not None and cur_len < min_length:
                    logits[:, eos_token_id] = -float("inf")
                if do_sample:
                    lg = logits / temperature if temperature != 1.0 else logits
                    lg = _top_k_top_p_filtering(lg.clone(), top_k=top_k, top_p=top_p)
                    nxt = torch.multinomial(torch.softmax(lg, dim=-1), num_samples=1).squeeze(1)
                else:
                    nxt = torch.argmax(logits, dim=-1)
                tok = nxt * unfinished + pad_token_id * (1 - unfinished) if eos_token_id is not None else nxt
                cols.append(tok)
                cur_len += 1
                if eos_token_id is not None:
                    unfinished = unfinished * (tok != eos_token_id).long()
                    flags[cur_len].copy_(unfinished.max(), non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    if pending is not None:
                        pending[1].synchronize()
                        if int(flags[pending[0]]) == 0:     # finished one step ago: the reference stopped there
                            keep = pending[0]
                            break
                    pending = (cur_len, ev)
            out = torch.stack(cols[:keep] if keep is not None else cols, dim=1)
            if not fp32:
                eng.check_inputs_end()
            return out

        # Host bookkeeping on plain Python lists: indexing small CPU tensors element by element (as the reference does)
        # costs ~10 us per access and made a beam step 6x longer than its GPU work.
        hyps = [BeamHypotheses(num_beams, max_length, length_penalty, early_stopping) for _ in range(B)]
        # greedy beam search starts from beam 0 only; beam sampling lets every beam draw (HF 3.0.2 _generate_beam_search)
        beam_scores = [0.0 if ((i % num_beams) == 0 or do_sample) else -1e9 for i in range(R)]
        sampler = getattr(self, "_sampler", None) or (lambda probs, n: torch.multinomial(probs, num_samples=n))
        seqs = [[int(decoder_start_token_id)] for _ in range(R)]   # decoder inputs of every beam row
        done = [False] * B
        k = 2 * num_beams
        last_tokens = torch.full((R,), decoder_start_token_id, dtype=torch.long, device=dev)
        eos = -1 if eos_token_id is None else int(eos_token_id)

        def bookkeeping(next_scores, next_tokens, step_len):
            """One step of the reference's host-side beam bookkeeping (transformers 3.0.2 _generate_beam_search) from the
            step's sorted candidates; returns (new_scores, new_tokens, new_idx) and updates hyps / done."""
            new_scores, new_tokens, new_idx = [], [], []
            for b in range(B):
                if done[b]:
                    new_scores += [0.0] * num_beams
                    new_tokens += [pad_token_id] * num_beams
                    new_idx += [0] * num_beams
                    continue
                n_sent = 0
                for rank in range(k):
                    tid, tscore = next_tokens[b][rank], next_scores[b][rank]
                    beam_id, token_id = tid // V, tid % V
                    eff = b * num_beams + beam_id
                    if eos_token_id is not None and token_id == eos_token_id:
                        if rank >= num_beams:
                            continue
                        hyps[b].add(list(seqs[eff]), tscore)
                    else:
                        new_scores.append(tscore)
                        new_tokens.append(token_id)
                        new_idx.append(eff)
                        n_sent += 1
                    if n_sent == num_beams:
                        break
                done[b] = done[b] or hyps[b].is_done(max(next_scores[b]), step_len)
                assert n_sent == num_beams, "Beam should always be full"
            return new_scores, new_tokens, new_idx

        host_loop = do_sample or processors_on or fp32
        if not host_loop:
            # Greedy beam search, pipelined: the device picks the next step's beams itself (kmb_beam_merge_select: the
            # first num_beams non-EOS candidates, exactly what the bookkeeping below sends on), so step t+1 is enqueued
            # before the host has seen step t.  The host replays the reference's bookkeeping one step behind from the
            # candidates (one small pinned copy per step): hypotheses, `done`, and the decision to stop -- a stop costs
            # one decode step that is thrown away.  A `done` batch item keeps decoding on the device (the reference feeds
            # it pad tokens); its rows feed nothing that is read.
            staging = eng.pinned((max_length, B, k, 2), torch.int32)
            beam_scores_dev = torch.full((B, num_beams), -1e9, dtype=torch.float32, device=dev)   # = beam_scores, built on the device
            beam_scores_dev[:, 0] = 0.0
            beam_scores_dev = beam_scores_dev.view(-1)
            pending = None

            def replay(item):
                nonlocal beam_scores, seqs
                slot, ev, step_len = item
                ev.synchronize()
                eng.check_inputs_end()
                c = staging[slot]
                ns = c[:, :, 0].contiguous().view(torch.float32).tolist()
                nt = c[:, :, 1].tolist()
                new_scores, new_tokens, new_idx = bookkeeping(ns, nt, step_len)
                if all(done):
                    return True
                beam_scores = new_scores
                seqs = [seqs[j] + [t] for j, t in zip(new_idx, new_tokens)]
                return False

            while cur_len < max_length:
                ban = eos if (eos >= 0 and cur_len < min_length) else -1
                force = -1
                if cur_len == 1:
                    force = cfg.bos_token_id          # adjust_logits_during_generation, mixins.py:400-405
                last = cur_len == max_length - 1
                if last and eos_token_id is not None:
                    force = eos_token_id
                # A forced step's scores are 0 at the forced token and -inf elsewhere whatever the model says
                # (log_softmax of a row with one finite entry): the vocabulary projection is skipped, and on the LAST
                # step, whose keys / values nobody will read, the decoder as well.
                if force >= 0 and last:
                    logits = eng._gen_logits
                else:
                    logits = eng.gen_step(last_tokens, cur_len - 1, want_logits=force < 0)
                # the candidates land in the page-locked staging buffer straight from the kernel (no copy launch per step)
                # ... and _reorder_cache (mixins.py:419-434) by the same call: the launch that picks the beams permutes the history index
                cand, beam_scores_dev, last_tokens, beam_idx = eng.beam_step(
                    logits, num_beams, k, beam_scores_dev, force_token=force, ban_token=ban, eos_token=eos,
                    cand_out=staging[cur_len - 1], reorder_step=-1 if last else cur_len - 1)
                ev = torch.cuda.Event()
                ev.record()
                if pending is not None and replay(pending):
                    pending = None
                    break
                pending = (cur_len - 1, ev, cur_len)
                cur_len += 1
            if pending is not None:
                replay(pending)
        while host_loop and cur_len < max_length:
            # The reference's step-by-step host loop (HF 3.0.2 _generate_beam_search), for the searches whose scores are
            # post-processed on the host: beam-search multinomial sampling (do_sample branch, reached from mixins.py:336-361
            # with generate_text's --do_sample/--top_p/--top_k and --num_beams: no forced BOS/EOS; 2*num_beams draws per batch
            # item from softmax over the beams' filtered (log-prob + beam score) / T) and repetition_penalty /
            # no_repeat_ngram_size / bad_words_ids (postprocess_next_token_scores on the log-probabilities).
            logits = step_logits(torch.tensor(seqs, dtype=torch.long)) if fp32 else eng.gen_step(last_tokens, cur_len - 1)[:, :V].float()
            if not do_sample:   # adjust_logits_during_generation (mixins.py:400-405): forced BOS / EOS, greedy beams only
                force = cfg.bos_token_id if cur_len == 1 else (eos_token_id if (cur_len == max_length - 1 and eos_token_id is not None) else None)
                if force is not None:
                    kept = logits[:, force].clone()
                    logits.fill_(-float("inf"))
                    logits[:, force] = kept
            add = torch.tensor(beam_scores, dtype=torch.float32).to(dev)
            sc = torch.log_softmax(logits, dim=-1)
            # min_length (EOS -inf AFTER log_softmax) and the other post-processing of transformers 3.0.2
            _postprocess_next_token_scores(sc, seqs if processors_on else None, cur_len, min_length, eos_token_id,
                                           repetition_penalty, no_repeat_ngram_size, bad_words_ids)
            sc = sc + add[:, None]
            if do_sample:
                if temperature != 1.0:
                    sc = sc / temperature
                sc = _top_k_top_p_filtering(sc, top_k=top_k, top_p=top_p, min_tokens_to_keep=2).view(B, num_beams * V)
                drawn = sampler(torch.softmax(sc, dim=-1), k)
                ns = torch.gather(sc, -1, drawn)
                ns, order = torch.sort(ns, descending=True, dim=1)
                next_scores = ns.cpu().tolist()
                next_tokens = torch.gather(drawn, -1, order).cpu().tolist()
            else:
                ns, nt_ = torch.topk(sc.view(B, num_beams * V), k, dim=1, largest=True, sorted=True)
                next_scores, next_tokens = ns.cpu().tolist(), nt_.cpu().tolist()
            new_scores, new_tokens, new_idx = bookkeeping(next_scores, next_tokens, cur_len)
            if all(done):
                break
            beam_scores = new_scores
            seqs = [seqs[j] + [t] for j, t in zip(new_idx, new_tokens)]
            tok_idx = torch.tensor([new_tokens, new_idx], dtype=torch.long).to(dev)
            last_tokens = tok_idx[0].contiguous()
            if not fp32:
                eng.gen_reorder(tok_idx[1], cur_len - 1)   # _reorder_cache, mixins.py:419-434
            cur_len += 1
        for b in range(B):
            if done[b]:
                continue
            for beam_id in range(num_beams):
                eff = b * num_beams + beam_id
                hyps[b].add(list(seqs[eff]), beam_scores[eff])
        best, best_scores, lens = [], [], []
        nret_each = 1 if do_sample else num_return_sequences   # sampling replicated the batch instead (mixins.py:259-262)
        for h in hyps:
            sh = sorted(h.beams, key=lambda x: x[0])
            for _ in range(nret_each):
                sc, hyp = sh.pop()
                best.append(hyp)
                best_scores.append(sc)
                lens.append(len(hyp))
        if min(lens) != max(lens):
            L = min(max(lens) + 1, max_length)
            rows = []
            for hyp, n in zip(best, lens):   # hypothesis, EOS behind it if it stopped early, pads (one tensor build, not one per row)
                row = list(hyp) + ([eos_token_id] if n < max_length else [])
                rows.append(row + [pad_token_id] * (L - len(row)))
            out = torch.tensor(rows, dtype=torch.long)
        else:
            out = torch.tensor(best, dtype=torch.long)
        if not fp32:
            eng.check_inputs_end()
        out = out.to(dev)
        return (out, torch.tensor(best_scores)) if return_scores else out

    @staticmethod
    def _fp32_step_fn(eng, input_ids, image_features, attention_mask, num_beams):
        """generate() in the fp32 validation mode: returns f(rows' decoder tokens [R, t]) -> fp32 logits [R, V] of the last
        position.  The encoder runs once (first call) and its float32 states, repeated per beam, are handed back in
        (reference src/model/model.py:76-83); the decoder gets NO padding mask -- the reference's cached steps
        (src/model/model.py:384-397, use_cache) build none, finished rows are fed pad tokens there too."""
        B = input_ids.shape[0]
        rep = torch.arange(B).repeat_interleave(num_beams)
        ids, am = input_ids[rep.to(input_ids.device)], attention_mask[rep.to(attention_mask.device)]
        feats = [image_features[i] for i in rep.tolist()]
        state = {}

        def step(rows):
            rows = rows.to(eng.device)
            ones = torch.ones_like(rows)
            if "enc" not in state:
                _, logits, enc = eng.forward(ids, feats, am, rows, ones, None, train=False, need_grad=False, want_logits=True,
                                             want_encoder=True)
                state["enc"] = enc
            else:
                _, logits, _ = eng.forward(ids, feats, am, rows, ones, None, train=False, need_grad=False, want_logits=True,
                                           want_encoder=False, encoder_states=state["enc"])
            return logits[:, -1, :].float().clone()

        return step

    # ------------------------------------------------------------------ checkpoints
    def save_pretrained(self, save_directory):
        """transformers `save_pretrained`: config.json + pytorch_model.bin with the reference's key names
        (vcg_train.py:249-254)."""
        os.makedirs(save_directory, exist_ok=True)
        self.config.save_pretrained(save_directory)
        torch.save(self.state_dict(), os.path.join(save_directory, WEIGHTS_NAME))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, state_dict=None,
                        error_on_mismatch=False, **kwargs):
        """Reference src/model/mixins.py:551-883 for local paths: directory with pytorch_model.bin or a file.
        Keys listed in `config.partial_load` are copied into the top-left slice of a larger parameter
        (mixins.py:511-530); other shape mismatches are skipped silently like the reference (mixins.py:856-863,
        the RuntimeError there is built but never raised).  The model is returned in eval mode (:867)."""
        path = pretrained_model_name_or_path
        if config is None:
            config = MultiModalBartConfig.from_pretrained(path)
        model = cls(config, *model_args)
        if state_dict is None:
            f = os.path.join(path, WEIGHTS_NAME) if os.path.isdir(path) else path
            if not os.path.isfile(f):
                raise EnvironmentError("no %s under %s (hub download is not available offline)" % (WEIGHTS_NAME, path))
            state_dict = torch.load(f, map_location="cpu")
        partial = set(getattr(config, "partial_load", ()) or ())
        own = {n: model._p[n] for n in model._names}
        with torch.no_grad():
            for key, val in state_dict.items():
                tgt_name = "model.shared.weight" if key in _TIED else key
                if key == "final_logits_bias":
                    tgt = model._flb
                elif tgt_name in own:
                    tgt = own[tgt_name]
                else:
                    continue
                if tuple(tgt.shape) == tuple(val.shape):
                    tgt.copy_(val)
                elif key in partial and val.dim() == tgt.dim() and all(a <= b for a, b in zip(val.shape, tgt.shape)):
                    tgt[tuple(slice(0, s) for s in val.shape)] = val
        model.eval()
        return model


class _DeviceLossDict(dict):
    """`outputs[0]` of MultiModalBartForPreTraining: {'loss', 'lm_loss', 'mrm_loss', ...} of 0-dim device tensors."""


class MultiModalBartForPreTraining(MultiModalBartForConditionalGeneration):
    """Multi-task pre-training model (reference src/model/model.py:125-309): the conditional-generation model plus
    three BartClassificationHeads on the decoder states -- masked-region modelling (KL divergence to soft labels),
    attribute prediction and relation prediction (cross entropy) -- and the weighted loss
    lm_loss_factor * lm + mrm_loss_factor * mrm + attribute_loss_factor * attr + relation_loss_factor * rel."""
    _with_heads = True

    def forward(self, input_ids, image_features, attention_mask=None, encoder_outputs=None, decoder_input_ids=None,
                decoder_attention_mask=None, decoder_cached_states=None, labels=None, mrm_labels=None, mrm_mask=None,
                attribute_labels=None, attribute_mask=None, relation_labels=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_logits=None, **unused):
        """Reference src/model/model.py:162-309.  With any label the loss dict comes first: (losses, logits[, decoder hidden
        states, decoder attentions], encoder states[, encoder hidden states, encoder attentions]) -- `outputs[1:]` of the bare
        model behind the logits (:291, :309); without labels it is the conditional-generation forward (cache included).
        `encoder_outputs` is passed through to the model as there (:225-242)."""
        eng = self._need_engine()
        cfg = self.config
        if labels is None and mrm_labels is None and attribute_labels is None and relation_labels is None:
            return super().forward(input_ids, image_features, attention_mask=attention_mask, encoder_outputs=encoder_outputs,
                                   decoder_input_ids=decoder_input_ids, decoder_attention_mask=decoder_attention_mask,
                                   decoder_cached_states=decoder_cached_states, use_cache=use_cache,
                                   output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                                   return_logits=return_logits, **unused)
        # any label switches the cache off (model.py:222-223); decoder_cached_states then has no meaning
        output_attentions = cfg.output_attentions if output_attentions is None else output_attentions
        output_hidden_states = cfg.output_hidden_states if output_hidden_states is None else output_hidden_states
        enc_states = None
        if encoder_outputs is not None:
            enc_states = encoder_outputs[0] if isinstance(encoder_outputs, (tuple, list)) else encoder_outputs
        if mrm_labels is not None and mrm_mask is None:
            raise ValueError('"mrm_mask" cannot be None while "mrm_labels" is set')   # model.py:228-229
        B, T = decoder_input_ids.shape
        dev = eng.device
        mrm = attr = rel = None
        if mrm_labels is not None:
            rows = torch.nonzero(mrm_mask.reshape(-1).to(dev), as_tuple=False).reshape(-1)
            tgt = torch.cat([m.to(dev) for m in mrm_labels], 0) if len(mrm_labels) else torch.zeros((0, cfg.num_labels))
            if rows.numel() != tgt.shape[0]:
                raise RuntimeError("mrm_mask selects %d rows but mrm_labels holds %d" % (rows.numel(), tgt.shape[0]))
            mrm = (rows, tgt)
        if attribute_labels is not None:
            rows = torch.nonzero(attribute_mask.reshape(-1).to(dev), as_tuple=False).reshape(-1)
            lab = torch.cat([a.to(dev).reshape(-1) for a in attribute_labels], 0) if len(attribute_labels) else torch.zeros(0)
            if rows.numel() != lab.numel():
                raise RuntimeError("attribute_mask selects %d rows but attribute_labels holds %d" % (rows.numel(), lab.numel()))
            attr = (rows, lab)
        if relation_labels is not None:  # list (per sample) of dicts {object_index, subject_index, label} (model.py:274-280)
            ro, rs, lab = [], [], []
            for i, rels in enumerate(relation_labels):
                for r in rels:
                    ro.append(i * T + int(r["object_index"]))
                    rs.append(i * T + int(r["subject_index"]))
                    lab.append(int(r["label"]))
            rel = (torch.tensor(ro, dtype=torch.int32), torch.tensor(rs, dtype=torch.int32),
                   torch.tensor(lab, dtype=torch.int64))
        lm_labels = None
        if labels is not None:
            lm_labels = labels.clone()
            lm_labels[lm_labels == cfg.cls_token_id] = -100   # model.py:297-298
        need_grad = torch.is_grad_enabled()
        factors = (float(cfg.lm_loss_factor), float(cfg.mrm_loss_factor), float(cfg.attribute_loss_factor),
                   float(cfg.relation_loss_factor))
        lazy_enc = need_grad and enc_states is None and not eng.fp32_mode   # (the benchmarked pre-training step copied ~100 MB of encoder states nobody read)
        res = eng.forward_pretrain(input_ids, image_features, attention_mask, decoder_input_ids,
                                   decoder_attention_mask, lm_labels, mrm=mrm, attr=attr, rel=rel,
                                   factors=factors, train=self.training, need_grad=need_grad,
                                   want_logits=bool(return_logits), encoder_states=enc_states,
                                   want_encoder=not lazy_enc)
        losses, logits = res[0], res[1]
        enc = LazyEncoderStates(eng) if lazy_enc else res[2]
        dec_extra, enc_extra = (), ()
        if output_hidden_states:
            dec_extra += (eng.hidden_states(1)[:-1],)
            if enc_states is None:
                enc_extra += (list(eng.hidden_states(0)),)
        if output_attentions:
            dec_extra += (eng.attention_probs(1),)
            if enc_states is None:
                enc_extra += (list(eng.attention_probs(0)),)
        if enc_states is not None and isinstance(encoder_outputs, (tuple, list)):
            enc_extra = tuple(x for x in encoder_outputs[1:] if isinstance(x, torch.Tensor) or x)
        total = (_LossFn.apply(self._anchor, self, losses[0:1], enc_states if torch.is_tensor(enc_states) else None)
                 if need_grad else losses[0])
        out = _DeviceLossDict(loss=total)
        if lm_labels is not None:   # model.py:293-302: without LM labels the term is 0 and the key is absent
            out["lm_loss"] = losses[1]
        if mrm is not None and mrm[0].numel() > 0:
            out["mrm_loss"] = losses[2]
        if attr is not None and attr[0].numel() > 0:
            out["attribute_loss"] = losses[3]
        if rel is not None and rel[0].numel() > 0:
            out["relation_loss"] = losses[4]
        if logits is None:
            logits = LazyLogits(eng)
        return (out, logits) + dec_extra + (enc,) + enc_extra


def _postprocess_next_token_scores(scores, prev_ids, cur_len, min_length, eos_token_id, repetition_penalty=1.0,
                                   no_repeat_ngram_size=0, bad_words_ids=None):
    """transformers 3.0.2 GenerationMixin.postprocess_next_token_scores on a device tensor `scores` [rows, V], in its
    order: repetition penalty (CTRL: a score < 0 is multiplied by the penalty, a score >= 0 divided), EOS banned below
    min_length, no-repeat n-grams (calc_banned_ngram_tokens), bad words (calc_banned_bad_words_ids, including its length
    test against the number of ROWS).  prev_ids: the rows' tokens so far as Python lists (None: nothing but min_length).
    The sets are built on the host, each kind is applied with one indexed update."""
    dev = scores.device
    if prev_ids is not None and repetition_penalty != 1.0:
        rows = [i for i, row in enumerate(prev_ids) for _ in set(row)]
        cols = [t for row in prev_ids for t in set(row)]
        if rows:
            ri, ci = torch.tensor(rows, device=dev), torch.tensor(cols, device=dev)
            v = scores[ri, ci]
            scores[ri, ci] = torch.where(v < 0, v * repetition_penalty, v / repetition_penalty)
    if eos_token_id is not None and cur_len < min_length:
        scores[:, eos_token_id] = -float("inf")
    banned_rows, banned_cols = [], []
    if prev_ids is not None and no_repeat_ngram_size > 0 and cur_len + 1 >= no_repeat_ngram_size:
        n = no_repeat_ngram_size
        for i, row in enumerate(prev_ids):
            seen = {}
            for ng in zip(*[row[j:] for j in range(n)]):
                seen.setdefault(tuple(ng[:-1]), []).append(ng[-1])
            for t in seen.get(tuple(row[cur_len + 1 - n:cur_len]), []):
                banned_rows.append(i)
                banned_cols.append(t)
    if prev_ids is not None and bad_words_ids is not None:
        n_rows = len(prev_ids)
        for i, row in enumerate(prev_ids):
            for seq in bad_words_ids:
                assert len(seq) > 0, "Banned words token sequences {} cannot have an empty list".format(bad_words_ids)
                head = list(seq[:-1])
                if len(head) == 0 or (len(head) <= n_rows and row[-len(head):] == head):
                    banned_rows.append(i)
                    banned_cols.append(seq[-1])
    if banned_rows:
        scores[torch.tensor(banned_rows, device=dev), torch.tensor(banned_cols, device=dev)] = -float("inf")
    return scores


def _top_k_top_p_filtering(logits, top_k=0, top_p=1.0, filter_value=-float("inf"), min_tokens_to_keep=1):
    """transformers 3.0.2 `top_k_top_p_filtering` (sampling path of generate)."""
    if top_k > 0:
        top_k = min(max(top_k, min_tokens_to_keep), logits.size(-1))
        remove = logits < torch.topk(logits, top_k)[0][..., -1, None]
        logits[remove] = filter_value
    if top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        cum = torch.cumsum(torch.softmax(sorted_logits, dim=-1), dim=-1)
        rm = cum > top_p
        if min_tokens_to_keep > 1:
            rm[..., :min_tokens_to_keep] = 0
        rm[..., 1:] = rm[..., :-1].clone()
        rm[..., 0] = 0
        remove = rm.scatter(1, sorted_indices, rm)
        logits[remove] = filter_value
    return logits


class MultiModalBartModel(MultiModalBartForConditionalGeneration):
    """The bare encoder-decoder (reference src/model/model.py:27-103): forward returns the decoder's last hidden states
    followed by the encoder's, `decoder_outputs + encoder_outputs` with the empty entries filtered out (:100-103) --
    no LM head, no loss.  State-dict keys carry no `model.` prefix (this IS the reference's `.model` attribute); a
    conditional-generation checkpoint loads as well.  It runs on the same engine, whose arena simply keeps the tied
    matrix for the two embedding lookups."""

    def forward(self, input_ids, image_features, attention_mask=None, decoder_input_ids=None, encoder_outputs=None,
                decoder_attention_mask=None, decoder_cached_states=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, **unused):
        """Reference src/model/model.py:39-103: (decoder states[, cache][, decoder hidden states, decoder attentions], encoder
        states[, encoder hidden states, encoder attentions]).  use_cache (default config.use_cache, _resolve_use_cache): the
        KV-cached step -- decoder states of the new position [rows, 1, d], the DecoderCache, the encoder states."""
        eng = self._need_engine()
        use_cache = self._resolve_use_cache(use_cache, False, decoder_input_ids, decoder_cached_states, output_attentions,
                                            output_hidden_states)
        if use_cache or decoder_cached_states is not None:
            if output_attentions or output_hidden_states:
                raise NotImplementedError("attention / hidden-state outputs are not materialised by the fused decode blocks")
            return self._forward_cached(input_ids, image_features, attention_mask, encoder_outputs, decoder_input_ids,
                                        decoder_cached_states, unused.get("max_cache_length"), want_hidden=True)
        output_attentions = self.config.output_attentions if output_attentions is None else output_attentions
        output_hidden_states = self.config.output_hidden_states if output_hidden_states is None else output_hidden_states
        if decoder_input_ids is None:   # transformers 3.0.2 _prepare_bart_decoder_inputs: shift_tokens_right(input_ids)
            pad = self.config.pad_token_id
            prev = input_ids.clone()
            idx_eos = (input_ids.ne(pad).sum(dim=1) - 1).unsqueeze(-1)
            prev[:, 0] = input_ids.gather(1, idx_eos).squeeze()
            prev[:, 1:] = input_ids[:, :-1]
            decoder_input_ids = prev
        enc_states = None
        if encoder_outputs is not None:
            assert isinstance(encoder_outputs, tuple)   # model.py:84
            enc_states = encoder_outputs[0]
        if decoder_attention_mask is None and bool((decoder_input_ids == self.config.pad_token_id).any()):
            decoder_attention_mask = decoder_input_ids.ne(self.config.pad_token_id).long()
        _, _, enc, dec = eng.forward(input_ids, image_features, attention_mask, decoder_input_ids, decoder_attention_mask,
                                     None, train=self.training, need_grad=False, want_logits=False, want_encoder=True,
                                     encoder_states=enc_states, want_decoder_states=True, skip_head=True)
        dec_extra, enc_extra = (), ()
        if output_hidden_states:
            dec_extra += (eng.hidden_states(1)[:-1],)
            if enc_states is None:
                enc_extra += (list(eng.hidden_states(0)),)
        if output_attentions:
            dec_extra += (eng.attention_probs(1),)
            if enc_states is None:
                enc_extra += (list(eng.attention_probs(0)),)
        if enc_states is not None:
            enc_extra = tuple(x for x in encoder_outputs[1:] if isinstance(x, torch.Tensor) or x)
        return (dec,) + dec_extra + (enc,) + enc_extra

    def generate(self, *args, **kwargs):
        raise AttributeError("MultiModalBartModel has no LM head; use MultiModalBartForConditionalGeneration.generate")

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        sd.pop("final_logits_bias", None)
        return {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}

    def load_state_dict(self, state_dict, strict=True):
        full = {(k if (k.startswith("model.") or k == "final_logits_bias") else "model." + k): v
                for k, v in state_dict.items()}
        return super().load_state_dict(full, strict=strict)
