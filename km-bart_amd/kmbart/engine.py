"""Python host of libkmbart_hip.so: owns the device arenas (torch tensors) and forwards calls
through the C-ABI.  torch is used for memory, streams and torch.distributed only."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import KmbAdamW, KmbBatch, KmbConfig, KmbForwardOpts, KmbPretrain, check, ptr

_CFG_INT_FIELDS = ("vocab_size", "d_model", "encoder_layers", "decoder_layers", "encoder_attention_heads",
                   "decoder_attention_heads", "encoder_ffn_dim", "decoder_ffn_dim", "max_position_embeddings",
                   "extra_pos_embeddings", "image_feature_size", "pad_token_id", "bos_token_id", "eos_token_id",
                   "img_feat_id", "cls_token_id")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack_features(image_features, feat_dim, device):
    """list[B] of [R_i, F] fp32 (R_i may be 0 -> torch.empty(0), collation.py:73-76)
    -> (packed [Ntot, F] fp32 on device, offsets int32 [B+1] on device, Ntot)."""
    if hasattr(image_features, "packed") and hasattr(image_features, "offsets"):   # kmbart.data.PackedFeatures
        return (image_features.packed.to(device=device, dtype=torch.float32).contiguous(),
                image_features.offsets.to(device=device, dtype=torch.int32).contiguous(), image_features.n_total)
    lens = [int(x.shape[0]) if x.dim() == 2 else 0 for x in image_features]
    offs = [0]
    for n in lens:
        offs.append(offs[-1] + n)
    non_empty = [x.to(device=device, dtype=torch.float32) for x, n in zip(image_features, lens) if n > 0]
    for x in non_empty:
        if x.shape[1] != feat_dim:
            raise ValueError("image feature width %d != config.image_feature_size %d" % (x.shape[1], feat_dim))
    if not non_empty:
        packed = torch.zeros((1, feat_dim), device=device)
    elif len(non_empty) == 1:
        packed = non_empty[0].contiguous()
    elif torch.device(device).type != "cuda":
        packed = torch.cat(non_empty, 0).contiguous()   # host-side use (tests of the batch layout): no kernel to launch
    else:
        # ONE launch per 128 tensors (kmb_pack_features) instead of torch.cat, which on this stack is a batched kernel plus ~one blit
        # per tensor (70 copy launches in front of a 64-sample generate); the sources are kept alive by the caller's `keep` list
        srcs = [x.contiguous() for x in non_empty]
        packed = torch.empty((offs[-1], feat_dim), dtype=torch.float32, device=device)
        ptrs = (C.c_void_p * len(srcs))(*[x.data_ptr() for x in srcs])
        rows = (C.c_int32 * len(srcs))(*[int(x.shape[0]) for x in srcs])
        with torch.cuda.device(device):
            check(_lib.load().kmb_pack_features(ptrs, rows, len(srcs), int(feat_dim), ptr(packed), _stream()))
        packed._kmb_sources = srcs   # the kernel reads them in stream order: alive as long as the packed buffer is
    offsets = torch.tensor(offs, dtype=torch.int32).to(device)
    return packed, offsets, offs[-1]


class Engine:
    """One model replica on one GPU."""

    def __init__(self, config, device, with_heads=False):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.KmbError("the KM-BART hot path needs an MI355X (HIP) device; there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.KmbError("Engine device must be a HIP device ('cuda:N'), got %s" % device)
        self.config = config
        c = KmbConfig()
        for f in _CFG_INT_FIELDS:
            setattr(c, f, int(getattr(config, f)))
        c.scale_embedding = 1 if getattr(config, "scale_embedding", False) else 0
        c.dropout = float(config.dropout)
        c.attention_dropout = float(config.attention_dropout)
        c.activation_dropout = float(config.activation_dropout)
        c.layer_norm_eps = 1e-5
        if with_heads:  # MultiModalBartForPreTraining (reference src/model/model.py:133-158)
            if float(getattr(config, "classif_dropout", 0.0)) != 0.0:
                raise NotImplementedError("classif_dropout != 0 is not implemented (pretrain_base.json uses 0.0)")
            c.num_labels = int(config.num_labels)
            c.num_attributes = int(config.num_attributes)
            c.num_relations = int(config.num_relations)
        self._ccfg = c
        h = C.c_void_p()
        check(self.lib.kmb_create(C.byref(c), C.byref(h)))
        self.h = h
        with torch.cuda.device(self.device):
            n = self.lib.kmb_arena_elems(h)
            nb = self.lib.kmb_bf16_arena_elems(h)
            self.n = n
            self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.exp_avg = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.params_bf16 = torch.zeros(nb, dtype=torch.bfloat16, device=self.device)
            self.final_logits_bias = torch.zeros(int(config.vocab_size), dtype=torch.float32, device=self.device)
            check(self.lib.kmb_bind_arenas(h, ptr(self.params), ptr(self.grads), ptr(self.exp_avg),
                                           ptr(self.exp_avg_sq), ptr(self.params_bf16), ptr(self.final_logits_bias)))
        self.index = {}  # name -> (offset, rows, cols)
        name, off, rows, cols = C.c_char_p(), C.c_int64(), C.c_int32(), C.c_int32()
        for i in range(self.lib.kmb_param_count(h)):
            check(self.lib.kmb_param_info(h, i, C.byref(name), C.byref(off), C.byref(rows), C.byref(cols)))
            self.index[name.value.decode()] = (off.value, rows.value, cols.value)
        self.logits_ld = self.lib.kmb_logits_ld(h)
        self.workspace = None
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._keep = []  # tensors the in-flight kernels read
        self.step_count = 0
        self.fwd_serial = 0      # bumped by every call that rewrites the workspace (LazyLogits validity)
        self.gen_logits_bytes = 4   # bytes per element of the decode step's logits buffer
        self.fp32_mode = False

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.kmb_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------------
    def view(self, arena, name):
        off, rows, cols = self.index[name]
        t = arena[off: off + rows * cols]
        return t.view(cols) if rows == 1 else t.view(rows, cols)

    def sync_params(self):
        """fp32 master -> bf16 mirror (after init / load_state_dict / manual edits)."""
        with torch.cuda.device(self.device):
            check(self.lib.kmb_sync_params(self.h, _stream()))

    def set_precision(self, fp32):
        """fp32 validation mode (True) / bf16 product mode (False).  Mode True runs the eval-mode forward with float
        activations on exact-fp32 kernels against the fp32 master weights (parity evidence, never the measured path);
        encoder states then come back as float32 tensors."""
        torch.cuda.synchronize(self.device)
        check(self.lib.kmb_set_precision(self.h, 1 if fp32 else 0))
        self.fp32_mode = bool(fp32)
        self.workspace = None
        self.fwd_serial += 1

    @property
    def act_dtype(self):
        return torch.float32 if self.fp32_mode else torch.bfloat16

    def set_seed(self, seed):
        check(self.lib.kmb_set_seed(self.h, C.c_uint64(int(seed) & (2 ** 64 - 1))))

    # ---- workspace --------------------------------------------------------------------------
    def _ensure_ws(self, nbytes):
        if self.workspace is None or self.workspace.numel() < nbytes:
            torch.cuda.synchronize(self.device)
            self.workspace = None
            self.workspace = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=self.device)
            check(self.lib.kmb_bind_workspace(self.h, ptr(self.workspace), self.workspace.numel()))
            self._poison()

    def _poison(self):
        """KMB_POISON=1 (diagnostic, tests/conftest.py --poison): every byte of the workspace becomes 0xFF -- a NaN in
        bf16 and in fp32, -1 in the integer tables -- when it is allocated and before every forward / generate, so a
        kernel that reads activation, scratch or slab memory nobody wrote in THIS call returns NaN deterministically
        instead of whatever the allocator's block happened to hold.  (torch.empty is the product behaviour.)"""
        if self.workspace is not None and os.environ.get("KMB_POISON") == "1":
            self.workspace.fill_(0xFF)
            # (re-binding tells the library that nothing it left in the workspace -- e.g. the packed decoder weights of the last
            #  generate -- is there any more)
            check(self.lib.kmb_bind_workspace(self.h, ptr(self.workspace), self.workspace.numel()))

    def _batch(self, input_ids, image_features, attention_mask, decoder_input_ids, decoder_attention_mask, labels):
        dev = self.device

        def i64(t):
            return None if t is None else t.to(device=dev, dtype=torch.int64).contiguous()

        input_ids = i64(input_ids)
        B, S = input_ids.shape
        packed, offsets, ntot = pack_features(image_features, int(self.config.image_feature_size), dev)
        attention_mask = i64(attention_mask)
        decoder_input_ids = i64(decoder_input_ids)
        decoder_attention_mask = i64(decoder_attention_mask)
        labels = i64(labels)
        T = decoder_input_ids.shape[1] if decoder_input_ids is not None else 1
        b = KmbBatch(B=B, S=S, T=T, input_ids=ptr(input_ids), attention_mask=ptr(attention_mask),
                     image_features=ptr(packed), feat_offsets=ptr(offsets), n_features=ntot,
                     decoder_input_ids=ptr(decoder_input_ids), decoder_attention_mask=ptr(decoder_attention_mask),
                     labels=ptr(labels))
        keep = [input_ids, packed, offsets, attention_mask, decoder_input_ids, decoder_attention_mask, labels]
        return b, keep, (B, S, T, ntot)

    # ---- training step ----------------------------------------------------------------------
    def forward(self, input_ids, image_features, attention_mask=None, decoder_input_ids=None,
                decoder_attention_mask=None, labels=None, train=False, need_grad=False, want_logits=False,
                want_encoder=True, encoder_states=None, want_decoder_states=False, skip_head=False):
        """Returns (loss [1] or None, logits [B,T,V] fp32 or None, encoder_out [B,S,D] or None); with
        want_decoder_states a 4th element, the last decoder hidden states [B,T,D].  `encoder_states` [B,S,D] skips the
        encoder (reference src/model/model.py:76-83); activations are bf16 (float32 in the fp32 validation mode)."""
        with torch.cuda.device(self.device):
            b, keep, (B, S, T, ntot) = self._batch(input_ids, image_features if encoder_states is None else [],
                                                   attention_mask, decoder_input_ids, decoder_attention_mask, labels)
            self._ensure_ws(self.lib.kmb_workspace_bytes(self.h, B, S, T, ntot))
            self._poison()
            D = int(self.config.d_model)
            logits = None
            if want_logits:
                logits = torch.empty((B * T, self.logits_ld), dtype=torch.float32, device=self.device)
            enc = None
            if want_encoder:
                enc = torch.empty((B, S, D), dtype=self.act_dtype, device=self.device)
            loss = torch.empty(1, dtype=torch.float32, device=self.device) if labels is not None else None
            opts = KmbForwardOpts()
            dec = None
            if encoder_states is not None:
                if tuple(encoder_states.shape) != (B, S, D):
                    raise ValueError("encoder_outputs[0] must be [batch, src_len, d_model] = %s, got %s"
                                     % ((B, S, D), tuple(encoder_states.shape)))
                encoder_states = encoder_states.to(device=self.device, dtype=self.act_dtype).contiguous()
                opts.encoder_states = ptr(encoder_states)
                keep.append(encoder_states)
            if want_decoder_states:
                dec = torch.empty((B, T, D), dtype=self.act_dtype, device=self.device)
                opts.decoder_states_out = ptr(dec)
            opts.skip_head = 1 if skip_head else 0
            self.fwd_serial += 1
            check(self.lib.kmb_forward_ex(self.h, C.byref(b), C.byref(opts), 1 if train else 0, 1 if need_grad else 0,
                                          ptr(loss), ptr(logits), ptr(enc), _stream()))
            self._keep = keep
            if logits is not None:
                logits = logits.view(B, T, self.logits_ld)[:, :, : int(self.config.vocab_size)]
            self._last_bt = (B, T)
            self._last_S = S
            return (loss, logits, enc, dec) if want_decoder_states else (loss, logits, enc)

    def hidden_states(self, which):
        """`output_hidden_states` of the forward still in the workspace: tuple of [B, T, d] activations of the encoder
        (which = 0: embedding output + every layer's output) or decoder (which = 1) stack."""
        B, T = self._last_bt
        rows_T = T if which == 1 else self._last_S
        n = int(self.config.encoder_layers if which == 0 else self.config.decoder_layers) + 1
        out = []
        with torch.cuda.device(self.device):
            for i in range(n):
                t = torch.empty((B, rows_T, int(self.config.d_model)), dtype=torch.bfloat16, device=self.device)
                check(self.lib.kmb_hidden_state(self.h, which, i, ptr(t), _stream()))
                out.append(t)
        return tuple(out)

    def encoder_last_state(self):
        """[B, S, d] bf16: the encoder output of the forward still in the workspace (what want_encoder=True copies eagerly)."""
        B, _ = self._last_bt
        t = torch.empty((B, self._last_S, int(self.config.d_model)), dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_hidden_state(self.h, 0, int(self.config.encoder_layers), ptr(t), _stream()))
        return t

    def encoder_states_grad(self):
        """dL / d(encoder output) of the backward that just ran, bf16 [B, S, d]: what autograd hands to a tensor passed as
        `encoder_outputs` (src/model/model.py:76-83)."""
        B, _ = self._last_bt
        t = torch.empty((B, self._last_S, int(self.config.d_model)), dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_encoder_states_grad(self.h, ptr(t), _stream()))
        return t

    def attention_probs(self, which):
        """`output_attentions`: tuple over layers of fp32 [B, H, T, T] self-attention probabilities."""
        B, T = self._last_bt
        Tq = T if which == 1 else self._last_S
        H = int(self.config.encoder_attention_heads if which == 0 else self.config.decoder_attention_heads)
        n = int(self.config.encoder_layers if which == 0 else self.config.decoder_layers)
        out = []
        with torch.cuda.device(self.device):
            for l in range(n):
                t = torch.empty((B, H, Tq, Tq), dtype=torch.float32, device=self.device)
                check(self.lib.kmb_attention_probs(self.h, which, l, ptr(t), _stream()))
                out.append(t)
        return tuple(out)

    def last_logits(self):
        """fp32 logits [B,T,V] of the decoder states the last forward left in the workspace (one head GEMM)."""
        B, T = self._last_bt
        with torch.cuda.device(self.device):
            logits = torch.empty((B * T, self.logits_ld), dtype=torch.float32, device=self.device)
            check(self.lib.kmb_last_logits(self.h, ptr(logits), _stream()))
        return logits.view(B, T, self.logits_ld)[:, :, : int(self.config.vocab_size)]

    def forward_pretrain(self, input_ids, image_features, attention_mask, decoder_input_ids, decoder_attention_mask,
                         labels, mrm=None, attr=None, rel=None, factors=(1.0, 1.0, 1.0, 1.0), train=False,
                         need_grad=False, want_logits=False, encoder_states=None, want_encoder=False):
        """mrm = (rows int32 [n], soft targets fp32 [n, C]); attr = (rows, labels int64); rel = (obj rows, subj rows,
        labels).  Returns (losses fp32 [5] = total, lm, mrm, attribute, relation; logits or None) -- plus the encoder states
        with want_encoder.  `encoder_states` [B,S,D] skips the encoder (reference src/model/model.py:225-242 passes
        `encoder_outputs` through to self.model)."""
        with torch.cuda.device(self.device):
            b, keep, (B, S, T, ntot) = self._batch(input_ids, image_features if encoder_states is None else [], attention_mask,
                                                   decoder_input_ids, decoder_attention_mask, labels)
            dev = self.device

            def i32(t):
                return t.to(device=dev, dtype=torch.int32).contiguous()

            ex = KmbPretrain(lm_factor=factors[0], mrm_factor=factors[1], attr_factor=factors[2], rel_factor=factors[3])
            nmax = 0
            if mrm is not None and mrm[0].numel() > 0:
                rows, tgt = i32(mrm[0]), mrm[1].to(device=dev, dtype=torch.float32).contiguous()
                ex.n_mrm, ex.mrm_rows, ex.mrm_targets = rows.numel(), ptr(rows), ptr(tgt)
                keep += [rows, tgt]
                nmax = max(nmax, rows.numel())
            if attr is not None and attr[0].numel() > 0:
                rows, lab = i32(attr[0]), attr[1].to(device=dev, dtype=torch.int64).contiguous()
                ex.n_attr, ex.attr_rows, ex.attr_labels = rows.numel(), ptr(rows), ptr(lab)
                keep += [rows, lab]
                nmax = max(nmax, rows.numel())
            if rel is not None and rel[0].numel() > 0:
                ro, rs, lab = i32(rel[0]), i32(rel[1]), rel[2].to(device=dev, dtype=torch.int64).contiguous()
                ex.n_rel, ex.rel_obj_rows, ex.rel_subj_rows, ex.rel_labels = ro.numel(), ptr(ro), ptr(rs), ptr(lab)
                keep += [ro, rs, lab]
                nmax = max(nmax, ro.numel())
            check(self.lib.kmb_reserve_head_rows(self.h, max(nmax, 8)))
            self._ensure_ws(self.lib.kmb_workspace_bytes(self.h, B, S, T, ntot))
            self._poison()
            losses = torch.zeros(5, dtype=torch.float32, device=dev)
            ex.losses_out = ptr(losses)
            logits = torch.empty((B * T, self.logits_ld), dtype=torch.float32, device=dev) if want_logits else None
            self.fwd_serial += 1
            self._last_bt = (B, T)
            self._last_S = S
            D = int(self.config.d_model)
            opts = KmbForwardOpts()
            if encoder_states is not None:
                if tuple(encoder_states.shape) != (B, S, D):
                    raise ValueError("encoder_outputs[0] must be [batch, src_len, d_model] = %s, got %s"
                                     % ((B, S, D), tuple(encoder_states.shape)))
                encoder_states = encoder_states.to(device=dev, dtype=self.act_dtype).contiguous()
                opts.encoder_states = ptr(encoder_states)
                keep.append(encoder_states)
            enc = torch.empty((B, S, D), dtype=self.act_dtype, device=dev) if want_encoder else None
            check(self.lib.kmb_forward_pretrain_ex(self.h, C.byref(b), C.byref(ex), C.byref(opts), 1 if train else 0,
                                                   1 if need_grad else 0, ptr(logits), ptr(enc), _stream()))
            self._keep = keep
            if logits is not None:
                logits = logits.view(B, T, self.logits_ld)[:, :, : int(self.config.vocab_size)]
            return (losses, logits, enc) if want_encoder else (losses, logits)

    def check_inputs(self):
        """Raises if the device-side validation of the last forward flagged the batch (syncs)."""
        st = C.c_int32(0)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_read_status(self.h, C.byref(st), _stream()))
        self._raise_on_status(st.value)

    @staticmethod
    def _raise_on_status(st):
        if st & 1:
            raise RuntimeError("number of <img_feat>/<cls> ids differs from the number of region features "
                               "(reference src/model/modules.py:98-100 would raise a shape mismatch)")
        if st & 2:
            raise RuntimeError("a label is neither -100 nor inside [0, number of classes) "
                               "(the reference's CrossEntropyLoss raises on such a target, src/model/model.py:400-402)")
        if st & 8:
            raise RuntimeError("a group barrier of the resident decoder-layers kernel gave up (csrc/decode.hip: its workgroups were "
                               "not all co-resident -- another kernel held CUs?); the generated tokens of this call are invalid. "
                               "KMB_GEN_FUSED=1 selects the six-launches-per-layer blocks")

    def read_status(self):
        """The device status word of the last forward / generation (syncs); bits: _raise_on_status."""
        st = C.c_int32(0)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_read_status(self.h, C.byref(st), _stream()))
        return int(st.value)

    def check_inputs_begin(self):
        """check_inputs without the wait: the status word is copied to page-locked memory in stream order; the caller
        calls check_inputs_end() after its next synchronisation of the stream."""
        st = self.pinned((1,), torch.int32)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_read_status_async(self.h, ptr(st), _stream()))
            ev = torch.cuda.Event()
            ev.record()
        self._status_pending = (st, ev)

    def check_inputs_end(self):
        pend = self.__dict__.pop("_status_pending", None)
        if pend is None:
            return
        st, ev = pend
        ev.synchronize()
        self._raise_on_status(int(st[0]))

    def backward(self, loss_scale=1.0):
        """loss_scale: a Python float, or a 1-element fp32 device tensor (autograd's upstream gradient: no host sync)."""
        with torch.cuda.device(self.device):
            if torch.is_tensor(loss_scale):
                sc = loss_scale.detach().to(device=self.device, dtype=torch.float32).reshape(1).contiguous()
                check(self.lib.kmb_backward_dev(self.h, ptr(sc), _stream()))
                self._keep_scale = sc
            else:
                check(self.lib.kmb_backward(self.h, C.c_float(loss_scale), _stream()))

    def adamw_step(self, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True, grad_scale=1.0,
                   offset=0, count=None, bump=True):
        if bump:
            self.step_count += 1
        self.fwd_serial += 1   # the weights move: logits of an earlier forward can no longer be reproduced
        hp = KmbAdamW(lr=lr, beta1=betas[0], beta2=betas[1], eps=eps, weight_decay=weight_decay,
                      step=self.step_count, correct_bias=1 if correct_bias else 0, grad_scale=grad_scale)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_adamw_step(self.h, C.byref(hp), offset, self.n - offset if count is None else count,
                                          _stream()))

    def adamw_step_overlapped(self, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True, grad_scale=1.0,
                              offset=0, count=None):
        """The same update issued per gradient bucket on a second stream, each launch behind that bucket's completion
        event of the backward pass that is still executing on the GPU (the host enqueues a step in ~2 ms, the GPU needs
        ~20 ms for backward): the HBM-bound update of the decoder and encoder layers runs beside the compute-bound rest of
        backward; only the tied matrix (last bucket) is left for after it.  Parameters of a bucket are not read again by
        the backward pass once its event is recorded.  Returns False (nothing issued) when a piece is not 8-aligned or a
        data-parallel reducer owns the gradients (they are final only after its all-reduce)."""
        if not getattr(self, "adamw_overlap_ok", True):
            return False
        count = self.n - offset if count is None else count
        if getattr(self, "_bucket_list", None) is None:
            self._bucket_list = self.buckets()
            self._opt_stream = torch.cuda.Stream(device=self.device)
        end = offset + count
        pieces = []
        for i, (boff, bcnt) in enumerate(self._bucket_list):
            lo, hi = max(offset, boff), min(end, boff + bcnt)
            if lo < hi:
                if lo & 7:
                    return False
                pieces.append((i, lo, hi - lo))
        if sum(c for _, _, c in pieces) != count:
            return False
        hp = KmbAdamW(lr=lr, beta1=betas[0], beta2=betas[1], eps=eps, weight_decay=weight_decay,
                      step=self.step_count, correct_bias=1 if correct_bias else 0, grad_scale=grad_scale)
        self.fwd_serial += 1
        with torch.cuda.device(self.device):
            main, side = torch.cuda.current_stream(), self._opt_stream
            for i, lo, cnt in pieces:
                self.stream_wait_bucket(i, side)
                check(self.lib.kmb_adamw_step(self.h, C.byref(hp), lo, cnt, C.c_void_p(side.cuda_stream)))
            main.wait_stream(side)
        return True

    # ---- data-parallel buckets --------------------------------------------------------------
    def buckets(self):
        out = []
        off, cnt = C.c_int64(), C.c_int64()
        for i in range(self.lib.kmb_bucket_count(self.h)):
            check(self.lib.kmb_bucket_range(self.h, i, C.byref(off), C.byref(cnt)))
            out.append((off.value, cnt.value))
        return out

    def stream_wait_bucket(self, i, stream):
        check(self.lib.kmb_stream_wait_bucket(self.h, i, C.c_void_p(stream.cuda_stream)))

    # ---- native RCCL gradient exchange (kmb_comm_*, include/kmbart.h) ------------------------
    def comm_init(self, process_group=None):
        """Creates the library's own RCCL communicator over the ranks of `process_group` (torch.distributed is only the
        bootstrap channel for the 128-byte id, as torchrun's rendezvous is for init_process_group in the reference,
        src/utils.py:9-17).  Returns (rank, world)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
        box = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            check(self.lib.kmb_comm_unique_id(buf))
            box[0] = bytes(buf.raw)
        src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
        dist.broadcast_object_list(box, src=src, group=process_group)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_comm_init(self.h, rank, world, C.c_char_p(box[0])))
        self.comm_world = world
        return rank, world

    def comm_destroy(self):
        """Drops the library's communicator (the data-parallel wrapper's fallback to torch.distributed when the native
        bootstrap failed on another rank).  If a reduce-scatter exchange left exp_avg / exp_avg_sq sharded, they are gathered first
        (a COLLECTIVE: teardown runs on every rank; after a failed bootstrap nothing is sharded) -- kmb_comm_destroy clears the flag."""
        with torch.cuda.device(self.device):
            if self.moments_sharded:
                check(self.lib.kmb_comm_gather_moments(self.h, _stream()))
                torch.cuda.synchronize(self.device)
            check(self.lib.kmb_comm_destroy(self.h))
        self.comm_world = 0

    def comm_broadcast_params(self, root=0):
        with torch.cuda.device(self.device):
            check(self.lib.kmb_comm_broadcast_params(self.h, int(root), _stream()))

    def allreduce_grads(self, algo=0, adamw=None, max_piece_elems=0, after_compute=False):
        """One call = the whole gradient exchange of a step on the library's communication stream (mean over ranks per
        bucket behind its completion event, optionally each piece's fused AdamW behind it).  adamw: KmbAdamW or None."""
        from ._lib import KmbAllreduceOpts
        o = KmbAllreduceOpts(algo=int(algo), after_compute=1 if after_compute else 0, max_piece_elems=int(max_piece_elems),
                             adamw=C.pointer(adamw) if adamw is not None else None)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_allreduce_grads(self.h, C.byref(o), _stream()))
        if adamw is not None:
            self.fwd_serial += 1   # the weights move

    def comm_wait(self):
        with torch.cuda.device(self.device):
            check(self.lib.kmb_comm_wait(self.h, _stream()))

    def comm_gather_moments(self):
        """COLLECTIVE (every rank must call it): all-gathers the shards of exp_avg / exp_avg_sq that an algo-1 exchange with a
        fused optimizer leaves current on their owning rank only."""
        with torch.cuda.device(self.device):
            check(self.lib.kmb_comm_gather_moments(self.h, _stream()))

    @property
    def moments_sharded(self):
        """True after a reduce-scatter exchange with a fused optimizer on more than one rank, until comm_gather_moments()."""
        return bool(self.lib.kmb_comm_moments_sharded(self.h))

    def comm_plan(self, world, rank, max_piece_elems=0):
        """The exchange's partition as rank `rank` of `world` sees it (kmb_comm_plan: a pure host function, no communicator
        needed): list of dicts bucket / offset / count / shard / mine / repad_piece / repad_shard."""
        from ._lib import KmbCommPiece
        out = []
        for i in range(int(self.lib.kmb_comm_pieces(self.h, int(max_piece_elems)))):
            pc = KmbCommPiece()
            check(self.lib.kmb_comm_plan(self.h, int(world), int(rank), int(max_piece_elems), i, C.byref(pc)))
            out.append({k: int(getattr(pc, k)) for k in ("bucket", "offset", "count", "shard", "mine", "repad_piece", "repad_shard")})
        return out

    # ---- generation -------------------------------------------------------------------------
    def gen_begin(self, input_ids, image_features, attention_mask, num_beams, max_length):
        with torch.cuda.device(self.device):
            b, keep, (B, S, _, ntot) = self._batch(input_ids, image_features, attention_mask, None, None, None)
            self._ensure_ws(self.lib.kmb_gen_workspace_bytes(self.h, B, S, num_beams, max_length, ntot))
            self._poison()
            self.fwd_serial += 1
            check(self.lib.kmb_gen_begin(self.h, C.byref(b), num_beams, max_length, _stream()))
            self._keep = keep
            self._gen_rows = B * num_beams
            self._gen_enc_shape = (B, S)
            self._gen_max_length = int(max_length)
            self._gen_logits = torch.empty((self._gen_rows, self.logits_ld), dtype=torch.float32, device=self.device)

    def gen_encoder_states(self):
        """[B, S, d] bf16: the encoder output of the active gen_begin (a copy)."""
        B, S = self._gen_enc_shape
        out = torch.empty((B, S, int(self.config.d_model)), dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_gen_encoder_states(self.h, ptr(out), _stream()))
        return out

    def gen_step(self, tokens, step, want_logits=True):
        """tokens int64 [B*num_beams] (device) at 0-based position `step` -> fp32 logits [R, V] (padded view).
        want_logits=False: the decoder layers run (the KV cache gets position `step`) but the vocabulary projection is
        skipped; the returned buffer then holds stale values (for steps whose token is forced)."""
        with torch.cuda.device(self.device):
            tokens = tokens.to(device=self.device, dtype=torch.int64).contiguous()
            check(self.lib.kmb_gen_step(self.h, ptr(tokens), int(step), ptr(self._gen_logits) if want_logits else None,
                                        _stream()))
            self._keep_tok = tokens
        return self._gen_logits

    def gen_last_hidden(self):
        """[rows, d] bf16: the final decoder states of the last gen_step (kmb_gen_last_hidden)."""
        out = torch.empty((self._gen_rows, int(self.config.d_model)), dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_gen_last_hidden(self.h, ptr(out), _stream()))
        return out

    def gen_reorder(self, beam_idx, step):
        with torch.cuda.device(self.device):
            beam_idx = beam_idx.to(device=self.device, dtype=torch.int32).contiguous()
            check(self.lib.kmb_gen_reorder(self.h, ptr(beam_idx), int(step), _stream()))
            self._keep_idx = beam_idx

    def beam_candidates(self, logits, num_beams, k, add=None, force_token=-1, ban_token=-1):
        """log_softmax(+beam score) top-k per beam row, merged per batch item: int32 [B, k, 2] on the device,
        [..., 0] = fp32 score bits, [..., 1] = beam * V + token (one small D2H copy per decode step)."""
        R = logits.shape[0]
        val, idx = self.logsoftmax_topk(logits, k, add=add, force_token=force_token, ban_token=ban_token)
        out = torch.empty((R // num_beams, k, 2), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_beam_merge(ptr(val), ptr(idx), R // num_beams, int(num_beams), int(k),
                                          int(self.config.vocab_size), ptr(out), _stream()))
        return out

    def pinned(self, shape, dtype):
        """A cached page-locked host tensor (asynchronous device-to-host staging)."""
        key = (tuple(shape), dtype)
        cache = self.__dict__.setdefault("_pinned", {})
        if key not in cache:
            cache[key] = torch.empty(shape, dtype=dtype, pin_memory=True)
        return cache[key]

    def beam_step(self, logits, num_beams, k, add, force_token=-1, ban_token=-1, eos_token=-1, cand_out=None, reorder_step=-1):
        """beam_candidates plus the next step's beams chosen on the device (kmb_beam_merge_select): returns
        (cand int32 [B, k, 2], next_scores fp32 [R], next_tokens int64 [R], next_beam_idx int32 [R]); nothing is
        copied to the host.  `add` may be one of the returned next_scores (stream order makes that safe).
        reorder_step >= 0: the call also reorders the generation caches by next_beam_idx (gen_reorder(next_beam_idx,
        reorder_step), folded into the beam step's launch when `logits` are gen_step's)."""
        R = logits.shape[0]
        B = R // num_beams
        # cand_out: a page-locked host tensor [B, k, 2] int32 the kernel writes directly (device-visible host memory: no copy
        # launch between two decode steps); the caller reads it after an event recorded behind this call
        cand = cand_out if cand_out is not None else torch.empty((B, k, 2), dtype=torch.int32, device=self.device)
        nscore = torch.empty((R,), dtype=torch.float32, device=self.device)
        ntok = torch.empty((R,), dtype=torch.int64, device=self.device)
        nidx = torch.empty((R,), dtype=torch.int32, device=self.device)
        if k <= 16 and num_beams <= 16 and num_beams * k <= 256 and logits.stride(0) % 4 == 0:
            # kmb_beam_step: the rows' top-k lists never leave the workgroup that merges them (one launch less per decode step)
            scr = self._topk_scratch_for(R)
            if logits is self.__dict__.get("_gen_logits"):
                # the decode loop: kmb_gen_beam_step selects from the statistics the step's vocabulary projection left (one launch,
                # no second pass over the logits) when there are any, and is kmb_beam_step otherwise
                with torch.cuda.device(self.device):
                    check(self.lib.kmb_gen_beam_step(self.h, ptr(logits), logits.stride(0), int(num_beams), ptr(add), int(force_token),
                                                     int(ban_token), int(k), ptr(cand), int(eos_token), ptr(nscore), ptr(ntok),
                                                     ptr(nidx), ptr(scr), scr.numel(), int(reorder_step), _stream()))
                    self._keep_idx = nidx
                return cand, nscore, ntok, nidx
            with torch.cuda.device(self.device):
                check(self.lib.kmb_beam_step(ptr(logits), logits.stride(0), int(self.config.vocab_size), B, int(num_beams), ptr(add),
                                             int(force_token), int(ban_token), int(k), ptr(cand), int(eos_token), ptr(nscore),
                                             ptr(ntok), ptr(nidx), ptr(scr), scr.numel(), _stream()))
            if reorder_step >= 0:
                self.gen_reorder(nidx, reorder_step)
            return cand, nscore, ntok, nidx
        val, idx = self.logsoftmax_topk(logits, k, add=add, force_token=force_token, ban_token=ban_token)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_beam_merge_select(ptr(val), ptr(idx), B, int(num_beams), int(k), int(self.config.vocab_size),
                                                 ptr(cand), int(eos_token), ptr(nscore), ptr(ntok), ptr(nidx), _stream()))
        if reorder_step >= 0:
            self.gen_reorder(nidx, reorder_step)
        return cand, nscore, ntok, nidx

    def _topk_scratch_for(self, R):
        nscr = int(self.lib.kmb_logsoftmax_topk_scratch(R))
        scr = self.__dict__.get("_topk_scratch")
        if scr is None or scr.numel() < nscr:
            scr = self._topk_scratch = torch.empty(nscr, dtype=torch.float32, device=self.device)
        return scr

    def logsoftmax_topk(self, logits, k, add=None, force_token=-1, ban_token=-1):
        R = logits.shape[0]
        val = torch.empty((R, k), dtype=torch.float32, device=self.device)
        idx = torch.empty((R, k), dtype=torch.int32, device=self.device)
        scr = self._topk_scratch_for(R)
        with torch.cuda.device(self.device):
            check(self.lib.kmb_logsoftmax_topk_ws(ptr(logits), logits.stride(0), int(self.config.vocab_size), R,
                                                  ptr(add), int(force_token), int(ban_token), int(k), ptr(val), ptr(idx),
                                                  ptr(scr), scr.numel(), _stream()))
        return val, idx
