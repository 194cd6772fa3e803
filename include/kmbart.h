/* libkmbart_hip.so -- C ABI of the MI355X-native KM-BART hot path.
 *
 * The reference (fomalhautb/KM-BART) has no native interface: its hot path is reached through the
 * Python API of src.model / src.training / src.generation.  This header is
 * the boundary a maintainer binds instead (ctypes stub in INTEGRATION.md).  Every entry point
 * cites the reference code whose arithmetic it replaces (paths into the reference checkout).
 *
 * Conventions: plain pointers and sizes only; all data pointers are DEVICE pointers unless the
 * name ends in _host; `stream` is a hipStream_t passed as void*; every function returns 0 on
 * success, non-zero on failure with the message available from kmb_last_error().  Nothing here
 * allocates device memory: the host binds arenas / workspace it owns (torch tensors in the
 * Python host code).  A handle is not thread-safe (one host thread per process per GPU, as
 * the reference's mp.spawn layout, vcg_train.py:350-355).
 */
#ifndef KMBART_H
#define KMBART_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t kmb_bf16;            /* raw bfloat16 bits */
typedef struct kmb_handle kmb_handle;

/* src/model/config.py:4-92 + config/vcg_base.json */
typedef struct kmb_config {
  int32_t vocab_size, d_model;
  int32_t encoder_layers, decoder_layers;
  int32_t encoder_attention_heads, decoder_attention_heads;
  int32_t encoder_ffn_dim, decoder_ffn_dim;
  int32_t max_position_embeddings, extra_pos_embeddings;
  int32_t image_feature_size;
  int32_t pad_token_id, bos_token_id, eos_token_id, img_feat_id, cls_token_id;
  int32_t scale_embedding;
  float dropout, attention_dropout, activation_dropout;
  float layer_norm_eps;
  /* pre-training heads (src/model/model.py:133-158); 0 = head absent (MultiModalBartForConditionalGeneration) */
  int32_t num_labels, num_attributes, num_relations;
} kmb_config;

/* one training / scoring batch; layout = what the reference Collator emits
 * (src/data/collation.py:68-213), region features packed row-wise with CSR offsets */
typedef struct kmb_batch {
  int32_t B, S, T;                     /* batch, encoder length, decoder length */
  const int64_t* input_ids;            /* [B,S] */
  const int64_t* attention_mask;       /* [B,S] 1 = keep, or NULL (all ones) */
  const float* image_features;         /* [Ntot, image_feature_size] fp32, rows of sample i at feat_offsets[i] */
  const int32_t* feat_offsets;         /* [B+1] device */
  int32_t n_features;                  /* Ntot (host copy of feat_offsets[B]) */
  const int64_t* decoder_input_ids;    /* [B,T] */
  const int64_t* decoder_attention_mask; /* [B,T] or NULL */
  const int64_t* labels;               /* [B,T], -100 = ignore, or NULL */
} kmb_batch;

/* ---- GEMM with fused epilogue (csrc/gemm.hip) --------------------------------------------- */
typedef struct KmbGemm {
  const kmb_bf16* A; const kmb_bf16* B;
  int32_t lda, ldb;
  int32_t a_kc, b_kc;
  int32_t M, N, K;
  const float* bias;
  float col_scale; int32_t col_scale_n;
  int32_t act;                    /* 0 none, 1 GeLU (erf form), 2 multiply by aux (the stored GeLU'), 3 tanh, 4 multiply by 1 - aux^2,
                                   * 5 exp(v - row_shift[row]) with per-row sums (the tied head's cross-entropy: fields at the end) */
  kmb_bf16* preact; int32_t ld_preact;   /* act 1, optional: receives GeLU'(pre-activation) -- what the backward pass needs */
  const kmb_bf16* aux; int32_t ld_aux;   /* act 2: the tensor act 1 stored; act 4: tanh output */
  uint32_t drop_thr16; uint32_t drop_seed; float drop_scale;
  const kmb_bf16* residual; int32_t ld_res;
  kmb_bf16* out_bf16; int32_t ld_out_bf16;
  float* out_f32; int32_t ld_out_f32; float beta;
  int32_t split_k; float* slab;   /* split_k > 1: slice s writes raw fp32 accumulators to slab[s][M][N]; no epilogue */
  float* colsum;                  /* optional [ceil(M/64)][N]: per-64-row-block column sums of the stored values */
  int32_t tile_order;             /* set by the launcher (results never depend on it): bit 0 = contiguous tile range per
                                   * XCD; bit 1 = L2 prefetch of the activation panel (persistent kernels); bit 2 = split-K
                                   * (tile, slice) pairs enumerated slice-major: an XCD works on one or two K slices */
  /* act 5 (reference src/model/model.py:397-402 without a pass over the logits): the output is exp(v - row_shift[row]);
   * row_sums[row * row_sums_ld + col / 64] receives, for every 64-column block, the fp32 sum of the row's values in it (a
   * wider wave block puts its whole sum into its first slot and zeros into the others: every slot is written exactly
   * once), pick_out[row] the value v - row_shift[row] at column pick_col[row] (rows whose pick_col is outside [0, N)
   * are not written).  Persistent 256- / 128-column variants only, M and N multiples of 256, bias required. */
  const float* row_shift; float* row_sums; int32_t row_sums_ld; const int64_t* pick_col; float* pick_out;
} KmbGemm;

/* ---- fused attention (csrc/attention.hip) -------------------------------------------------- */
typedef struct KmbAttn {
  const kmb_bf16* Q; const kmb_bf16* K; const kmb_bf16* V;
  int32_t ldq, ldk, ldv;
  int32_t B, H, Tq, Tk;
  const int64_t* key_mask;
  int32_t causal;
  kmb_bf16* O; int32_t ldo;
  float* lse;
  const kmb_bf16* dO; int32_t lddo;
  kmb_bf16* dQ; kmb_bf16* dK; kmb_bf16* dV; int32_t lddq, lddk, lddv;
  float dq_scale;
  /* optional bias-gradient partials (backward): per-batch-item column sums of dQ / dK / dV, row stride ld_colsum */
  float* dq_colsum; float* dk_colsum; float* dv_colsum; int32_t ld_colsum;
} KmbAttn;

typedef struct KmbAttnDecode {
  const kmb_bf16* Q; int32_t ldq;
  const kmb_bf16* Kc; const kmb_bf16* Vc;
  int32_t Tmax; int32_t ldc;      /* cache element (row,t,h,e) at X[(row*Tmax + t)*ldc + h*64 + e] */
  const int32_t* kv_row;
  const int64_t* key_mask; int32_t mask_ld;
  const int32_t* mask_row;
  int32_t R, H, Tk;
  kmb_bf16* O; int32_t ldo;
  /* self-attention of a decode step: key / value row Tk-1 is NOT in the cache yet -- it is read from new_k / new_v
   * (row r at new_k + r*ld_new + h*64) and written into caches Kw / Vw at position Tk-1 by the same launch (NULL: plain
   * attention over the cache) */
  const kmb_bf16* new_k; const kmb_bf16* new_v; int32_t ld_new;
  kmb_bf16* Kw; kmb_bf16* Vw;
  /* optional history index of the self-attention cache ([R, Tmax] int32; NULL: a row's history lives in its own cache row): position
   * t < Tk-1 of row r is read from cache row hist[r*Tmax + t]; the launch records hist[r*Tmax + Tk-1] = r for the row it appends.  A beam
   * reorder then permutes these small rows instead of copying the caches (src/model/mixins.py:419-434 _reorder_cache). */
  int32_t* hist;
} KmbAttnDecode;

/* One fused block of a KV-cached decode step (csrc/decode.hip): [LayerNorm ->] projection of R rows [-> attention].
 * Replaces the GEMM + attention + LayerNorm launches of one sub-layer of the reference's DecoderLayer under use_cache
 * (transformers 3.0.2 modeling_bart.py:386-466, reached from src/model/mixins.py:386-434).
 *   kind 0: out[R, N] = act(LN?(in) W^T + bias) + residual                         (N % 64 == 0)
 *   kind 1: self-attention of head h = 0..H-1: W = [q | k | v] rows (N = 3*H*64); the new key / value row is written to
 *           the cache at position Tk-1 and attended to together with cache rows 0 .. Tk-2; out[R, H*64]
 *   kind 2: cross-attention: W = q rows (N = H*64); keys / values Kc / Vc of batch item kv_row[row], masked by key_mask
 * K % 768 == 0, K <= 3072 (kinds 1, 2: K = 768).  gamma != NULL: `in` holds pre-LayerNorm sums and the LayerNorm is
 * applied on the way in; ln_out (optional, row stride K) receives the normalised rows once. */
typedef struct KmbDecodeBlock {
  int32_t kind;
  const kmb_bf16* in; int32_t ld_in;
  const float* gamma; const float* beta; float eps;
  kmb_bf16* ln_out;
  const kmb_bf16* W; const float* bias;     /* W: the [N, K] weight in fragment order (kmb_op_decode_pack), bias [N] */
  int32_t R, K, N;
  int32_t act;                              /* kind 0: 1 = GeLU */
  const kmb_bf16* residual; int32_t ld_res; /* kind 0 */
  kmb_bf16* out; int32_t ld_out;
  int32_t H; float q_scale;                 /* kinds 1, 2: (q + bias) * q_scale */
  kmb_bf16* Kc; kmb_bf16* Vc;               /* cache element (row,t,h,e) at X[(row*Tmax + t)*ldc + h*64 + e] */
  int32_t Tmax, ldc, Tk;
  const int32_t* kv_row;                    /* kind 2: cache row (batch item) of every row */
  const int64_t* key_mask; int32_t mask_ld; /* kind 2: key t of cache row c is masked when key_mask[c*mask_ld + t] == 0 */
  int32_t* hist;                            /* kind 1, optional: history index of the cache (see KmbAttnDecode.hist) */
  int32_t kv_group;                         /* kind 2, optional: a promise that kv_row[r] == r / kv_group (beam rows of a
                                             * batch item are consecutive): lets a workgroup stage each item's keys /
                                             * values once for all of its rows.  0: no promise. */
} KmbDecodeBlock;

typedef struct KmbDrop { uint32_t thr16; uint32_t seed; float scale; } KmbDrop;
typedef struct KmbAdamW { double lr, beta1, beta2, eps, weight_decay; int32_t step; int32_t correct_bias; float grad_scale; } KmbAdamW;

const char* kmb_last_error(void);
int kmb_version(void);

/* ================= model handle ================= */
/* MultiModalBartForConditionalGeneration(config): src/model/model.py:317-323 */
int kmb_create(const kmb_config* cfg, kmb_handle** out);
void kmb_destroy(kmb_handle* h);

/* parameter census (state-dict names are the reference's: HF BART keys +
 * model.encoder.embed_images.linear.{weight,bias}); offsets are in ELEMENTS into the flat arenas */
int kmb_param_count(const kmb_handle* h);
int kmb_param_info(const kmb_handle* h, int idx, const char** name, int64_t* offset, int32_t* rows, int32_t* cols);
int64_t kmb_arena_elems(const kmb_handle* h);        /* fp32 arenas: params / grads / exp_avg / exp_avg_sq */
int64_t kmb_bf16_arena_elems(const kmb_handle* h);   /* bf16 mirror (+ padded tied matrix + padded image weight) */
int kmb_bind_arenas(kmb_handle* h, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                    kmb_bf16* params_bf16, float* final_logits_bias);
int64_t kmb_workspace_bytes(const kmb_handle* h, int B, int S, int T, int n_features);
int kmb_bind_workspace(kmb_handle* h, void* ws, int64_t bytes);
/* refresh the bf16 mirror from the fp32 master parameters (after init / load_state_dict) */
int kmb_sync_params(kmb_handle* h, void* stream);
int kmb_set_seed(kmb_handle* h, uint64_t seed);

/* gradient buckets for data-parallel overlap (DDP reducer, vcg_train.py:98): bucket i is complete
 * on the compute stream once kmb_backward has passed its event */
int kmb_bucket_count(const kmb_handle* h);
int kmb_bucket_range(const kmb_handle* h, int i, int64_t* offset, int64_t* count);
int kmb_stream_wait_bucket(kmb_handle* h, int i, void* stream);

/* ================= training step ================= */
/* MultiModalBartForConditionalGeneration.forward, src/model/model.py:325-405
 * (encoder src/model/modules.py:104-165, ImageEmbedding :24-41, _embed_multi_modal :89-102).
 *   train       : dropout active (model.train())
 *   need_grad   : also produce what kmb_backward needs (and the tied-head gradients)
 *   loss_out    : device float[1] (mean CE over labels != -100), may be NULL when labels == NULL
 *   logits_out  : device float [B*T, kmb_logits_ld()] or NULL; columns >= vocab_size are padding
 *   enc_out     : device bf16 [B*S, d_model] or NULL (encoder_last_hidden_state copy) */
int kmb_logits_ld(const kmb_handle* h);
int kmb_forward(kmb_handle* h, const kmb_batch* batch, int train, int need_grad, float* loss_out,
                float* logits_out, kmb_bf16* enc_out, void* stream);
/* Optional inputs / outputs of the forward that the bare MultiModalBartModel and the `encoder_outputs` keyword need
 * (src/model/model.py:39-103): */
typedef struct kmb_forward_opts {
  const kmb_bf16* encoder_states;   /* [B*S, d_model] or NULL: skip the encoder and use these states (model.py:76-83) */
  kmb_bf16* decoder_states_out;     /* [B*T, d_model] or NULL: copy of the last decoder hidden states (model.py:87-103) */
  int32_t skip_head;                /* 1: stop after the decoder (no logits, no loss): MultiModalBartModel.forward */
} kmb_forward_opts;
/* kmb_forward with those options.  With encoder_states AND need_grad the following kmb_backward stops at the given states:
 * the encoder's parameter gradients are zero, dL / d states comes from kmb_encoder_states_grad (the reference's
 * `forward(encoder_outputs=...)` with a tensor that requires grad, src/model/model.py:76-83).  In the fp32 validation mode the
 * kmb_bf16 pointers of kmb_forward / kmb_forward_ex carry floats (kmb_act_bytes() == 4). */
int kmb_forward_ex(kmb_handle* h, const kmb_batch* batch, const kmb_forward_opts* opts, int train, int need_grad,
                   float* loss_out, float* logits_out, kmb_bf16* enc_out, void* stream);
/* logits [B*T, kmb_logits_ld()] of the decoder states the LAST forward left in the workspace: `outputs[1]` of a training
 * forward (src/model/model.py:397-405) as one extra head GEMM instead of 1.6 GB written every step */
int kmb_last_logits(kmb_handle* h, float* logits_out, void* stream);
/* `output_hidden_states` / `output_attentions` of the forward still in the workspace (reference src/model/modules.py:143-165,
 * transformers 3.0.2 BartDecoder): hidden state `index` of the encoder (which = 0: 0 .. encoder_layers, 0 = the embedding
 * output, l + 1 = the output of layer l) or of the decoder (which = 1: 0 .. decoder_layers) as [rows, d_model] activations,
 * and the self-attention probabilities of layer `layer` as fp32 [B, H, T, T], recomputed from the saved q | k and the rows'
 * log-sum-exp (the fused attention kernels never store them).  bf16 product mode only. */
int kmb_hidden_state(kmb_handle* h, int which, int index, kmb_bf16* out, void* stream);
int kmb_attention_probs(kmb_handle* h, int which, int layer, float* out, void* stream);
/* dL / d(encoder output) of the last kmb_backward as bf16 [B*S, d_model] (what autograd hands to `encoder_outputs[0].grad`) */
int kmb_encoder_states_grad(kmb_handle* h, kmb_bf16* out, void* stream);
/* fp32 VALIDATION mode (1) / bf16 product mode (0, default).  Mode 1 keeps every activation in float and runs the
 * eval-mode forward on exact-fp32 kernels (csrc/fp32_validate.hip) against the fp32 master weights: parity evidence for
 * north_star's "logits within 1e-3 of the fp32 reference path", never the measured path.  Training, backward and
 * generation are refused in mode 1.  Rebind the workspace after switching (kmb_workspace_bytes doubles). */
int kmb_set_precision(kmb_handle* h, int fp32);
int kmb_act_bytes(const kmb_handle* h);
/* loss.backward() (src/training.py:138,142); loss_scale multiplies every gradient */
int kmb_backward(kmb_handle* h, float loss_scale, void* stream);
/* the same with the scale as a DEVICE scalar: autograd hands `loss.backward()` its upstream gradient (ones, or the
 * GradScaler's scale, src/training.py:137-142) as a device tensor; reading it on the device keeps the API path free of a
 * host synchronisation.  A scale of exactly 1 costs three early-exit launches. */
int kmb_backward_dev(kmb_handle* h, const float* loss_scale_dev, void* stream);
/* transformers.AdamW.step (vcg_train.py:100, src/training.py:139,143) over [offset, offset+count) */
int kmb_adamw_step(kmb_handle* h, const KmbAdamW* hp, int64_t offset, int64_t count, void* stream);
/* status word written by device-side input validation (bit 0: #<img_feat> ids != #region rows) */
int kmb_read_status(kmb_handle* h, int32_t* status_host, void* stream);
/* The same without the synchronisation: the status word is copied to `status_host` (page-locked memory) in stream order; the
 * caller reads it after a later synchronisation of that stream (generation: at the end of generate(), so that the decode
 * steps are enqueued while the encoder still runs). */
int kmb_read_status_async(kmb_handle* h, int32_t* status_host, void* stream);

/* ---- multi-task pre-training (MultiModalBartForPreTraining.forward, src/model/model.py:162-309) ----
 * rows index the flattened decoder positions [B*T]; every pointer is a device pointer; n_* may be 0 */
typedef struct kmb_pretrain {
  int32_t n_mrm; const int32_t* mrm_rows; const float* mrm_targets;     /* soft labels [n_mrm, num_labels] (model.py:248-257) */
  int32_t n_attr; const int32_t* attr_rows; const int64_t* attr_labels; /* (model.py:259-268) */
  int32_t n_rel; const int32_t* rel_obj_rows; const int32_t* rel_subj_rows; const int64_t* rel_labels; /* (:270-289) */
  float lm_factor, mrm_factor, attr_factor, rel_factor;                 /* config.*_loss_factor (:304-307) */
  float* losses_out;                                                    /* device float[5]: loss, lm, mrm, attribute, relation */
} kmb_pretrain;
/* room for up to n gathered head rows in the workspace (call before kmb_workspace_bytes) */
int kmb_reserve_head_rows(kmb_handle* h, int n);
/* kmb_forward plus the three classification heads; `labels` must already carry -100 at <cls> positions (:297-298) */
int kmb_forward_pretrain(kmb_handle* h, const kmb_batch* batch, const kmb_pretrain* extra, int train, int need_grad,
                         float* logits_out, kmb_bf16* enc_out, void* stream);
/* ... with a kmb_forward_opts record -- encoder_states: the reference passes `encoder_outputs` through to self.model,
 * src/model/model.py:225-242; decoder_states_out; skip_head is refused */
int kmb_forward_pretrain_ex(kmb_handle* h, const kmb_batch* batch, const kmb_pretrain* extra, const kmb_forward_opts* opts,
                            int train, int need_grad, float* logits_out, kmb_bf16* enc_out, void* stream);

/* ================= generation ================= */
/* encoder once (src/model/mixins.py:281-283) + cross-attention K/V of every decoder layer.  Asynchronous: everything is
   enqueued on `stream` and the call returns without synchronising (the batch's device buffers must stay valid until the
   generation's last kmb_gen_step has run). */
int kmb_gen_begin(kmb_handle* h, const kmb_batch* batch, int num_beams, int max_length, void* stream);
/* encoder output [B*S, d_model] (bf16) of the kmb_gen_begin still active: the `encoder_outputs` element of the cached
 * forward's return tuple (src/model/model.py:384-397 returns decoder_outputs + encoder_outputs) */
int kmb_gen_encoder_states(kmb_handle* h, kmb_bf16* enc_out, void* stream);
/* one cached decoder step (src/model/mixins.py:386-398 -> model.py:384-397): tokens [B*num_beams]
 * at position `step` (0-based), logits_out fp32 [B*num_beams, kmb_logits_ld()] */
int kmb_gen_step(kmb_handle* h, const int64_t* tokens, int step, float* logits_out, void* stream);
/* _reorder_cache (src/model/mixins.py:419-434): self-attention caches follow beam_idx [B*num_beams] */
int kmb_gen_reorder(kmb_handle* h, const int32_t* beam_idx, int step, void* stream);
/* The final decoder states of the last kmb_gen_step as bf16 [rows, d_model] (the `decoder_outputs[0]` of a cached bare-model
 * forward, reference src/model/model.py:87-103 with use_cache; transformers 3.0.2 BartDecoder returns the one new position). */
int kmb_gen_last_hidden(kmb_handle* h, kmb_bf16* out, void* stream);
/* log_softmax + top-k per row of (logp + add[row]); force_token >= 0 forces that token
 * (adjust_logits_during_generation, src/model/mixins.py:400-417); ban_token >= 0 scores that token -inf AFTER the
 * normalisation (min_length: transformers 3.0.2 postprocess_next_token_scores acts on the log-probabilities) */
int kmb_logsoftmax_topk(const float* logits, int ld, int V, int rows, const float* add, int force_token, int ban_token,
                        int k, float* out_val, int32_t* out_idx, void* stream);
/* the same with a scratch buffer of kmb_logsoftmax_topk_scratch(rows) floats: every row is split over four workgroups and
 * combined by a second launch -- the form the decode loop uses (a few hundred rows); results are identical up to the
 * rounding of the log-sum-exp */
int64_t kmb_logsoftmax_topk_scratch(int rows);
int kmb_logsoftmax_topk_ws(const float* logits, int ld, int V, int rows, const float* add, int force_token, int ban_token,
                           int k, float* out_val, int32_t* out_idx, float* scratch, int64_t scratch_floats, void* stream);
/* one beam-search step's candidate selection (mixins.py beam loop: topk over num_beams * V): per batch item the best k of
 * its beams' top-k lists; out[B][k][2] int32 = {fp32 score bits, beam * V + token} */
int kmb_beam_merge(const float* val, const int32_t* idx, int B, int num_beams, int k, int V, int32_t* out, void* stream);
/* the same, and the beams of the next decode step chosen on the device: in candidate order the first num_beams candidates
 * whose token is not eos_token (-1: none) -> next_scores [B*num_beams] (the `add` of the next kmb_logsoftmax_topk),
 * next_tokens [B*num_beams] (int64, the next kmb_gen_step), next_beam_idx [B*num_beams] (kmb_gen_reorder): the selection
 * of transformers 3.0.2 _generate_beam_search (reached from src/model/mixins.py:336-361) without its host round trip;
 * `out` still carries every candidate for the host's hypothesis bookkeeping. */
int kmb_beam_merge_select(const float* val, const int32_t* idx, int B, int num_beams, int k, int V, int32_t* out,
                          int eos_token, float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx, void* stream);
/* kmb_logsoftmax_topk_ws over the B * num_beams rows + kmb_beam_merge_select in two launches instead of three (one when the
 * token is forced): the per-row top-k lists stay in the workgroup that merges them.  Same outputs as the two calls
 * (src/model/mixins.py:336-361, one beam-search step).  Fails for shapes the fused form does not cover (k > 16,
 * num_beams > 16, num_beams * k > 256): use the two calls. */
int kmb_beam_step(const float* logits, int ld, int V, int B, int num_beams, const float* add, int force_token, int ban_token,
                  int k, int32_t* out, int eos_token, float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx,
                  float* scratch, int64_t scratch_floats, void* stream);
/* The decode loop's form of kmb_beam_step, on the logits the last kmb_gen_step wrote (`logits` / `ld` as passed there; V and B are
 * the handle's): same arguments, outputs and selection.  When that step's vocabulary projection ran the all-rows kernel
 * (257 .. 320 beam rows -- the benchmarked 64 x 5) it also left every row's maximum and sum-exp per 256-column block, and ONE
 * launch takes the row's log-sum-exp from those and reads only the blocks that can hold one of the k best, instead of
 * streaming the logits a second time; the scores then differ from kmb_beam_step's in the last bits (the log-sum-exp is
 * grouped by 197 blocks instead of 4 parts), ties still go to the smaller index.  Any other shape, a forced token, or
 * KMB_GEN_HEAD_STATS=0 in the environment: kmb_beam_step itself.  Reference: one step of transformers 3.0.2
 * _generate_beam_search as reached from src/model/mixins.py:336-361, scores adjusted as in mixins.py:386-417.
 * reorder_step >= 0: the call is also kmb_gen_reorder(h, next_beam_idx, reorder_step, stream) (_reorder_cache,
 * src/model/mixins.py:419-434) -- with the history index by the launch that has just chosen the beams; -1: no reorder. */
int kmb_gen_beam_step(kmb_handle* h, const float* logits, int ld, int num_beams, const float* add, int force_token, int ban_token,
                      int k, int32_t* out, int eos_token, float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx,
                      float* scratch, int64_t scratch_floats, int reorder_step, void* stream);
int64_t kmb_gen_workspace_bytes(const kmb_handle* h, int B, int S, int num_beams, int max_length, int n_features);

/* Data-parallel runs share the GPU between the GEMMs and RCCL's all-reduce kernel (reference: torch DDP's NCCL streams,
 * vcg_train.py:98).  The persistent GEMM variants keep one workgroup per CU; with on != 0 they hand out EVERY tile through
 * an atomic counter, so workgroups that cannot be placed while the communication kernel holds CUs leave no work behind.
 * Process-wide; results do not depend on it. */
int kmb_gemm_shared_device(int on);

/* ================= data parallelism: native RCCL (vcg_train.py:98 DDP, src/utils.py:9-17 init_process_group) =================
 * One process per GPU; the library owns the communicator and a communication stream, and a step's whole gradient
 * exchange is ONE call: every gradient bucket (kmb_bucket_range, backward completion order) is reduced on the
 * communication stream behind that bucket's completion event of kmb_backward, in pieces of at most max_piece_elems,
 * so the collectives overlap the rest of backward; with `adamw` set each piece's fused optimizer update is chained
 * right behind its collective on the same stream.  Semantics = torch DistributedDataParallel's: after the call every
 * rank holds the arithmetic MEAN over ranks of the per-rank gradients.  The communicator is bootstrapped with a
 * 128-byte id made by rank 0 (kmb_comm_unique_id) and handed to the other ranks by any host channel. */
#define KMB_COMM_ID_BYTES 128
int kmb_comm_unique_id(void* id_host);                                   /* ncclGetUniqueId: call on rank 0 */
int kmb_comm_init(kmb_handle* h, int rank, int world, const void* id_host);   /* ncclCommInitRank on the current device */
int kmb_comm_destroy(kmb_handle* h);
int kmb_comm_info(const kmb_handle* h, int32_t* rank, int32_t* world);   /* world 0: no communicator */
/* parameters and final_logits_bias of rank `root` to every rank (DDP's broadcast at wrap time), bf16 mirror refreshed */
int kmb_comm_broadcast_params(kmb_handle* h, int root, void* stream);
typedef struct kmb_allreduce_opts {
  int32_t algo;               /* 0: ncclAllReduce(avg) per piece; 1: ncclReduceScatter(avg) -> [AdamW on this rank's shard ->
                               * ncclAllGather of the updated parameters] (ZeRO-1 form: optimizer traffic / world; moments are
                               * valid on the owning rank only until kmb_comm_gather_moments); world must divide 8 */
  int32_t after_compute;      /* 1: the communication stream first waits for everything already enqueued on compute_stream */
  int64_t max_piece_elems;    /* <= 0: 16 Mi elements (64 MB of fp32) */
  const KmbAdamW* adamw;      /* NULL: gradients only (algo 1 then all-gathers the gradients) */
} kmb_allreduce_opts;
int kmb_allreduce_grads(kmb_handle* h, const kmb_allreduce_opts* opts, void* compute_stream);
int kmb_comm_wait(kmb_handle* h, void* compute_stream);                  /* compute_stream waits for the communication stream */
int kmb_comm_gather_moments(kmb_handle* h, void* compute_stream);        /* algo 1: every rank gets every shard's exp_avg / exp_avg_sq */
int64_t kmb_comm_pieces(const kmb_handle* h, int64_t max_piece_elems);   /* number of collectives one kmb_allreduce_grads issues */
/* The partition itself, as a PURE HOST function of the arena layout (no communicator, no device: the world > 1 offsets of
 * kmb_allreduce_grads / kmb_comm_gather_moments come from here and are checked on a CPU, tests/test_comm_plan_cpu.py).
 * Piece i (0 <= i < kmb_comm_pieces) of a `world`-rank exchange as rank `rank` sees it; what DDP's bucket assignment is
 * in the reference (vcg_train.py:98). */
typedef struct kmb_comm_piece {
  int32_t bucket;        /* gradient bucket (kmb_bucket_range) whose completion event the piece waits for */
  int32_t repad_piece;   /* 1: the piece overlaps the image-projection weight (its padded bf16 copy is rebuilt after an update) */
  int32_t repad_shard;   /* 1: this rank's shard does */
  int32_t reserved;
  int64_t offset, count; /* arena range of the piece, in elements; 64-element aligned */
  int64_t shard;         /* count / world, 8-element aligned; 0 when the piece does not split that way (algo 1 refuses it) */
  int64_t mine;          /* offset + rank * shard: the range [mine, mine + shard) this rank reduces / updates / owns moments of */
} kmb_comm_piece;
int kmb_comm_plan(const kmb_handle* h, int world, int rank, int64_t max_piece_elems, int64_t i, kmb_comm_piece* out);
/* 1 after an algo-1 exchange with a fused optimizer on more than one rank (exp_avg / exp_avg_sq current on the owning
 * rank's shards only), 0 again after kmb_comm_gather_moments */
int kmb_comm_moments_sharded(const kmb_handle* h);

/* ================= measurement ================= */
/* time every GEMM launch of the following calls with HIP events on its own stream (bench.py roofline leg);
 * variant = a_kc*2 + b_kc: 3 forward (X W^T), 2 dgrad (dY W), 0 wgrad (dY^T X) */
int kmb_debug_trace(int on);                          /* diagnostic: checksum intermediate buffers of kmb_backward */
int kmb_debug_trace_dump(const char* path);           /* "index layer name checksum" per recorded buffer */
int kmb_set_side_stream(kmb_handle* h, int enable);   /* weight-gradient GEMMs on a side stream (default on) */
int kmb_profile_gemm(int enable);
int kmb_profile_read(int variant, int64_t* launches, double* total_ms, double* total_flops);
int kmb_profile_dump(const char* path);   /* one text line per profiled GEMM launch */
/* out16 (device, zeroed by the caller): per XCD x (s_memtime shader-clock ticks, s_memrealtime 100 MHz ticks) at the point
 * of the stream where it runs; (d ticks / d real) x 100 MHz between two stamps = the clock the chip held in between
 * (bench.py `clock_mhz`: a 3 % box difference shows as clock, not as a regression) */
int kmb_clock_stamp(int64_t* out16, void* stream);

/* The reference's batch carries the region features as a LIST of per-sample [R_i, image_feature_size] fp32 tensors
 * (/root/reference/src/data/collation.py:73-76; /root/reference/src/model/modules.py:24-41 concatenates the non-empty ones).
 * rows_dev_ptrs: HOST array of n DEVICE pointers (entry i may be NULL when rows[i] == 0), rows: HOST array of the R_i.
 * packed_out (device, [sum R_i, feat_dim] fp32) receives them in list order -- kmb_batch.image_features -- in ceil(n / 128)
 * kernel launches on `stream`, no host synchronisation; the caller keeps the source tensors alive until the stream has passed. */
int kmb_pack_features(const float* const* rows_dev_ptrs, const int32_t* rows, int32_t n, int32_t feat_dim, float* packed_out,
                      void* stream);

/* ================= single operators (unit tests / profiling) ================= */
int kmb_op_gemm(const KmbGemm* p, void* stream);
/* the same product on the "all rows" kernel (one workgroup per 256 output columns holds every row: M <= 320, forward layout, bias
 * only, fp32 output): what a generation decode step's vocabulary projection runs; bit-identical to kmb_op_gemm */
int kmb_op_gemm_allrows(const KmbGemm* p, void* stream);
/* ... and its statistics epilogue: stats[(row * blocks + blk) * 2] = max, [... + 1] = sum of exp(v - max) over the
 * 256 columns of block blk, blocks = ceil(N / 256) (kmb_op_gemm_allrows_stats_floats(N) floats); kmb_beam_step_stats is kmb_beam_step selecting from them
 * (stats_blocks = ceil(V / 256); what kmb_gen_step + kmb_gen_beam_step run) */
int64_t kmb_op_gemm_allrows_stats_floats(int N);
int kmb_op_gemm_allrows_stats(const KmbGemm* p, float* stats, void* stream);
int kmb_beam_step_stats(const float* logits, int ld, int V, int B, int num_beams, const float* add, int force_token, int ban_token,
                        int k, int32_t* out, int eos_token, float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx,
                        const float* stats, int stats_blocks, void* stream);
/* n (1 .. 8) independent weight-gradient products dW_i = dY_i^T X_i (both operands token-major, fp32 output, no split-K) as ONE
 * launch walking all their 128 x 128 tiles: what kmb_backward issues per layer when the batch is short (<= 3072 tokens: `group_tokens`,
 * csrc/engine.cpp::backward_impl; torch
 * autograd's per-Linear weight gradients behind the reference's loss.backward(), src/training.py:55-59, 138-142); every output
 * bit-identical to the same problem through kmb_op_gemm */
int kmb_op_gemm_group(const KmbGemm* probs, int32_t n, void* stream);
int kmb_op_attn_fwd(const KmbAttn* p, void* stream);
int kmb_op_attn_bwd(const KmbAttn* p, void* stream);
int kmb_op_attn_decode(const KmbAttnDecode* p, void* stream);
int kmb_op_decode_block(const KmbDecodeBlock* p, void* stream);
/* packed <- W ([N, K] row-major, row stride ld; N % 16 == 0, K % 64 == 0) in the order KmbDecodeBlock.W is read */
int kmb_op_decode_pack(const kmb_bf16* W, int ld, int N, int K, kmb_bf16* packed, void* stream);
int kmb_op_ln_fwd(const kmb_bf16* z, const float* gamma, const float* beta, kmb_bf16* y, float* mean, float* rstd,
                  int M, int D, float eps, void* stream);
/* partials: device float scratch of kmb_op_ln_bwd_scratch(M, D) floats */
int64_t kmb_op_ln_bwd_scratch(int M, int D);
int kmb_op_ln_bwd(const kmb_bf16* dy, const kmb_bf16* z, const float* mean, const float* rstd, const float* gamma,
                  kmb_bf16* dz, kmb_bf16* out2, const KmbDrop* dy_drop, const KmbDrop* out2_drop, float* dgamma,
                  float* dbeta, float* scratch, int M, int D, void* stream);
int64_t kmb_op_colsum_scratch(int M, int N);
int kmb_op_colsum(const kmb_bf16* X, int ld, int M, int N, float* out, float* scratch, void* stream);
int kmb_op_img_rowmap(const int64_t* ids, const int32_t* feat_off, int B, int S, int64_t img_feat_id, int64_t cls_id,
                      int32_t* img_src, int32_t* status, void* stream);
int kmb_op_cast_pad(const float* x, int N, int Fin, kmb_bf16* y, int Fpad, void* stream);
int kmb_op_embed_ln_fwd(const int64_t* ids, const int32_t* img_src, const float* E, const float* img_emb,
                        const float* P, int pos_base, int S, float scale, const float* gamma, const float* beta,
                        kmb_bf16* z, kmb_bf16* y, float* mean, float* rstd, int M, int D, float eps,
                        const KmbDrop* drop, void* stream);
int kmb_op_embed_bwd(const kmb_bf16* dz, const int64_t* ids, const int32_t* img_src, float scale, float* dE,
                     kmb_bf16* dimg, int64_t pad_id, int M, int D, void* stream);
int kmb_op_pos_bwd(const kmb_bf16* dz, int B, int S, int D, float* dP, int pos_base, int P_rows, void* stream);
int kmb_op_ce(const float* logits, int ldv, int V, const int64_t* labels, int rows, float grad_scale,
              float* loss_rows, kmb_bf16* dlogits, int32_t* count, float* loss, void* stream);
int kmb_op_adamw(float* p, const float* g, float* m, float* v, kmb_bf16* p_bf16, int64_t n, const KmbAdamW* hp,
                 void* stream);
int kmb_op_cast_bf16(const float* x, kmb_bf16* y, int64_t n, void* stream);
/* keep[i] = 1 if the dropout generator keeps element (row, col) for this site seed / probability */
int kmb_op_dropout_mask(uint32_t seed, float p, int rows, int cols, uint8_t* keep, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KMBART_H */
