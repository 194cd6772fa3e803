// Internal launch API of the HIP kernels (host side).  Every launcher enqueues on `stream`
// and returns hipGetLastError(); none of them allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/kmbart.h"

typedef uint16_t bf16_t;

// ------------------------------------------------------------------ gemm.hip
const char* kmb_gemm_check(const KmbGemm& p);
hipError_t kmb_gemm_launch(const KmbGemm& p, hipStream_t stream);
// the "all rows" kernel for decode-sized forward GEMMs (<= 320 rows, fp32 output, bias only): bit-identical to kmb_gemm_launch
constexpr int KMB_GEMM_GROUP_MAX = 8;
const char* kmb_gemm_group_check(const KmbGemm* probs, int n);
hipError_t kmb_gemm_group_launch(const KmbGemm* probs, int n, hipStream_t stream);
const char* kmb_gemm_allrows_check(const KmbGemm& p);
// stats != nullptr: also per (row, 256-column block) the maximum logit and the sum of exp(v - maximum), kmb_gemm_allrows_stats_floats(N) floats
hipError_t kmb_gemm_allrows_launch(const KmbGemm& p, float* stats, hipStream_t stream);
int kmb_gemm_allrows_blocks(int N);
size_t kmb_gemm_allrows_stats_floats(int N);
void kmb_gemm_set_shared_device(int on);   // persistent variants: hand out every tile dynamically
// variant 6 (gemm_lean.hip): the eight-wave persistent 256 x 256 kernel rebuilt around the bare K loop of tools/mfma_loop.hip
bool kmb_gemm_lean_ok(const KmbGemm& p);
hipError_t kmb_gemm_lean_launch(const KmbGemm& p, hipStream_t stream, uint32_t* sched, int dyn_first);
// variant 9 (gemm_pair.hip): two persistent 256 x 128 workgroups per CU, 32-deep stages (forward layout)
bool kmb_gemm_pair_ok(const KmbGemm& p);
hipError_t kmb_gemm_pair_launch(const KmbGemm& p, hipStream_t stream);
// variant 10 (tools/experiments/gemm_rolesplit.hip, experiment builds only: -DKMB_WITH_ROLESPLIT): role-split persistent
// kernel, epilogue of tile t under the MFMAs of tile t + 1
bool kmb_gemm_rs_ok(const KmbGemm& p);
hipError_t kmb_gemm_rs_launch(const KmbGemm& p, hipStream_t stream);

// ------------------------------------------------------------- attention.hip
hipError_t kmb_attn_fwd_launch(const KmbAttn& p, hipStream_t stream);
hipError_t kmb_attn_bwd_launch(const KmbAttn& p, hipStream_t stream);
const char* kmb_attn_check(const KmbAttn& p, int backward);

// single-query attention over a KV cache (generation): one query row per (row, head)
hipError_t kmb_attn_decode_launch(const KmbAttnDecode& p, hipStream_t stream);

// --------------------------------------------------------------- decode.hip
// fused [LayerNorm ->] projection [-> attention] block of a decode step (R = batch x beams rows)
const char* kmb_decode_block_check(const KmbDecodeBlock& p);
hipError_t kmb_decode_block_launch(const KmbDecodeBlock& p, hipStream_t stream);
// the resident decoder-layers kernel (decode.hip, round 5): n_layers whole decoder layers of a decode step in ONE launch,
// 12 co-resident workgroups per 16-row tile separated by counter barriers instead of kernel boundaries
constexpr int KMB_DL_MAX_LAYERS = 6;
struct KmbDecodeLayerP {
  const bf16_t *Wqkv, *Wo, *Wcq, *Wco, *W1, *W2;      // fragment-order copies (kmb_decode_pack_launch)
  const float *bqkv, *bo, *bcq, *bco, *b1, *b2;
  const float *lnin_g, *lnin_b;                       // LayerNorm of the rows entering the layer (null: already normalised)
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;         // self_attn_layer_norm, encoder_attn_layer_norm
  bf16_t *Kc, *Vc;                                    // self-attention caches [R, Tmax, 768]
  const bf16_t *cK, *cV;                              // cross-attention keys / values of this layer [B * S rows, stride ldc]
};
struct KmbDecodeLayers {
  KmbDecodeLayerP L[KMB_DL_MAX_LAYERS];
  int n_layers;
  const bf16_t* x_in;             // [R, 768] rows entering L[0]
  bf16_t *o, *z, *hh;             // exchange buffers [R, 768], [R, 768] (also the output: the last layer's pre-LayerNorm sums), [R, F]
  unsigned* bars;                 // kmb_decode_layers_bar_words() counters, zeroed by the launcher
  int32_t* status;                // status VALUE 8 (bit 3): a group barrier gave up
  int R, F, H, Tmax, Tk, S, ldc, kv_group;
  const int64_t* key_mask; int mask_ld;
  float eps, q_scale;
  int32_t* hist;                  // optional history index of the self-attention caches (KmbAttnDecode.hist)
};
size_t kmb_decode_layers_lds(int Tmax, int S, int F);
size_t kmb_decode_layers_bar_words(int R, int n_layers);
const char* kmb_decode_layers_check(const KmbDecodeLayers& a);
hipError_t kmb_decode_layers_launch(const KmbDecodeLayers& a, hipStream_t stream);
// packed[i] <- copy of the row-major weight W[i] ([N, K], row stride ld) in the MFMA-fragment order the decode blocks read
// (N % 16 == 0, K % 64 == 0), n <= 48 matrices in one launch
hipError_t kmb_decode_pack_launch(const bf16_t* const* W, const int* ld, const int* N, const int* K, bf16_t* const* packed, int n,
                                  hipStream_t stream);

// ---------------------------------------------------------------- norm.hip

// y = LN(z) * gamma + beta ; saves mean / rstd.  z, y bf16 [M, D]
hipError_t kmb_ln_fwd_launch(const bf16_t* z, const float* gamma, const float* beta, bf16_t* y,
                             float* mean, float* rstd, int M, int D, float eps, hipStream_t stream);
// dz = LN backward; partial dgamma / dbeta / column-sums-of-(out2 or dz) go to `partials` [nparts][3][D]
// (nparts returned by kmb_ln_bwd_parts)
// dy_drop: dropout that was applied to the LN OUTPUT in forward (embedding LN), thr16 == 0 if none.
// dz_drop: if out2 != null, out2 = dz * keep(out2 site) * scale (gradient of a dropped sub-layer output).
int kmb_ln_bwd_parts(int M);
hipError_t kmb_ln_bwd_launch(const bf16_t* dy, const bf16_t* z, const float* mean, const float* rstd,
                             const float* gamma, bf16_t* dz, bf16_t* out2, KmbDrop dy_drop, KmbDrop out2_drop,
                             float* partials, int M, int D, hipStream_t stream);
// out[c] = sum_p partials[p*stride + c]  for c < n   (overwrites)
hipError_t kmb_reduce_parts_launch(const float* partials, int nparts, int stride, float* out, int n,
                                   hipStream_t stream);
hipError_t kmb_reduce_parts2_launch(const float* partials, int nparts, int stride, float* out1, int n1, float* out2, int n2,
                                    hipStream_t stream);
hipError_t kmb_ln_fwd_slabs_launch(const float* slabs, int nslabs, size_t stride, const float* bias, const bf16_t* residual,
                                   int ld_res, const float* gamma, const float* beta, bf16_t* y, int M, int D, float eps,
                                   hipStream_t stream);
hipError_t kmb_reduce_slabs_bf16_launch(const float* slabs, int nslabs, size_t stride, bf16_t* out, size_t n,
                                        hipStream_t stream);
hipError_t kmb_reduce_slabs_launch(const float* slabs, int nslabs, size_t stride, float* out, size_t n, float beta,
                                   hipStream_t stream);
// out[M, N] (bf16, row stride ld_out) = dropout((sum of slabs + bias) * q-scale) + residual   (N % 8 == 0)
hipError_t kmb_reduce_slabs_epi_launch(const float* slabs, int nslabs, size_t stride, const float* bias, float col_scale,
                                       int col_scale_n, KmbDrop drop, const bf16_t* residual, int ld_res, bf16_t* out,
                                       int ld_out, int M, int N, hipStream_t stream);
// column sums of a bf16 matrix -> partials [nparts][N]; nparts = kmb_colsum_parts(M)
int kmb_colsum_parts(int M);
hipError_t kmb_colsum_launch(const bf16_t* X, int ld, int M, int N, float* partials, hipStream_t stream);

// ---------------------------------------------------------------- embed.hip
// img_src[b*S + s] = packed feature row feeding token (b,s), or -1.  status[0] |= 1 on a count mismatch.
hipError_t kmb_img_rowmap_launch(const int64_t* ids, const int32_t* feat_off, int B, int S, int64_t img_feat_id,
                                 int64_t cls_id, int32_t* img_src, int32_t* status, hipStream_t stream);
// fp32 region features [N, Fin] -> bf16 [N, Fpad] (zero padded)
hipError_t kmb_cast_pad_launch(const float* x, int N, int Fin, bf16_t* y, int Fpad, hipStream_t stream);
constexpr int KMB_PACK_MAX = 128;   // per-sample feature tensors gathered by one launch (the list travels as a kernel argument: 2 KB)
struct KmbPackList { const float* src[KMB_PACK_MAX]; int32_t rows[KMB_PACK_MAX]; int32_t off[KMB_PACK_MAX]; int n; };
hipError_t kmb_pack_features_launch(const KmbPackList& l, int feat_dim, float* dst, hipStream_t stream);
// z = (img_src>=0 ? img_emb[img_src] : E[id]) * scale + P[pos_base + (row % S)] ; y = dropout(LN(z))
hipError_t kmb_embed_ln_fwd_launch(const int64_t* ids, const int32_t* img_src, const float* E, const float* img_emb,
                                   const float* P, int pos_base, int S, float scale, const float* gamma,
                                   const float* beta, bf16_t* z, bf16_t* y, float* mean, float* rstd, int M, int D,
                                   float eps, KmbDrop drop, hipStream_t stream);
// token rows: dE[id] += dz*scale (atomic); image rows: dimg[img_src] = dz*scale (bf16, packed)
// (token rows whose id == pad_id get no gradient: nn.Embedding(padding_idx))
hipError_t kmb_embed_bwd_launch(const bf16_t* dz, const int64_t* ids, const int32_t* img_src, float scale,
                                float* dE, bf16_t* dimg, int64_t pad_id, int M, int D, hipStream_t stream);
// dP[pos_base + s] = sum_b dz[b*S + s]; every other row of dP[0:P_rows] is zeroed
hipError_t kmb_pos_bwd_launch(const bf16_t* dz, int B, int S, int D, float* dP, int pos_base, int P_rows,
                              hipStream_t stream);

// ----------------------------------------------------------------- loss.hip
// count[0] = number of labels in [0, V); labels that are neither -100 nor in range set bit 1 of status[0] (status may be null)
hipError_t kmb_count_valid_launch(const int64_t* labels, int n, int V, int32_t* count, int32_t* status, hipStream_t stream);
// tied-head cross-entropy without a pass over the logits (loss.hip, KmbGemm act 5)
hipError_t kmb_ce_label_logit_launch(const bf16_t* H, int ldh, const bf16_t* E, int lde, const float* bias, const int64_t* labels,
                                     int rows, int d, int V, float* shift, hipStream_t stream);
hipError_t kmb_ce_pad_bias_launch(const float* bias, int V, int Vpad, float* out, hipStream_t stream);
// S_r, loss_r, a_r = lm_factor / (count S_r), ah = a . H (bf16, may be null), and P[r][label_r] := exp(pick_r) - S_r (P may be null)
hipError_t kmb_ce_rows_finish_launch(const float* row_sums, int ld_sums, int nparts, const float* pick, const int64_t* labels,
                                     const int32_t* count, float lm_factor, int rows, int d, int V, const bf16_t* H, int ldh,
                                     float* loss_rows, float* srow, float* alpha, bf16_t* ah, bf16_t* P, int ldp, hipStream_t stream);
// out[r] = bf16(alpha[r] * sum_s slab[s][r])
hipError_t kmb_ce_dgrad_finish_launch(const float* slab, int nslabs, size_t stride, const float* alpha, bf16_t* out, int rows, int d,
                                      hipStream_t stream);
// per row: loss_rows[r] = lse - logit[label] (0 if ignored); dlogits (bf16, ld = ldv, pad columns zeroed)
//          = (softmax - onehot) * grad_scale / count   (0 rows if ignored).  dlogits may be null.
hipError_t kmb_ce_launch(const float* logits, int ldv, int V, const int64_t* labels, int rows,
                         const int32_t* count, float grad_scale, float* loss_rows, bf16_t* dlogits,
                         hipStream_t stream);
// the same on bf16 logits (ldv <= 65536), dlogits may alias logits (in place)
hipError_t kmb_ce_bf16_launch(const bf16_t* logits, int ldv, int V, const int64_t* labels, int rows, const int32_t* count,
                              float grad_scale, float* loss_rows, bf16_t* dlogits, hipStream_t stream);
// loss[0] = sum(loss_rows) / count
hipError_t kmb_loss_finish_launch(const float* loss_rows, int rows, const int32_t* count, float* loss,
                                  hipStream_t stream);
// generation: per row log_softmax over V then top-k of (logp + add[row]); writes k (value, index) pairs
// hist (optional, needs next_*): the launch that picks the next beams also gathers the self-attention caches' history index for them:
// dst[r][t] = src[next_beam_idx[r]][t], t < nt, rows of ld ints (kmb_gather_hist_launch's work, one launch less per decode step)
struct KmbHistGather { const int32_t* src; int32_t* dst; int ld, nt; };
// embed (optional, needs next_*; 512 < D <= 1024): ... and embeds the tokens it has chosen for the next decode step: row r of y [rows, D] bf16 =
// LayerNorm(E[next_tokens[r]] * scale + prow) (kmb_embed_ln_fwd_launch's work on those tokens, bit-identical rows; E == nullptr: no)
struct KmbEmbedNext { const float* E = nullptr; const float* prow = nullptr; const float* gamma = nullptr; const float* beta = nullptr;
                      bf16_t* y = nullptr; float scale = 1.f; int D = 0; float eps = 0.f; int V = 0; };   // V: rows of E
hipError_t kmb_beam_step_launch(const float* logits, int ldv, int V, int B, int nb, const float* add, int force_token, int ban_token,
                                int k, int32_t* out, int eos, float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx,
                                float* scratch, size_t scratch_floats, hipStream_t stream, const KmbHistGather* hist = nullptr,
                                const KmbEmbedNext* embed = nullptr);
hipError_t kmb_beam_step_stats_launch(const float* logits, int ldv, int V, int B, int nb, const float* add, int force_token, int ban_token,
                                      int k, int32_t* out, int eos, float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx,
                                      const float* stats, int nblk, hipStream_t stream, const KmbHistGather* hist = nullptr,
                                      const KmbEmbedNext* embed = nullptr);
hipError_t kmb_beam_merge_launch(const float* val, const int32_t* idx, int B, int nb, int k, int V, int32_t* out, int eos,
                                 float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx, hipStream_t stream);
// ban_token >= 0: that token's score is -inf AFTER the normalisation (min_length, transformers 3.0.2
// postprocess_next_token_scores)
// scratch (optional, kmb_logsoftmax_topk_scratch_floats(rows) floats): rows are split over four workgroups each and
// combined by a second launch (decode-sized problems); without it one workgroup per row
size_t kmb_logsoftmax_topk_scratch_floats(int rows);
hipError_t kmb_logsoftmax_topk_launch(const float* logits, int ldv, int V, int rows, const float* add,
                                      int force_token, int ban_token, int k, float* out_val, int32_t* out_idx,
                                      float* scratch, size_t scratch_floats, hipStream_t stream);

// ---------------------------------------------------------------- optim.hip
// transformers-3.0.2 AdamW on a flat fp32 arena; also refreshes the bf16 mirror of the parameters
hipError_t kmb_adamw_launch(float* p, const float* g, float* m, float* v, bf16_t* p_bf16, size_t n,
                            KmbAdamW h, hipStream_t stream);
hipError_t kmb_cast_f32_bf16_launch(const float* x, bf16_t* y, size_t n, hipStream_t stream);
hipError_t kmb_fill_f32_launch(float* x, float v, size_t n, hipStream_t stream);
// y[r*ldy + c] = bf16(x[r*ldx + c]) for c < cols, zero for cols <= c < ldy
hipError_t kmb_cast_rows_launch(const float* x, int ldx, bf16_t* y, int ldy, int rows, int cols, hipStream_t stream);
// copy a KV projection [R, ld] slice into a cache [R, Tmax, HD] at position t
hipError_t kmb_kv_append_launch(const bf16_t* src, int ld_src, bf16_t* cache, int Tmax, int HD, int t, int R,
                                hipStream_t stream);
// gather rows: dst[i] = src[idx[i]]  (16-byte chunks, row_bytes % 16 == 0; rows are `stride_bytes` apart)
hipError_t kmb_gather_rows_multi_launch(const void* const* src, void* const* dst, int n, const int32_t* idx, int rows,
                                        int row_bytes, size_t stride_bytes, hipStream_t stream);
// dst[i] = src[idx[i]]  (dst must not alias src)
hipError_t kmb_gather_i32_launch(const int32_t* src, const int32_t* idx, int32_t* dst, int n, hipStream_t stream);
hipError_t kmb_gather_hist_launch(const int32_t* src, const int32_t* idx, int32_t* dst, int rows, int ld, int nt, hipStream_t stream);
hipError_t kmb_iota_div_launch(int32_t* out, int n, int div, hipStream_t stream);   // out[i] = i / div
hipError_t kmb_gather_rows_launch(const void* src, const int32_t* idx, void* dst, int rows, int row_bytes,
                                  size_t stride_bytes, hipStream_t stream);

// keep-mask dump of the dropout generator (tests)
hipError_t kmb_dropout_mask_launch(uint32_t seed, uint32_t thr16, int rows, int cols, uint8_t* keep, hipStream_t stream);
// x *= s * (s_dev ? *s_dev : 1); no memory traffic when the product is 1
hipError_t kmb_scale_bf16_launch(bf16_t* x, size_t n, float s, const float* s_dev, hipStream_t stream);
hipError_t kmb_scale_f32_launch(float* x, size_t n, float s, const float* s_dev, hipStream_t stream);
hipError_t kmb_hash_words_launch(const void* x, size_t nbytes, unsigned long long* out, hipStream_t stream);

// --------------------------------------------------------- fp32_validate.hip
// fp32 validation forward (kmb_set_precision): the KmbGemm / KmbAttn activation pointers hold floats, B is the fp32 master
const char* kmb_f32_gemm_check(const KmbGemm& p);
hipError_t kmb_f32_gemm_launch(const KmbGemm& p, hipStream_t stream);
hipError_t kmb_f32_attn_fwd_launch(const KmbAttn& p, hipStream_t stream);
hipError_t kmb_f32_ln_fwd_launch(const float* z, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                 int M, int D, float eps, hipStream_t stream);
hipError_t kmb_f32_embed_ln_fwd_launch(const int64_t* ids, const int32_t* img_src, const float* E, const float* img_emb,
                                       const float* P, int pos_base, int S, float scale, const float* gamma,
                                       const float* beta, float* z, float* y, float* mean, float* rstd, int M, int D,
                                       float eps, hipStream_t stream);

// ---------------------------------------------------------------- heads.hip
hipError_t kmb_kl_div_launch(const float* logits, int ld, int C, const float* target, int ldt, int rows,
                             float grad_scale, float* loss_rows, bf16_t* dlogits, int ldd, hipStream_t stream);
hipError_t kmb_gather_rows_bf16_launch(const bf16_t* src, int src_ld, const int32_t* idx, bf16_t* dst, int dst_ld,
                                       int rows, int cols, hipStream_t stream);
hipError_t kmb_scatter_add_rows_launch(const bf16_t* src, int src_ld, const int32_t* idx, float* acc, int rows, int cols,
                                       hipStream_t stream);
hipError_t kmb_add_f32_into_bf16_launch(bf16_t* y, const float* a, size_t n, hipStream_t stream);
// P[b,h,i,j] = exp(q_i . k_j - lse_i), 0 where masked: the attention probabilities the fused kernels do not store (head dim 64)
hipError_t kmb_attn_probs_launch(const bf16_t* Q, int ldq, const bf16_t* K, int ldk, const float* lse, const int64_t* key_mask,
                                 int causal, int B, int H, int Tq, int Tk, float* out, hipStream_t stream);
hipError_t kmb_mean_rows_launch(const float* rows, int n, float factor, float denom, float* out, hipStream_t stream);
