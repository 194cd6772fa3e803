// C-ABI wrappers around single kernels (unit tests, profiling).  See include/kmbart.h.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <string>

#include "kernels.h"

extern "C" const char* kmb_last_error(void);
int kmb_set_error(const char* msg);  // engine.cpp

namespace {
int hipfail(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  char buf[512];
  snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  return kmb_set_error(buf);
}
}  // namespace

// (shader clock ticks, 100 MHz real-time ticks) per XCD: the quotient of two stamps' differences is the clock the chip held in
// between (MI355X_MICROARCH.md "DVFS give-back" item 6).  One wave per workgroup, the grid round-robins over the eight XCDs.
__global__ void clock_stamp_kernel(long long* out) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(0xF814) & 7u;   // HW_REG_XCC_ID
    const long long c = (long long)__builtin_amdgcn_s_memtime(), r = (long long)__builtin_amdgcn_s_memrealtime();
    out[2 * xcc] = c; out[2 * xcc + 1] = r;   // any workgroup of the XCD may win: the pairs are microseconds apart at most
  }
}

extern "C" {

int kmb_clock_stamp(int64_t* out16, void* stream) {
  if (!out16) return kmb_set_error("kmb_clock_stamp: out16 (device, 16 x int64) is required");
  hipLaunchKernelGGL(clock_stamp_kernel, dim3(64), dim3(64), 0, (hipStream_t)stream, (long long*)out16);
  return hipfail(hipGetLastError(), "clock_stamp");
}

// n per-sample region-feature tensors (device pointers in a HOST array; rows[i] rows of feat_dim contiguous floats each, rows[i] may be 0)
// -> packed_out [sum rows, feat_dim] in list order: the batch layout kmb_batch.image_features wants, in ceil(n / 128) launches
int kmb_pack_features(const float* const* rows_dev_ptrs, const int32_t* rows, int32_t n, int32_t feat_dim, float* packed_out, void* stream) {
  if (n < 0 || feat_dim <= 0 || (n > 0 && (!rows_dev_ptrs || !rows || !packed_out))) return kmb_set_error("kmb_pack_features: bad arguments");
  int32_t off = 0;
  for (int32_t i0 = 0; i0 < n; i0 += KMB_PACK_MAX) {
    KmbPackList l;
    l.n = 0;
    for (int32_t i = i0; i < n && i < i0 + KMB_PACK_MAX; ++i) {
      if (rows[i] < 0 || (rows[i] > 0 && !rows_dev_ptrs[i])) return kmb_set_error("kmb_pack_features: null tensor with rows > 0, or a negative row count");
      l.src[l.n] = rows_dev_ptrs[i]; l.rows[l.n] = rows[i]; l.off[l.n] = off; ++l.n;
      off += rows[i];
    }
    const int e = hipfail(kmb_pack_features_launch(l, feat_dim, packed_out, (hipStream_t)stream), "pack_features");
    if (e) return e;
  }
  return 0;
}

int kmb_op_gemm(const KmbGemm* p, void* stream) {
  const char* why = kmb_gemm_check(*p);
  if (why) return kmb_set_error(why);
  return hipfail(kmb_gemm_launch(*p, (hipStream_t)stream), "gemm");
}
int kmb_op_gemm_group(const KmbGemm* probs, int32_t n, void* stream) {
  if (!probs) return kmb_set_error("kmb_op_gemm_group: null problem list");
  const char* why = kmb_gemm_group_check(probs, n);
  if (why) return kmb_set_error(why);
  return hipfail(kmb_gemm_group_launch(probs, n, (hipStream_t)stream), "gemm_group");
}
int kmb_op_gemm_allrows(const KmbGemm* p, void* stream) {
  const char* why = kmb_gemm_check(*p);
  if (!why) why = kmb_gemm_allrows_check(*p);
  if (why) return kmb_set_error(why);
  return hipfail(kmb_gemm_allrows_launch(*p, nullptr, (hipStream_t)stream), "gemm_allrows");
}
int64_t kmb_op_gemm_allrows_stats_floats(int N) { return (int64_t)kmb_gemm_allrows_stats_floats(N); }
int kmb_op_gemm_allrows_stats(const KmbGemm* p, float* stats, void* stream) {
  const char* why = kmb_gemm_check(*p);
  if (!why) why = kmb_gemm_allrows_check(*p);
  if (why) return kmb_set_error(why);
  if (!stats) return kmb_set_error("kmb_op_gemm_allrows_stats: stats is required");
  return hipfail(kmb_gemm_allrows_launch(*p, stats, (hipStream_t)stream), "gemm_allrows_stats");
}
int kmb_beam_step_stats(const float* logits, int ld, int V, int B, int num_beams, const float* add, int force_token, int ban_token,
                        int k, int32_t* out, int eos_token, float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx,
                        const float* stats, int stats_blocks, void* stream) {
  if (!logits || !out || !stats) return kmb_set_error("kmb_beam_step_stats: missing tensor");
  if (!next_scores || !next_tokens || !next_beam_idx) return kmb_set_error("kmb_beam_step_stats: missing output");
  const hipError_t e = kmb_beam_step_stats_launch(logits, ld, V, B, num_beams, add, force_token, ban_token, k, out, eos_token, next_scores,
                                                  next_tokens, next_beam_idx, stats, stats_blocks, (hipStream_t)stream);
  if (e == hipErrorNotSupported)
    return kmb_set_error("kmb_beam_step_stats: unsupported shape (B * num_beams <= 320, k <= 16, num_beams <= 16, V %% 4 == 0, "
                         "stats_blocks == ceil(V / 256))");
  return hipfail(e, "beam_step_stats");
}
int kmb_op_attn_fwd(const KmbAttn* p, void* stream) {
  const char* why = kmb_attn_check(*p, 0);
  if (why) return kmb_set_error(why);
  return hipfail(kmb_attn_fwd_launch(*p, (hipStream_t)stream), "attn_fwd");
}
int kmb_op_attn_bwd(const KmbAttn* p, void* stream) {
  const char* why = kmb_attn_check(*p, 1);
  if (why) return kmb_set_error(why);
  return hipfail(kmb_attn_bwd_launch(*p, (hipStream_t)stream), "attn_bwd");
}
int kmb_op_attn_decode(const KmbAttnDecode* p, void* stream) {
  return hipfail(kmb_attn_decode_launch(*p, (hipStream_t)stream), "attn_decode");
}
int kmb_op_decode_block(const KmbDecodeBlock* p, void* stream) {
  const char* why = kmb_decode_block_check(*p);
  if (why) return kmb_set_error(why);
  return hipfail(kmb_decode_block_launch(*p, (hipStream_t)stream), "decode_block");
}
int kmb_op_decode_pack(const kmb_bf16* W, int ld, int N, int K, kmb_bf16* packed, void* stream) {
  return hipfail(kmb_decode_pack_launch(&W, &ld, &N, &K, &packed, 1, (hipStream_t)stream), "decode_pack");
}
int kmb_op_ln_fwd(const kmb_bf16* z, const float* gamma, const float* beta, kmb_bf16* y, float* mean, float* rstd,
                  int M, int D, float eps, void* stream) {
  return hipfail(kmb_ln_fwd_launch(z, gamma, beta, y, mean, rstd, M, D, eps, (hipStream_t)stream), "ln_fwd");
}
int64_t kmb_op_ln_bwd_scratch(int M, int D) { return (int64_t)kmb_ln_bwd_parts(M) * 3 * D; }
int kmb_op_ln_bwd(const kmb_bf16* dy, const kmb_bf16* z, const float* mean, const float* rstd, const float* gamma,
                  kmb_bf16* dz, kmb_bf16* out2, const KmbDrop* dy_drop, const KmbDrop* out2_drop, float* dgamma,
                  float* dbeta, float* scratch, int M, int D, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const KmbDrop none{0u, 0u, 1.f};
  int rc = hipfail(kmb_ln_bwd_launch(dy, z, mean, rstd, gamma, dz, out2, dy_drop ? *dy_drop : none,
                                     out2_drop ? *out2_drop : none, scratch, M, D, s), "ln_bwd");
  if (rc) return rc;
  const int np = kmb_ln_bwd_parts(M);
  rc = hipfail(kmb_reduce_parts_launch(scratch, np, 3 * D, dgamma, D, s), "ln_bwd reduce");
  if (rc) return rc;
  return hipfail(kmb_reduce_parts_launch(scratch + D, np, 3 * D, dbeta, D, s), "ln_bwd reduce");
}
int64_t kmb_op_colsum_scratch(int M, int N) { return (int64_t)kmb_colsum_parts(M) * N; }
int kmb_op_colsum(const kmb_bf16* X, int ld, int M, int N, float* out, float* scratch, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int rc = hipfail(kmb_colsum_launch(X, ld, M, N, scratch, s), "colsum");
  if (rc) return rc;
  return hipfail(kmb_reduce_parts_launch(scratch, kmb_colsum_parts(M), N, out, N, s), "colsum reduce");
}
int kmb_op_img_rowmap(const int64_t* ids, const int32_t* feat_off, int B, int S, int64_t img_feat_id, int64_t cls_id,
                      int32_t* img_src, int32_t* status, void* stream) {
  return hipfail(kmb_img_rowmap_launch(ids, feat_off, B, S, img_feat_id, cls_id, img_src, status, (hipStream_t)stream), "img_rowmap");
}
int kmb_op_cast_pad(const float* x, int N, int Fin, kmb_bf16* y, int Fpad, void* stream) {
  return hipfail(kmb_cast_pad_launch(x, N, Fin, y, Fpad, (hipStream_t)stream), "cast_pad");
}
int kmb_op_embed_ln_fwd(const int64_t* ids, const int32_t* img_src, const float* E, const float* img_emb,
                        const float* P, int pos_base, int S, float scale, const float* gamma, const float* beta,
                        kmb_bf16* z, kmb_bf16* y, float* mean, float* rstd, int M, int D, float eps,
                        const KmbDrop* drop, void* stream) {
  const KmbDrop none{0u, 0u, 1.f};
  return hipfail(kmb_embed_ln_fwd_launch(ids, img_src, E, img_emb, P, pos_base, S, scale, gamma, beta, z, y, mean,
                                         rstd, M, D, eps, drop ? *drop : none, (hipStream_t)stream), "embed_ln_fwd");
}
int kmb_op_embed_bwd(const kmb_bf16* dz, const int64_t* ids, const int32_t* img_src, float scale, float* dE,
                     kmb_bf16* dimg, int64_t pad_id, int M, int D, void* stream) {
  return hipfail(kmb_embed_bwd_launch(dz, ids, img_src, scale, dE, dimg, pad_id, M, D, (hipStream_t)stream), "embed_bwd");
}
int kmb_op_pos_bwd(const kmb_bf16* dz, int B, int S, int D, float* dP, int pos_base, int P_rows, void* stream) {
  return hipfail(kmb_pos_bwd_launch(dz, B, S, D, dP, pos_base, P_rows, (hipStream_t)stream), "pos_bwd");
}
int kmb_op_ce(const float* logits, int ldv, int V, const int64_t* labels, int rows, float grad_scale,
              float* loss_rows, kmb_bf16* dlogits, int32_t* count, float* loss, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int rc = hipfail(kmb_count_valid_launch(labels, rows, V, count, nullptr, s), "count_valid");
  if (rc) return rc;
  rc = hipfail(kmb_ce_launch(logits, ldv, V, labels, rows, count, grad_scale, loss_rows, dlogits, s), "ce");
  if (rc) return rc;
  return hipfail(kmb_loss_finish_launch(loss_rows, rows, count, loss, s), "loss_finish");
}
int kmb_op_adamw(float* p, const float* g, float* m, float* v, kmb_bf16* p_bf16, int64_t n, const KmbAdamW* hp,
                 void* stream) {
  return hipfail(kmb_adamw_launch(p, g, m, v, p_bf16, (size_t)n, *hp, (hipStream_t)stream), "adamw");
}
int kmb_op_cast_bf16(const float* x, kmb_bf16* y, int64_t n, void* stream) {
  return hipfail(kmb_cast_f32_bf16_launch(x, y, (size_t)n, (hipStream_t)stream), "cast");
}
int kmb_op_dropout_mask(uint32_t seed, float p, int rows, int cols, uint8_t* keep, void* stream) {
  uint32_t thr = (uint32_t)lrintf(p * 65536.f);
  if (thr > 65535u) thr = 65535u;
  return hipfail(kmb_dropout_mask_launch(seed, thr, rows, cols, keep, (hipStream_t)stream), "dropout_mask");
}
int kmb_gemm_shared_device(int on) {
  kmb_gemm_set_shared_device(on);
  return 0;
}
int kmb_beam_merge(const float* val, const int32_t* idx, int B, int num_beams, int k, int V, int32_t* out, void* stream) {
  return hipfail(kmb_beam_merge_launch(val, idx, B, num_beams, k, V, out, -1, nullptr, nullptr, nullptr, (hipStream_t)stream), "beam_merge");
}
int kmb_beam_merge_select(const float* val, const int32_t* idx, int B, int num_beams, int k, int V, int32_t* out,
                          int eos_token, float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx, void* stream) {
  if (!next_scores || !next_tokens || !next_beam_idx) return kmb_set_error("kmb_beam_merge_select: missing output");
  return hipfail(kmb_beam_merge_launch(val, idx, B, num_beams, k, V, out, eos_token, next_scores, next_tokens, next_beam_idx,
                                       (hipStream_t)stream), "beam_merge_select");
}
int kmb_beam_step(const float* logits, int ld, int V, int B, int num_beams, const float* add, int force_token, int ban_token,
                  int k, int32_t* out, int eos_token, float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx,
                  float* scratch, int64_t scratch_floats, void* stream) {
  if (!logits || !out || !scratch) return kmb_set_error("kmb_beam_step: missing tensor");
  if (!next_scores || !next_tokens || !next_beam_idx) return kmb_set_error("kmb_beam_step: missing output");
  const hipError_t e = kmb_beam_step_launch(logits, ld, V, B, num_beams, add, force_token, ban_token, k, out, eos_token, next_scores,
                                            next_tokens, next_beam_idx, scratch, scratch_floats > 0 ? (size_t)scratch_floats : 0,
                                            (hipStream_t)stream);
  if (e == hipErrorNotSupported)
    return kmb_set_error("kmb_beam_step: unsupported shape (k <= 16, num_beams <= 16, num_beams * k <= 256, ld %% 4 == 0, "
                         "scratch of kmb_logsoftmax_topk_scratch(B * num_beams) floats): use kmb_logsoftmax_topk_ws + kmb_beam_merge_select");
  return hipfail(e, "beam_step");
}
int kmb_logsoftmax_topk(const float* logits, int ld, int V, int rows, const float* add, int force_token, int ban_token,
                        int k, float* out_val, int32_t* out_idx, void* stream) {
  return hipfail(kmb_logsoftmax_topk_launch(logits, ld, V, rows, add, force_token, ban_token, k, out_val, out_idx, nullptr, 0,
                                            (hipStream_t)stream), "logsoftmax_topk");
}
int kmb_logsoftmax_topk_ws(const float* logits, int ld, int V, int rows, const float* add, int force_token, int ban_token,
                           int k, float* out_val, int32_t* out_idx, float* scratch, int64_t scratch_floats, void* stream) {
  return hipfail(kmb_logsoftmax_topk_launch(logits, ld, V, rows, add, force_token, ban_token, k, out_val, out_idx, scratch,
                                            scratch_floats > 0 ? (size_t)scratch_floats : 0, (hipStream_t)stream), "logsoftmax_topk");
}
int64_t kmb_logsoftmax_topk_scratch(int rows) { return (int64_t)kmb_logsoftmax_topk_scratch_floats(rows); }

}  // extern "C"
