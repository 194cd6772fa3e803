// Pre-training heads (reference src/model/model.py:248-289): soft-label KL divergence for masked-region
// modelling, and the row gather / scatter-add that moves decoder states in and out of the small head GEMMs.
#include "common.h"
#include "kernels.h"

namespace {

// per row r: logp = log_softmax(logits[r, :C]); loss_rows[r] = sum_c t (log t - logp)   (0 where t == 0);
// dlogits[r, c] = (sum_c t) * softmax_c - t_c, times grad_scale / rows  (F.kl_div(..., reduction='batchmean'))
__global__ __launch_bounds__(256) void kl_div_kernel(const float* __restrict__ logits, int ld, int C,
                                                     const float* __restrict__ target, int ldt, int rows,
                                                     float grad_scale, float* __restrict__ loss_rows,
                                                     bf16_t* __restrict__ dlogits, int ldd) {
  __shared__ float sh[12];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* row = logits + (size_t)r * ld;
  const float* trow = target + (size_t)r * ldt;
  float m = -INFINITY;
  for (int c = tid; c < C; c += 256) m = fmaxf(m, row[c]);
  m = wave_max(m);
  if ((tid & 63) == 0) sh[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  float s = 0.f, st = 0.f;
  for (int c = tid; c < C; c += 256) { s += __expf(row[c] - m); st += trow[c]; }
  s = wave_sum(s); st = wave_sum(st);
  if ((tid & 63) == 0) { sh[4 + (tid >> 6)] = s; sh[8 + (tid >> 6)] = st; }
  __syncthreads();
  s = sh[4] + sh[5] + sh[6] + sh[7];
  st = sh[8] + sh[9] + sh[10] + sh[11];
  const float lse = m + __logf(s);
  float l = 0.f;
  const float gs = grad_scale / (float)rows;
  for (int c = tid; c < ldd; c += 256) {
    float g = 0.f;
    if (c < C) {
      const float t = trow[c];
      const float logp = row[c] - lse;
      if (t > 0.f) l += t * (__logf(t) - logp);
      g = (st * __expf(logp) - t) * gs;
    }
    if (dlogits != nullptr) dlogits[(size_t)r * ldd + c] = f2bf(g);
  }
  l = wave_sum(l);
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = l;
  __syncthreads();
  if (tid == 0) loss_rows[r] = sh[0] + sh[1] + sh[2] + sh[3];
}

// dst[i*dst_ld + c] = src[idx[i]*src_ld + c], c < cols (bf16, 16-byte chunks)
__global__ __launch_bounds__(256) void gather_rows_bf16_kernel(const bf16_t* __restrict__ src, int src_ld,
                                                               const int32_t* __restrict__ idx,
                                                               bf16_t* __restrict__ dst, int dst_ld, int rows, int cols) {
  const int chunks = cols >> 3;
  const size_t total = (size_t)rows * chunks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / chunks), c = (int)(i % chunks);
    *reinterpret_cast<u32x4*>(dst + (size_t)r * dst_ld + c * 8) =
        *reinterpret_cast<const u32x4*>(src + (size_t)idx[r] * src_ld + c * 8);
  }
}

// acc[idx[i]*cols + c] += src[i*src_ld + c]   (fp32 atomics; one wave per source row, 256-B contiguous per instruction)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const bf16_t* __restrict__ src, int src_ld,
                                                               const int32_t* __restrict__ idx, float* __restrict__ acc,
                                                               int rows, int cols) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* o = acc + (size_t)idx[row] * cols;
  for (int c = lane; c < cols; c += 64) atomicAdd(o + c, bf2f(src[(size_t)row * src_ld + c]));
}

// y (bf16) += a (fp32), n % 8 == 0
__global__ __launch_bounds__(256) void add_f32_into_bf16_kernel(bf16_t* __restrict__ y, const float* __restrict__ a,
                                                                size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float v[8];
    unpack8(reinterpret_cast<const u32x4*>(y)[i], v);
    const f32x4 a0 = reinterpret_cast<const f32x4*>(a)[2 * i], a1 = reinterpret_cast<const f32x4*>(a)[2 * i + 1];
    v[0] += a0[0]; v[1] += a0[1]; v[2] += a0[2]; v[3] += a0[3];
    v[4] += a1[0]; v[5] += a1[1]; v[6] += a1[2]; v[7] += a1[3];
    reinterpret_cast<u32x4*>(y)[i] = pack8(v);
  }
}

// out[0] = factor * sum(rows) / denom
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ rows, int n, float factor, float denom,
                                                        float* __restrict__ out) {
  __shared__ float sh[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += rows[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = factor * (sh[0] + sh[1] + sh[2] + sh[3]) / denom;
}

inline int grid_for(size_t work, int cap = 4096) {
  size_t b = (work + 255) / 256;
  if (b > (size_t)cap) b = cap;
  return b < 1 ? 1 : (int)b;
}

}  // namespace

hipError_t kmb_kl_div_launch(const float* logits, int ld, int C, const float* target, int ldt, int rows,
                             float grad_scale, float* loss_rows, bf16_t* dlogits, int ldd, hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(kl_div_kernel, dim3(rows), dim3(256), 0, stream, logits, ld, C, target, ldt, rows, grad_scale, loss_rows, dlogits, ldd);
  return hipGetLastError();
}
hipError_t kmb_gather_rows_bf16_launch(const bf16_t* src, int src_ld, const int32_t* idx, bf16_t* dst, int dst_ld,
                                       int rows, int cols, hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  if ((cols & 7) || (src_ld & 7) || (dst_ld & 7)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(grid_for((size_t)rows * (cols >> 3))), dim3(256), 0, stream, src, src_ld, idx, dst, dst_ld, rows, cols);
  return hipGetLastError();
}
hipError_t kmb_scatter_add_rows_launch(const bf16_t* src, int src_ld, const int32_t* idx, float* acc, int rows, int cols,
                                       hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, src, src_ld, idx, acc, rows, cols);
  return hipGetLastError();
}
hipError_t kmb_add_f32_into_bf16_launch(bf16_t* y, const float* a, size_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  if (n & 7) return hipErrorInvalidValue;
  hipLaunchKernelGGL(add_f32_into_bf16_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, stream, y, a, n >> 3);
  return hipGetLastError();
}
hipError_t kmb_mean_rows_launch(const float* rows, int n, float factor, float denom, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(mean_rows_kernel, dim3(1), dim3(256), 0, stream, rows, n, factor, denom, out);
  return hipGetLastError();
}

namespace {
// Attention probabilities of one self-attention, recomputed for `output_attentions` (reference src/model/modules.py:143-165;
// HF 3.0.2 SelfAttention returns softmax(q k^T + masks) as [B, H, Tq, Tk]): the fused attention kernels never store them,
// but they keep q | k (q already scaled) and the rows' log-sum-exp, so P[b,h,i,j] = exp(q_i . k_j - lse_i) with masked
// entries exactly 0.  One workgroup per (b, h, i); not a hot path (diagnostic output of a forward).
__global__ __launch_bounds__(64) void attn_probs_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                        const float* __restrict__ lse, const int64_t* __restrict__ key_mask, int causal,
                                                        int H, int Tq, int Tk, float* __restrict__ out) {
  const int i = blockIdx.x % Tq, h = (blockIdx.x / Tq) % H, b = blockIdx.x / (Tq * H);
  const bf16_t* q = Q + ((size_t)b * Tq + i) * ldq + h * 64;
  const float l = lse[((size_t)b * H + h) * Tq + i];
  float qf[64];
#pragma unroll
  for (int e = 0; e < 64; e += 8) unpack8(*reinterpret_cast<const u32x4*>(q + e), qf + e);
  for (int j = threadIdx.x; j < Tk; j += 64) {
    const bf16_t* k = K + ((size_t)b * Tk + j) * ldk + h * 64;
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 64; e += 8) {
      float kf[8];
      unpack8(*reinterpret_cast<const u32x4*>(k + e), kf);
#pragma unroll
      for (int t = 0; t < 8; ++t) acc = fmaf(qf[e + t], kf[t], acc);
    }
    const bool masked = (causal && j > i) || (key_mask != nullptr && key_mask[(size_t)b * Tk + j] == 0) || l == -INFINITY;
    out[(((size_t)b * H + h) * Tq + i) * Tk + j] = masked ? 0.f : __expf(acc - l);
  }
}
}  // namespace

hipError_t kmb_attn_probs_launch(const bf16_t* Q, int ldq, const bf16_t* K, int ldk, const float* lse, const int64_t* key_mask,
                                 int causal, int B, int H, int Tq, int Tk, float* out, hipStream_t stream) {
  if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return hipSuccess;
  hipLaunchKernelGGL(attn_probs_kernel, dim3((unsigned)(B * H * Tq)), dim3(64), 0, stream, Q, ldq, K, ldk, lse, key_mask, causal, H, Tq, Tk, out);
  return hipGetLastError();
}
