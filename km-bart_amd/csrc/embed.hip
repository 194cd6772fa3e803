// Multimodal embedding: region-feature row map, fp32->bf16 feature packing, fused
// (token | region) gather + learned position + LayerNorm (+dropout), and its backward.
// Reference: src/model/modules.py:89-102 (_embed_multi_modal), :133-137 (pos + LN + dropout).
#include "common.h"
#include "kernels.h"
#include "embed_row.h"

namespace {

// one wave per sample: the j-th <img_feat>/<cls> id of the row takes packed feature row feat_off[b] + j
__global__ __launch_bounds__(64) void img_rowmap_kernel(const int64_t* __restrict__ ids,
                                                        const int32_t* __restrict__ feat_off, int S,
                                                        int64_t img_feat_id, int64_t cls_id,
                                                        int32_t* __restrict__ img_src, int32_t* __restrict__ status) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int base = feat_off[b];
  const int R = feat_off[b + 1] - base;
  int running = 0;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    bool is_img = false;
    if (s < S) {
      const int64_t id = ids[(size_t)b * S + s];
      is_img = (id == img_feat_id) || (id == cls_id);
    }
    const unsigned long long m = __ballot(is_img);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (s < S) img_src[(size_t)b * S + s] = (is_img && R > 0) ? base + running + before : -1;
    running += __popcll(m);
  }
  // modules.py:98-100: a non-empty feature list must match the number of placeholder ids
  if (lane == 0 && R > 0 && running != R) atomicOr(status, 1);
}

__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ x, int N, int Fin,
                                                       bf16_t* __restrict__ y, int Fpad) {
  const int chunks = Fpad >> 3;
  const size_t total = (size_t)N * chunks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / chunks), c = (int)(i % chunks);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = c * 8 + e;
      v[e] = col < Fin ? x[(size_t)row * Fin + col] : 0.f;
    }
    *reinterpret_cast<u32x4*>(y + (size_t)row * Fpad + c * 8) = pack8(v);
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(
    const int64_t* __restrict__ ids, const int32_t* __restrict__ img_src, const float* __restrict__ E,
    const float* __restrict__ img_emb, const float* __restrict__ P, int pos_base, int S, float scale,
    const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ z,
    bf16_t* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int M, int D, float eps,
    KmbDrop drop) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int src = img_src != nullptr ? img_src[row] : -1;
  const float* erow = src >= 0 ? img_emb + (size_t)src * D : E + (size_t)ids[row] * D;
  const float* prow = P + (size_t)(pos_base + (row % S)) * D;
  embed_ln_row<NCH>(erow, prow, scale, gamma, beta, z, y, mean, rstd, row, D, eps, drop, lane);
}

// one wave per row; lane owns columns lane + 64*j so that each atomic wave-instruction is 256 contiguous bytes
__global__ __launch_bounds__(256) void embed_bwd_kernel(const bf16_t* __restrict__ dz, const int64_t* __restrict__ ids,
                                                        const int32_t* __restrict__ img_src, float scale,
                                                        float* __restrict__ dE, bf16_t* __restrict__ dimg,
                                                        int64_t pad_id, int M, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int src = img_src != nullptr ? img_src[row] : -1;
  const bf16_t* g = dz + (size_t)row * D;
  if (src >= 0) {
    bf16_t* o = dimg + (size_t)src * D;
    for (int c = lane; c < D; c += 64) o[c] = f2bf(bf2f(g[c]) * scale);
  } else {
    const int64_t id = ids[row];
    if (id == pad_id) return;  // nn.Embedding(padding_idx): no gradient for the pad row
    float* o = dE + (size_t)id * D;
    for (int c = lane; c < D; c += 64) atomicAdd(o + c, bf2f(g[c]) * scale);
  }
}

// One block per position row; rows outside [pos_base, pos_base+S) are zeroed.  A position's gradient is the sum over
// the batch of rows b*S + s: 1024 threads = (D/8 column chunks) x (1024 / (D/8) batch lanes), 16-byte loads, the batch
// lanes folded through LDS in a fixed order (deterministic).  The first version gave one thread a column and walked
// the batch serially with 2-byte loads: 145 us for 50 MB.
__global__ __launch_bounds__(1024) void pos_bwd_kernel(const bf16_t* __restrict__ dz, int B, int S, int D,
                                                       float* __restrict__ dP, int pos_base) {
  extern __shared__ float red[];   // [batch lanes][D]
  const int prow = blockIdx.x, tid = threadIdx.x;
  const int s = prow - pos_base;
  if (s < 0 || s >= S) {
    for (int i = tid; i < D; i += 1024) dP[(size_t)prow * D + i] = 0.f;
    return;
  }
  const int nch = D >> 3;
  const int nbl = 1024 / nch;
  const int c = tid % nch, bl = tid / nch;
  if (bl < nbl) {
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    for (int b = bl; b < B; b += nbl) {
      float v[8];
      unpack8(*reinterpret_cast<const u32x4*>(dz + ((size_t)b * S + s) * D + c * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[bl * D + c * 8 + e] = a[e];
  }
  __syncthreads();
  for (int i = tid; i < D; i += 1024) {
    float t = 0.f;
    for (int k = 0; k < nbl; ++k) t += red[k * D + i];
    dP[(size_t)prow * D + i] = t;
  }
}

}  // namespace

__global__ __launch_bounds__(256) void pack_features_kernel(const KmbPackList l, int F, float* __restrict__ dst) {
  const int i = (int)blockIdx.y;
  const float* __restrict__ src = l.src[i];
  const size_t n = (size_t)l.rows[i] * F;
  float* __restrict__ out = dst + (size_t)l.off[i] * F;
  const size_t t0 = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  if ((F & 3) == 0 && (((uintptr_t)src | (uintptr_t)out) & 15) == 0) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    for (size_t k = t0; k < (n >> 2); k += step) o4[k] = s4[k];
  } else {
    for (size_t k = t0; k < n; k += step) out[k] = src[k];
  }
}

hipError_t kmb_img_rowmap_launch(const int64_t* ids, const int32_t* feat_off, int B, int S, int64_t img_feat_id,
                                 int64_t cls_id, int32_t* img_src, int32_t* status, hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(img_rowmap_kernel, dim3(B), dim3(64), 0, stream, ids, feat_off, S, img_feat_id, cls_id, img_src, status);
  return hipGetLastError();
}

// The reference hands the region features over as a Python LIST of per-sample [R_i, F] tensors (src/data/collation.py:73-76,
// src/model/modules.py:24-41 concatenates them): one launch gathers up to KMB_PACK_MAX of them -- separate device allocations --
// into the packed [Ntot, F] buffer the engine reads (sample i's rows at row offset off[i]).  torch.cat did this as one batched
// kernel PLUS ~one blit per tensor on this stack (rocprofv3: 70 __amd_rocclr_copyBuffer launches in front of every generate of
// 64 samples, 0.3 ms of a 9.9 ms generate).  16-byte pieces when F * 4 is a multiple of 16 (F = 2052: yes), else floats.
hipError_t kmb_pack_features_launch(const KmbPackList& l, int feat_dim, float* dst, hipStream_t stream) {
  if (l.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_features_kernel, dim3(8, l.n), dim3(256), 0, stream, l, feat_dim, dst);
  return hipGetLastError();
}

hipError_t kmb_cast_pad_launch(const float* x, int N, int Fin, bf16_t* y, int Fpad, hipStream_t stream) {
  if (N <= 0) return hipSuccess;
  const size_t total = (size_t)N * (Fpad >> 3);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cast_pad_kernel, dim3(blocks), dim3(256), 0, stream, x, N, Fin, y, Fpad);
  return hipGetLastError();
}

hipError_t kmb_embed_ln_fwd_launch(const int64_t* ids, const int32_t* img_src, const float* E, const float* img_emb,
                                   const float* P, int pos_base, int S, float scale, const float* gamma,
                                   const float* beta, bf16_t* z, bf16_t* y, float* mean, float* rstd, int M, int D,
                                   float eps, KmbDrop drop, hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  if ((D & 7) || D > 2048) return hipErrorInvalidValue;
  if (((uintptr_t)E & 15) || ((uintptr_t)P & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15) || ((uintptr_t)img_emb & 15))
    return hipErrorInvalidValue;
  dim3 grid((M + 3) / 4), block(256);
  if (D <= 512)
    hipLaunchKernelGGL((embed_ln_fwd_kernel<1>), grid, block, 0, stream, ids, img_src, E, img_emb, P, pos_base, S, scale, gamma, beta, z, y, mean, rstd, M, D, eps, drop);
  else if (D <= 1024)
    hipLaunchKernelGGL((embed_ln_fwd_kernel<2>), grid, block, 0, stream, ids, img_src, E, img_emb, P, pos_base, S, scale, gamma, beta, z, y, mean, rstd, M, D, eps, drop);
  else
    hipLaunchKernelGGL((embed_ln_fwd_kernel<4>), grid, block, 0, stream, ids, img_src, E, img_emb, P, pos_base, S, scale, gamma, beta, z, y, mean, rstd, M, D, eps, drop);
  return hipGetLastError();
}

hipError_t kmb_embed_bwd_launch(const bf16_t* dz, const int64_t* ids, const int32_t* img_src, float scale,
                                float* dE, bf16_t* dimg, int64_t pad_id, int M, int D, hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, dz, ids, img_src, scale, dE, dimg, pad_id, M, D);
  return hipGetLastError();
}

hipError_t kmb_pos_bwd_launch(const bf16_t* dz, int B, int S, int D, float* dP, int pos_base, int P_rows,
                              hipStream_t stream) {
  if ((D & 7) || D > 8192) return hipErrorInvalidValue;
  const size_t lds = (size_t)(1024 / (D >> 3)) * D * sizeof(float);   // <= 32 KB
  hipLaunchKernelGGL(pos_bwd_kernel, dim3(P_rows), dim3(1024), lds, stream, dz, B, S, D, dP, pos_base);
  return hipGetLastError();
}
