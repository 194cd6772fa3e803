// LayerNorm forward / backward, column-sum (bias gradient) and partial reducers.
// All of these are HBM-bound row kernels: one 64-lane wave per row, 16-byte accesses,
// fp32 statistics.  D must be a multiple of 8 and <= 2048.
#include "common.h"
#include "kernels.h"

namespace {

template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ z, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int D,
                                                     float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nch = D >> 3;
  // every load of the row and of its parameters is issued before the first use, from clamped addresses and outside any
  // branch (a load inside `if (c < nch) { load; use }` gets its own s_waitcnt vmcnt(0): NCH + 1 dependent round trips per
  // row, which is what a 320-row decode launch is made of)
  u32x4 raw[NCH];
  f32x4 gv[NCH][2], bv[NCH][2];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i < nch ? lane + 64 * i : 0;
    raw[i] = *reinterpret_cast<const u32x4*>(z + (size_t)row * D + c * 8);
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i < nch ? lane + 64 * i : 0;
    gv[i][0] = *reinterpret_cast<const f32x4*>(gamma + c * 8); gv[i][1] = *reinterpret_cast<const f32x4*>(gamma + c * 8 + 4);
    bv[i][0] = *reinterpret_cast<const f32x4*>(beta + c * 8); bv[i][1] = *reinterpret_cast<const f32x4*>(beta + c * 8 + 4);
  }
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (lane + 64 * i < nch) {
      unpack8(raw[i], v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (lane + 64 * i < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu) * rs * gv[i][e >> 2][e & 3] + bv[i][e >> 2][e & 3];
      *reinterpret_cast<u32x4*>(y + (size_t)row * D + c * 8) = pack8(o);
    }
  }
}

// Training shapes (tens of thousands of rows): the same row arithmetic as ln_fwd_kernel (same operations in the same order: same bits), but
// a wave keeps gamma / beta in registers and walks rows w, w + W, ... with the next row's load in flight.  ln_fwd_kernel gives every row its
// own wave, and that wave issues 4 * NCH parameter loads beside its NCH row loads: ten loads and four stores to move 3 KB at d = 768 -- the
// kernel sat at 4.1 TB/s with eight waves per SIMD (round 5: the count of memory instructions per wave is what such kernels are bound by,
// DESIGN.md section 4, attention study).
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_stream_kernel(const bf16_t* __restrict__ z, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int M, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int W = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = D >> 3;
  int cc[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) cc[i] = lane + 64 * i < nch ? lane + 64 * i : 0;
  u32x4 nxt[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) nxt[i] = *reinterpret_cast<const u32x4*>(z + (size_t)row * D + cc[i] * 8);
  f32x4 gv[NCH][2], bv[NCH][2];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    gv[i][0] = *reinterpret_cast<const f32x4*>(gamma + cc[i] * 8); gv[i][1] = *reinterpret_cast<const f32x4*>(gamma + cc[i] * 8 + 4);
    bv[i][0] = *reinterpret_cast<const f32x4*>(beta + cc[i] * 8); bv[i][1] = *reinterpret_cast<const f32x4*>(beta + cc[i] * 8 + 4);
  }
  for (; row < M; row += W) {
    u32x4 raw[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) raw[i] = nxt[i];
    if (row + W < M) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) nxt[i] = *reinterpret_cast<const u32x4*>(z + (size_t)(row + W) * D + cc[i] * 8);
    }
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (lane + 64 * i < nch) {
        unpack8(raw[i], v[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
      }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (lane + 64 * i < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
      }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu) * rs * gv[i][e >> 2][e & 3] + bv[i][e >> 2][e & 3];
        *reinterpret_cast<u32x4*>(y + (size_t)row * D + c * 8) = pack8(o);
      }
    }
  }
}

// Decode path: the split-K slabs of a residual projection are summed, bias and residual added, the sum rounded to
// bf16 (what the un-split GEMM epilogue stores and ln_fwd_kernel reads) and normalised -- one launch instead of the
// GEMM epilogue + LayerNorm pair, and the 320-row GEMM in front of it gets nslabs times as many workgroups.
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_slabs_kernel(const float* __restrict__ slabs, int nslabs, size_t stride,
                                                           const float* __restrict__ bias,
                                                           const bf16_t* __restrict__ residual, int ld_res,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           bf16_t* __restrict__ y, int M, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nch = D >> 3;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float* src = slabs + (size_t)row * D + c * 8;
      f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
      add_slabs2<4>(lo, hi, src, stride, nslabs);
      float rr[8];
      unpack8(*reinterpret_cast<const u32x4*>(residual + (size_t)row * ld_res + c * 8), rr);
      float t[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = (t[e] + (bias != nullptr ? bias[c * 8 + e] : 0.f)) + rr[e];
      unpack8(pack8(t), v[i]);   // bf16 rounding of the pre-LayerNorm sum, as the two-kernel path stores it
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (lane + 64 * i < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu) * rs * gamma[c * 8 + e] + beta[c * 8 + e];
      *reinterpret_cast<u32x4*>(y + (size_t)row * D + c * 8) = pack8(o);
    }
  }
}

// Each block owns rows [blockIdx.x * rows_per_block, ...); its 4 waves take them round-robin.
// One wave per row, lane l owns the 4-column groups l, l+64, ... (NQ of them: D = 768 is covered exactly by NQ = 3, no
// idle lanes), and the next row's dy / z are already in flight while the current one is reduced: the kernel is a pure
// stream (2 reads + 1-2 writes of M x D bf16) and a wave with one row in flight at a time is latency-bound.
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
template <int NQ>
__global__ __launch_bounds__(256, NQ <= 3 ? 4 : 1) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ z,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, bf16_t* __restrict__ dz,
                                                     bf16_t* __restrict__ out2, KmbDrop dy_drop, KmbDrop out2_drop,
                                                     float* __restrict__ partials, int M, int D, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [4][3][D]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nq = D >> 2;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(M, r_begin + rows_per_block);
  float* gam_s = red + 12 * D;   // gamma lives in LDS, three ds_read_b128 per row: 12 registers less = four waves per SIMD (see ln_bwd_rows_per_block)
  for (int idx = threadIdx.x; idx < D; idx += 256) gam_s[idx] = gamma[idx];
  __syncthreads();
  float dg[NQ][4], db[NQ][4], ds[NQ][4];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; ds[i][e] = 0.f; }
  }
  uint32_t colterm[NQ][2];   // dropout hash: the column part of each of this lane's column pairs (loop-invariant)
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    colterm[i][0] = drop_colterm((uint32_t)((lane + 64 * i) * 4));
    colterm[i][1] = drop_colterm((uint32_t)((lane + 64 * i) * 4 + 2));
  }
  const float inv_d = 1.0f / (float)D;
  u32x2 nd[NQ], nz[NQ];
  float nmu = 0.f, nrs = 0.f;
  int row = r_begin + wave;
  if (row < r_end) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int c = lane + 64 * i;
      if (c < nq) {
        nd[i] = *reinterpret_cast<const u32x2*>(dy + (size_t)row * D + c * 4);
        nz[i] = *reinterpret_cast<const u32x2*>(z + (size_t)row * D + c * 4);
      }
    }
    nmu = mean[row]; nrs = rstd[row];
  }
  for (; row < r_end; row += 4) {
    u32x2 cd[NQ], cz[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) { cd[i] = nd[i]; cz[i] = nz[i]; }
    const float mu = nmu, rs = nrs;
    const uint32_t rowterm = (uint32_t)row * 0x9E3779B1u;
    const int nrow = row + 4;
    if (nrow < r_end) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int c = lane + 64 * i;
        if (c < nq) {
          nd[i] = *reinterpret_cast<const u32x2*>(dy + (size_t)nrow * D + c * 4);
          nz[i] = *reinterpret_cast<const u32x2*>(z + (size_t)nrow * D + c * 4);
        }
      }
      nmu = mean[nrow]; nrs = rstd[nrow];
    }
    // (round 5) the element-wise arithmetic on float2: v_pk_add / v_pk_mul / v_pk_fma do two elements per instruction, and this kernel is bound by
    // its vector-ALU work (~40 instructions per element: 64 rows per SIMD at ~1 us each are the launch's 70 us), not by its memory instructions
    // (tools/experiments/ln_bwd_two_rows_per_wave.patch).  Written two-wide by hand: the SLP vectoriser is off for this file's sake (build.py).
    // The row sums run as (even columns, odd columns) pairs that are added at the end: a different summation order than before (last bits).
    float g[NQ][4], xh[NQ][4];
    kmb_f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
    const kmb_f32x2 mu2 = {mu, mu}, rs2 = {rs, rs};
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int c = lane + 64 * i;
      if (c < nq) {
        const float d4[4] = {lo_bf(cd[i][0]), hi_bf(cd[i][0]), lo_bf(cd[i][1]), hi_bf(cd[i][1])};
        const float z4[4] = {lo_bf(cz[i][0]), hi_bf(cz[i][0]), lo_bf(cz[i][1]), hi_bf(cz[i][1])};
        const f32x4 gq = *reinterpret_cast<const f32x4*>(gam_s + c * 4);
        const float gm4[4] = {gq[0], gq[1], gq[2], gq[3]};
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          kmb_f32x2 d = {d4[e], d4[e + 1]};
          if (dy_drop.thr16 != 0u) {
            bool ka, kb;
            drop_keep_pair(dy_drop.seed, rowterm, colterm[i][e >> 1], dy_drop.thr16, ka, kb);
            d[0] = ka ? d[0] * dy_drop.scale : 0.f;
            d[1] = kb ? d[1] * dy_drop.scale : 0.f;
          }
          const kmb_f32x2 x = (kmb_f32x2{z4[e], z4[e + 1]} - mu2) * rs2;
          xh[i][e] = x[0]; xh[i][e + 1] = x[1];
          kmb_f32x2 acc = {dg[i][e], dg[i][e + 1]};
          acc = __builtin_elementwise_fma(d, x, acc);
          dg[i][e] = acc[0]; dg[i][e + 1] = acc[1];
          db[i][e] += d[0]; db[i][e + 1] += d[1];
          const kmb_f32x2 gg = d * kmb_f32x2{gm4[e], gm4[e + 1]};
          g[i][e] = gg[0]; g[i][e + 1] = gg[1];
          s1v += gg;
          s2v = __builtin_elementwise_fma(gg, x, s2v);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[i][e] = 0.f; xh[i][e] = 0.f; }
      }
    }
    const float c1 = wave_sum(s1v[0] + s1v[1]) * inv_d;   // (a multiply by 1 / d, not a division sequence: last bit)
    const float c2 = wave_sum(s2v[0] + s2v[1]) * inv_d;
    const kmb_f32x2 c1v = {c1, c1}, nc2v = {-c2, -c2};
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int c = lane + 64 * i;
      if (c < nq) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const kmb_f32x2 t = __builtin_elementwise_fma(kmb_f32x2{xh[i][e], xh[i][e + 1]}, nc2v, kmb_f32x2{g[i][e], g[i][e + 1]} - c1v) * rs2;
          o[e] = t[0]; o[e + 1] = t[1];
        }
        u32x2 pk = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        *reinterpret_cast<u32x2*>(dz + (size_t)row * D + c * 4) = pk;
        if (out2 != nullptr) {
          if (out2_drop.thr16 != 0u) {
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              bool ka, kb;
              drop_keep_pair(out2_drop.seed, rowterm, colterm[i][e >> 1], out2_drop.thr16, ka, kb);
              o[e] = ka ? o[e] * out2_drop.scale : 0.f;
              o[e + 1] = kb ? o[e + 1] * out2_drop.scale : 0.f;
            }
          }
          pk[0] = pack2bf(o[0], o[1]); pk[1] = pack2bf(o[2], o[3]);
          *reinterpret_cast<u32x2*>(out2 + (size_t)row * D + c * 4) = pk;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ds[i][e] += o[e];  // column sums of the sub-layer gradient
      }
    }
  }
  // reduce the 4 waves' partial dgamma / dbeta through LDS, one partial row per block
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int c = lane + 64 * i;
    if (c < nq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[(wave * 3 + 0) * D + c * 4 + e] = dg[i][e];
        red[(wave * 3 + 1) * D + c * 4 + e] = db[i][e];
        red[(wave * 3 + 2) * D + c * 4 + e] = ds[i][e];
      }
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 3 * D; idx += 256) {
    const int which = idx / D, col = idx - which * D;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += red[(w * 3 + which) * D + col];
    partials[((size_t)blockIdx.x * 3 + which) * D + col] = s;
  }
}

// out[c] = beta*out[c] + sum_p partials[p*stride + c].  Two shapes of the same reduction:
//  (a) many partial rows, few columns (LayerNorm / bias gradients): 32 columns x 8 part-lanes per block
// (out2 != nullptr: columns [n, n + n2) of the partial rows go to out2 -- LayerNorm's dgamma | dbeta and the bias gradient
//  of the sub-layer in ONE launch)
template <int PL>  // part-lanes per block: 32 columns x PL part-lanes
__global__ __launch_bounds__(32 * PL) void reduce_parts_kernel(const float* __restrict__ partials, int nparts,
                                                               size_t stride, float* __restrict__ out, int n_all,
                                                               float* __restrict__ out2, int n1) {
  __shared__ float red[PL][33];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int n = n_all;
  float s = 0.f;
  if (c < n) {
#pragma unroll 8
    for (int p = pl; p < nparts; p += PL) s += partials[(size_t)p * stride + c];
  }
  red[pl][cl] = s;
  __syncthreads();
  if (pl == 0 && c < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < PL; ++k) t += red[k][cl];
    if (out2 != nullptr && c >= n1) out2[c - n1] = t;
    else out[c] = t;
  }
}
//  (b) few slabs, many elements (split-K weight gradients): float4 per thread
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int nslabs, size_t stride,
                                                           float* __restrict__ out, size_t n4, float beta) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 a = reinterpret_cast<const f32x4*>(slabs)[i];
    add_slabs<8>(a, slabs + 4 * i, stride, nslabs);
    if (beta != 0.f) {
      const f32x4 o = reinterpret_cast<const f32x4*>(out)[i];
      a[0] += beta * o[0]; a[1] += beta * o[1]; a[2] += beta * o[2]; a[3] += beta * o[3];
    }
    reinterpret_cast<f32x4*>(out)[i] = a;
  }
}

// the same sum stored as bf16 (a split-K data gradient): n % 8 == 0
__global__ __launch_bounds__(256) void reduce_slabs_bf16_kernel(const float* __restrict__ slabs, int nslabs, size_t stride,
                                                                bf16_t* __restrict__ out, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    f32x4 a = reinterpret_cast<const f32x4*>(slabs)[2 * i], b = reinterpret_cast<const f32x4*>(slabs)[2 * i + 1];
    add_slabs2<4>(a, b, slabs + 8 * i, stride, nslabs);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    reinterpret_cast<u32x4*>(out)[i] = pack8(v);
  }
}

// Split-K forward / data-gradient GEMMs of the small-batch regime (few output tiles, long K: M = 2048-4096 rows against
// 256 CUs): the slabs are summed in slice order and the linear layer's epilogue is applied here -- bias, q-scale, dropout
// (the same counter-based mask as the GEMM epilogue and the LayerNorm backward), residual -- then one bf16 store.
__global__ __launch_bounds__(256) void reduce_slabs_epi_kernel(const float* __restrict__ slabs, int nslabs, size_t stride,
                                                               const float* __restrict__ bias, float col_scale,
                                                               int col_scale_n, KmbDrop drop,
                                                               const bf16_t* __restrict__ residual, int ld_res,
                                                               bf16_t* __restrict__ out, int ld_out, int M, int N) {
  const int nch = N >> 3;
  const size_t total = (size_t)M * nch;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / nch), c = (int)(i % nch) * 8;
    const float* src = slabs + (size_t)row * N + c;
    f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    add_slabs2<4>(a, b, src, stride, nslabs);
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = (v[e] + (bias != nullptr ? bias[c + e] : 0.f)) * ((c + e) < col_scale_n ? col_scale : 1.f);
      if (drop.thr16 != 0u) v[e] = drop_keep(drop.seed, (uint32_t)row, (uint32_t)(c + e), drop.thr16) ? v[e] * drop.scale : 0.f;
    }
    if (residual != nullptr) {
      float rr[8];
      unpack8(*reinterpret_cast<const u32x4*>(residual + (size_t)row * ld_res + c), rr);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rr[e];
    }
    *reinterpret_cast<u32x4*>(out + (size_t)row * ld_out + c) = pack8(v);
  }
}

// grid (ceil(N/64), nparts); block 256 = 8 column chunks x 32 row lanes
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ X, int ld, int M, int N,
                                                     float* __restrict__ partials, int rows_per_part) {
  __shared__ float red[32][65];
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int col = blockIdx.x * 64 + tx * 8;
  const int r0 = blockIdx.y * rows_per_part;
  const int r1 = min(M, r0 + rows_per_part);
  float a[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = 0.f;
  if (col < N) {
    const int nvalid = min(8, N - col);
    for (int r = r0 + ty; r < r1; r += 32) {
      float v[8];
      if (nvalid == 8) {
        unpack8(*reinterpret_cast<const u32x4*>(X + (size_t)r * ld + col), v);
      } else {
        for (int e = 0; e < 8; ++e) v[e] = e < nvalid ? bf2f(X[(size_t)r * ld + col + e]) : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = a[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
#pragma unroll 8
    for (int y = 0; y < 32; ++y) s += red[y][threadIdx.x];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < N) partials[(size_t)blockIdx.y * N + c] = s;
  }
}

}  // namespace

hipError_t kmb_ln_fwd_launch(const bf16_t* z, const float* gamma, const float* beta, bf16_t* y, float* mean,
                             float* rstd, int M, int D, float eps, hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  if ((D & 7) || D > 2048) return hipErrorInvalidValue;
  if (((uintptr_t)z & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15) || ((uintptr_t)y & 15)) return hipErrorInvalidValue;
  dim3 grid((M + 3) / 4), block(256);
  if (M >= 8192 && D > 512 && D <= 1024) {   // rows streamed through resident waves, parameters in registers
    // 1280 workgroups = five per CU (80 registers allow six): 38.5 / 22.9 us for 65536 / 32768 rows from HBM; 768 ... 2048 measured within 4 % of that
    hipLaunchKernelGGL((ln_fwd_stream_kernel<2>), dim3(grid.x < 1280 ? grid.x : 1280), block, 0, stream, z, gamma, beta, y, mean, rstd, M, D, eps);
    return hipGetLastError();
  }
  if (D <= 512) hipLaunchKernelGGL((ln_fwd_kernel<1>), grid, block, 0, stream, z, gamma, beta, y, mean, rstd, M, D, eps);
  else if (D <= 1024) hipLaunchKernelGGL((ln_fwd_kernel<2>), grid, block, 0, stream, z, gamma, beta, y, mean, rstd, M, D, eps);
  else hipLaunchKernelGGL((ln_fwd_kernel<4>), grid, block, 0, stream, z, gamma, beta, y, mean, rstd, M, D, eps);
  return hipGetLastError();
}

hipError_t kmb_ln_fwd_slabs_launch(const float* slabs, int nslabs, size_t stride, const float* bias, const bf16_t* residual,
                                   int ld_res, const float* gamma, const float* beta, bf16_t* y, int M, int D, float eps,
                                   hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  if ((D & 7) || D > 1024 || nslabs < 1 || (stride & 3)) return hipErrorInvalidValue;
  dim3 grid((M + 3) / 4), block(256);
  if (D <= 512) hipLaunchKernelGGL((ln_fwd_slabs_kernel<1>), grid, block, 0, stream, slabs, nslabs, stride, bias, residual, ld_res, gamma, beta, y, M, D, eps);
  else hipLaunchKernelGGL((ln_fwd_slabs_kernel<2>), grid, block, 0, stream, slabs, nslabs, stride, bias, residual, ld_res, gamma, beta, y, M, D, eps);
  return hipGetLastError();
}

static int ln_bwd_rows_per_block(int M) {
  // <= 1024 blocks: FOUR per CU (16 waves).  The kernel keeps gamma in LDS instead of 12 registers per lane, which brings it to 128 registers
  // (launch bounds (256, 4), no spills) and its LDS to 13 * D floats = 39 KB at D = 768: four blocks fit both.  Round 5, tools/ln_bwd_time.py,
  // 65536 rows: 140 registers and 768 blocks (three per CU) 70.7 / 103.5 / 101.5 us (dz only / both masks / out2 mask), now 65.8 / 97.2 / 98.1;
  // 32768 rows 40.2 / 48.5 / 47.1 -> 40.5 / 46.6 / 44.0.  (Forcing 128 registers WITH gamma in registers spilled: 114-137 us.)
  int rpb = (M + 1023) / 1024;
  if (rpb < 4) rpb = 4;
  return rpb;
}
int kmb_ln_bwd_parts(int M) {
  const int rpb = ln_bwd_rows_per_block(M);
  return (M + rpb - 1) / rpb;
}

hipError_t kmb_ln_bwd_launch(const bf16_t* dy, const bf16_t* z, const float* mean, const float* rstd,
                             const float* gamma, bf16_t* dz, bf16_t* out2, KmbDrop dy_drop, KmbDrop out2_drop,
                             float* partials, int M, int D, hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  if ((D & 7) || D > 2048) return hipErrorInvalidValue;
  const int rpb = ln_bwd_rows_per_block(M);
  dim3 grid((M + rpb - 1) / rpb), block(256);
  const size_t lds = (size_t)(4 * 3 + 1) * D * sizeof(float);
  const int nq = (D / 4 + 63) / 64;
#define KMB_LN_BWD(NQ)                                                                                                                  \
  do {                                                                                                                                  \
    if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_kernel<NQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((ln_bwd_kernel<NQ>), grid, block, lds, stream, dy, z, mean, rstd, gamma, dz, out2, dy_drop, out2_drop, partials, M, D, rpb); \
  } while (0)
  if (nq <= 1) KMB_LN_BWD(1);
  else if (nq == 2) KMB_LN_BWD(2);
  else if (nq == 3) KMB_LN_BWD(3);
  else if (nq == 4) KMB_LN_BWD(4);
  else KMB_LN_BWD(8);
#undef KMB_LN_BWD
  return hipGetLastError();
}

hipError_t kmb_reduce_parts_launch(const float* partials, int nparts, int stride, float* out, int n,
                                   hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  // only n / 32 blocks exist (72 for a LayerNorm): with many partial rows the kernel is latency-bound, so give each
  // block 32 part-lanes (1024 threads) worth of loads in flight
  if (nparts >= 256)
    hipLaunchKernelGGL((reduce_parts_kernel<32>), dim3((n + 31) / 32), dim3(1024), 0, stream, partials, nparts, (size_t)stride, out, n, (float*)nullptr, n);
  else
    hipLaunchKernelGGL((reduce_parts_kernel<8>), dim3((n + 31) / 32), dim3(256), 0, stream, partials, nparts, (size_t)stride, out, n, (float*)nullptr, n);
  return hipGetLastError();
}

// the same with two destinations: columns [0, n1) -> out1, [n1, n1 + n2) -> out2
hipError_t kmb_reduce_parts2_launch(const float* partials, int nparts, int stride, float* out1, int n1, float* out2, int n2,
                                    hipStream_t stream) {
  const int n = n1 + n2;
  if (n <= 0) return hipSuccess;
  if (nparts >= 256)
    hipLaunchKernelGGL((reduce_parts_kernel<32>), dim3((n + 31) / 32), dim3(1024), 0, stream, partials, nparts, (size_t)stride, out1, n, out2, n1);
  else
    hipLaunchKernelGGL((reduce_parts_kernel<8>), dim3((n + 31) / 32), dim3(256), 0, stream, partials, nparts, (size_t)stride, out1, n, out2, n1);
  return hipGetLastError();
}

// out[i] = beta*out[i] + sum_s slabs[s*stride + i], n % 4 == 0, 16-byte aligned
hipError_t kmb_reduce_slabs_launch(const float* slabs, int nslabs, size_t stride, float* out, size_t n, float beta,
                                   hipStream_t stream) {
  if (n == 0) return hipSuccess;
  if ((n & 3) || (stride & 3)) return hipErrorInvalidValue;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slabs, nslabs, stride, out, n / 4, beta);
  return hipGetLastError();
}

hipError_t kmb_reduce_slabs_bf16_launch(const float* slabs, int nslabs, size_t stride, bf16_t* out, size_t n,
                                        hipStream_t stream) {
  if (n == 0) return hipSuccess;
  if ((n & 7) || (stride & 3) || ((uintptr_t)out & 15)) return hipErrorInvalidValue;
  size_t blocks = (n / 8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(reduce_slabs_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slabs, nslabs, stride, out, n / 8);
  return hipGetLastError();
}

hipError_t kmb_reduce_slabs_epi_launch(const float* slabs, int nslabs, size_t stride, const float* bias, float col_scale,
                                       int col_scale_n, KmbDrop drop, const bf16_t* residual, int ld_res, bf16_t* out,
                                       int ld_out, int M, int N, hipStream_t stream) {
  if (M <= 0 || N <= 0) return hipSuccess;
  if ((N & 7) || (stride & 3) || (ld_out & 7) || ((uintptr_t)out & 15) || (residual && ((ld_res & 7) || ((uintptr_t)residual & 15))))
    return hipErrorInvalidValue;
  size_t blocks = ((size_t)M * (N >> 3) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(reduce_slabs_epi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slabs, nslabs, stride, bias,
                     col_scale, col_scale_n, drop, residual, ld_res, out, ld_out, M, N);
  return hipGetLastError();
}

static int colsum_rows_per_part(int M) {
  int rpp = (M + 63) / 64;  // <= 64 parts
  if (rpp < 32) rpp = 32;
  return rpp;
}
int kmb_colsum_parts(int M) {
  const int rpp = colsum_rows_per_part(M);
  return (M + rpp - 1) / rpp;
}
hipError_t kmb_colsum_launch(const bf16_t* X, int ld, int M, int N, float* partials, hipStream_t stream) {
  if (M <= 0 || N <= 0) return hipSuccess;
  const int rpp = colsum_rows_per_part(M);
  dim3 grid((N + 63) / 64, (M + rpp - 1) / rpp), block(256);
  hipLaunchKernelGGL(colsum_kernel, grid, block, 0, stream, X, ld, M, N, partials, rpp);
  return hipGetLastError();
}
