// Fused blocks of a KV-cached decode step (generation: R = batch x beams rows, one new token per row).
//
// A decode step of a BartDecoderLayer (reference src/model/modules.py: DecoderLayer / SelfAttention with use_cache,
// transformers 3.0.2 modeling_bart.py:386-466) is six dependent projections with R rows each (R = 320 at the benchmark
// setting: 64 x 5 beams).  Every row is independent of every other row; only the weights are shared.  As separate
// GEMM / attention / LayerNorm launches that was 11 launches per layer, each a few microseconds of work behind a launch
// latency.  Here a layer is six launches, cut exactly where a projection needs ALL columns of the previous one:
//
//   kind 1  [LayerNorm] -> q|k|v projection of ONE head -> append k, v to the cache -> attention over the cache
//   kind 0  output projection + bias + residual                                   -> pre-LayerNorm sum (bf16)
//   kind 2  LayerNorm -> q projection of ONE head -> attention over the cached encoder keys / values
//   kind 0  output projection + bias + residual
//   kind 0  LayerNorm -> fc1 + bias + GeLU
//   kind 0  fc2 + bias + residual
//
// A workgroup owns 16 rows (one MFMA row tile) x 64 output columns (kind 0) or one head (kinds 1, 2).  The LayerNorm
// of its 16 input rows is recomputed by every workgroup that needs them (12-48 times 16 x 768 elements: noise) instead
// of being a launch of its own; the normalised rows are kept in LDS as the MFMA activation operand and written to
// memory once (by the workgroups of column block 0) because the next projection adds them as its residual.
//
// Weights go global -> registers directly as MFMA fragments (no LDS staging: a workgroup reads each weight element
// once), 32 contiguous bytes per lane per 64-deep K chunk: the 16 x 16 x 32 MFMA sums over its 32 K slots in no
// particular order, so lane group g feeds K elements g*16 .. g*16+7 to the first MFMA of a chunk and g*16+8 .. g*16+15
// to the second, for the weight and the activation fragment alike.  All fragments of a 768-deep K block (96 VGPRs per
// 16-column tile) are requested before the LayerNorm prologue runs, so the weight latency hides behind it.
//
// Numerics follow the unfused path: q, k, v, attention output, GeLU output and the pre-LayerNorm sums are rounded to
// bf16 where that path stores them; sums are fp32.
#include "common.h"
#include "kernels.h"
#include <math.h>

namespace {

constexpr int RT = 16;      // rows per workgroup
constexpr int KBLK = 768;   // K block whose weight fragments are in flight together
constexpr int HD = 64;

struct WBlock { u32x4 w[24]; };   // one 16-column tile x 768 K

__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// weight rows n0 .. n0+15, K range [k0, k0 + 768)
__device__ __forceinline__ void load_wblock(WBlock& f, const bf16_t* __restrict__ W, int ldw, int n0, int k0, int lane) {
  const int r = lane & 15, g = lane >> 4;
  const bf16_t* p = W + (size_t)(n0 + r) * ldw + k0 + g * 16;
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    f.w[2 * c] = *reinterpret_cast<const u32x4*>(p + c * 64);
    f.w[2 * c + 1] = *reinterpret_cast<const u32x4*>(p + c * 64 + 8);
  }
}

// acc (C^T tile: lane (r, g) holds columns n0 + 4g .. 4g+3 of row r) += W block x activation rows in LDS
__device__ __forceinline__ void mma_wblock(f32x4& acc, const WBlock& f, const char* lds_a, int a_stride, int k0, int lane) {
  const int r = lane & 15, g = lane >> 4;
  const char* pa = lds_a + r * a_stride + (k0 + g * 16) * 2;
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(pa + c * 128);
    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(pa + c * 128 + 16);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(f.w[2 * c]), a0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(f.w[2 * c + 1]), a1, acc, 0, 0, 0);
  }
}

// 16 input rows -> LDS (row stride a_stride bytes), LayerNorm'ed on the way when gamma != null (same arithmetic as
// ln_fwd_kernel).  Wave w stages rows 4w .. 4w+3; `keep` != null: the normalised rows are also written to memory.
template <int NCH>   // 16-byte chunks per lane: K / 8 / 64 rounded up
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ in, int ld_in, int row0, int R, int K,
                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                           bf16_t* __restrict__ keep, char* lds_a, int a_stride, int wave, int lane) {
  const int nch = K >> 3;
  u32x4 raw[4][NCH];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + wave * 4 + i;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      raw[i][j] = (row < R && c < nch) ? *reinterpret_cast<const u32x4*>(in + (size_t)row * ld_in + c * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lr = wave * 4 + i, row = row0 + lr;
    if (gamma == nullptr) {
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = lane + 64 * j;
        if (c < nch) *reinterpret_cast<u32x4*>(lds_a + lr * a_stride + c * 16) = raw[i][j];
      }
      continue;
    }
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      unpack8(raw[i][j], v[j]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[j][e];   // chunks past K are zero
    }
    const float mu = wave_sum(s) / (float)K;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      if (lane + 64 * j < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float dlt = v[j][e] - mu; q += dlt * dlt; }
      }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)K + eps);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + c * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c * 8), b1 = *reinterpret_cast<const f32x4*>(beta + c * 8 + 4);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (v[j][e] - mu) * rs * g0[e] + b0[e];
          o[4 + e] = (v[j][4 + e] - mu) * rs * g1[e] + b1[e];
        }
        const u32x4 pk = pack8(o);
        *reinterpret_cast<u32x4*>(lds_a + lr * a_stride + c * 16) = pk;
        if (keep != nullptr && row < R) *reinterpret_cast<u32x4*>(keep + (size_t)row * K + c * 8) = pk;
      }
    }
  }
}

__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 16));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
  return v;
}

// ------------------------------------------------------------------------------------------ kind 0: projection
// grid (N / 64, row tiles); wave w owns columns n0 + 16w .. +15
template <int NCH>
__global__ __launch_bounds__(256) void decode_proj_kernel(const KmbDecodeBlock p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = blockIdx.y * RT, n0 = blockIdx.x * 64 + wave * 16;
  const int a_stride = (p.K + 8) * 2;
  const int nblk = p.K / KBLK;
  WBlock wb[2];
  load_wblock(wb[0], p.W, p.K, n0, 0, lane);
  stage_rows<NCH>(p.in, p.ld_in, row0, p.R, p.K, p.gamma, p.beta, p.eps, blockIdx.x == 0 ? p.ln_out : nullptr, smem, a_stride,
                  wave, lane);
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < nblk; b += 2) {   // two K blocks per trip: the register arrays keep compile-time indices
    if (b + 1 < nblk) load_wblock(wb[1], p.W, p.K, n0, (b + 1) * KBLK, lane);
    mma_wblock(acc, wb[0], smem, a_stride, b * KBLK, lane);
    if (b + 1 < nblk) {
      if (b + 2 < nblk) load_wblock(wb[0], p.W, p.K, n0, (b + 2) * KBLK, lane);
      mma_wblock(acc, wb[1], smem, a_stride, (b + 1) * KBLK, lane);
    }
  }
  const int r = lane & 15, g = lane >> 4;
  const int row = row0 + r, col = n0 + g * 4;
  if (row >= p.R) return;
  const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + col);
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = acc[e] + bias[e];
  if (p.act == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
  }
  if (p.residual != nullptr) {
    const uint2 rr = *reinterpret_cast<const uint2*>(p.residual + (size_t)row * p.ld_res + col);
    v[0] += lo_bf(rr.x); v[1] += hi_bf(rr.x); v[2] += lo_bf(rr.y); v[3] += hi_bf(rr.y);
  }
  *reinterpret_cast<uint2*>(p.out + (size_t)row * p.ld_out + col) = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
}

// ------------------------------------------------------------------------------------------ kinds 1, 2: attention
// grid (H, row tiles).  SELF: W rows [q | k | v] (3 x H x 64), wave w owns 16-column tiles 3w .. 3w+2 of the head's
// 192 columns; the new key / value row goes to the cache at position Tk - 1.  Cross: W rows are the q rows only, wave w
// owns tile w; keys / values are the cached projections of the encoder output of the row's batch item.
// Attention: 16 lanes per row -- lane s scores keys s, s+16, ... and then owns output elements 4s .. 4s+3.
template <bool SELF, int NCH>
__global__ __launch_bounds__(256) void decode_attn_kernel(const KmbDecodeBlock p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTW = SELF ? 3 : 1;          // 16-column tiles per wave
  constexpr int QW = SELF ? 3 * HD : HD;     // projected columns of the head
  constexpr int QS = (QW + 8) * 2;           // LDS row stride of the projected tile
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = blockIdx.x, row0 = blockIdx.y * RT;
  const int d = p.H * HD;
  const int a_stride = (p.K + 8) * 2;
  char* const lds_q = smem + RT * a_stride;
  float* const sc = reinterpret_cast<float*>(lds_q + RT * QS);   // [16][Tk] scores
  // weight row of tile t (0 .. 11 | 0 .. 3) of this head: part (q | k | v) * d + h * 64 + (t % 4) * 16
  auto tile_row = [&](int t) { return (t >> 2) * d + h * HD + (t & 3) * 16; };
  WBlock wb[2];
  f32x4 acc[NTW];
  load_wblock(wb[0], p.W, p.K, tile_row(wave * NTW), 0, lane);
  stage_rows<NCH>(p.in, p.ld_in, row0, p.R, p.K, p.gamma, p.beta, p.eps, h == 0 ? p.ln_out : nullptr, smem, a_stride, wave,
                  lane);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t + 1 < NTW) load_wblock(wb[(t + 1) & 1], p.W, p.K, tile_row(wave * NTW + t + 1), 0, lane);
    mma_wblock(acc[t], wb[t & 1], smem, a_stride, 0, lane);
  }
  {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int tile = wave * NTW + t, part = tile >> 2;
      const int col = (tile & 3) * 16 + g * 4;           // within the head
      const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + part * d + h * HD + col);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[t][e] + bias[e];
      if (part == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.q_scale;
      }
      const uint2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      *reinterpret_cast<uint2*>(lds_q + r * QS + (part * HD + col) * 2) = pk;
      if (SELF && part > 0 && row0 + r < p.R) {   // append to the cache
        bf16_t* dst = (part == 1 ? p.Kc : p.Vc) + ((size_t)(row0 + r) * p.Tmax + (p.Tk - 1)) * p.ldc + h * HD + col;
        *reinterpret_cast<uint2*>(dst) = pk;
      }
    }
  }
  __syncthreads();
  // ---- attention ----
  const int lr = threadIdx.x >> 4, s = threadIdx.x & 15;
  const int row = row0 + lr;
  if (row >= p.R) return;   // no barrier below
  const int crow = p.kv_row != nullptr ? p.kv_row[row] : row;
  const bf16_t* Kc = p.Kc + (size_t)crow * p.Tmax * p.ldc + h * HD;
  const bf16_t* Vc = p.Vc + (size_t)crow * p.Tmax * p.ldc + h * HD;
  float q[HD];
#pragma unroll
  for (int c = 0; c < 8; ++c) unpack8(*reinterpret_cast<const u32x4*>(lds_q + lr * QS + c * 16), q + c * 8);
  const int t_new = SELF ? p.Tk - 1 : -1;
  float* const my = sc + (size_t)lr * p.Tk;
  float mx = -INFINITY;
  for (int t = s; t < p.Tk; t += 16) {
    float dot = 0.f;
    if (t == t_new) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float k8[8];
        unpack8(*reinterpret_cast<const u32x4*>(lds_q + lr * QS + HD * 2 + c * 16), k8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dot += q[c * 8 + e] * k8[e];
      }
    } else {
      const bf16_t* krow = Kc + (size_t)t * p.ldc;
      u32x4 kr[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) kr[c] = *reinterpret_cast<const u32x4*>(krow + c * 8);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float k8[8];
        unpack8(kr[c], k8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dot += q[c * 8 + e] * k8[e];
      }
    }
    if (p.key_mask != nullptr && p.key_mask[(size_t)crow * p.mask_ld + t] == 0) dot = -INFINITY;
    my[t] = dot;
    mx = fmaxf(mx, dot);
  }
  mx = group16_max(mx);
  float l = 0.f;
  for (int t = s; t < p.Tk; t += 16) {
    const float e = (mx == -INFINITY) ? 0.f : __expf(my[t] - mx);
    my[t] = e;
    l += e;
  }
  l = group16_sum(l);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the 16 lanes of a row are in one wave
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float a[4][4];   // four partial sums (independent chains) x four output elements
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) a[i][e] = 0.f;
  const int Tc = SELF ? p.Tk - 1 : p.Tk;   // rows that live in the cache
  int t = 0;
  for (; t + 4 <= Tc; t += 4) {
    uint2 vv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vv[i] = *reinterpret_cast<const uint2*>(Vc + (size_t)(t + i) * p.ldc + s * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float w = my[t + i];
      a[i][0] += w * lo_bf(vv[i].x); a[i][1] += w * hi_bf(vv[i].x); a[i][2] += w * lo_bf(vv[i].y); a[i][3] += w * hi_bf(vv[i].y);
    }
  }
  for (; t < Tc; ++t) {
    const uint2 vv = *reinterpret_cast<const uint2*>(Vc + (size_t)t * p.ldc + s * 4);
    const float w = my[t];
    a[0][0] += w * lo_bf(vv.x); a[0][1] += w * hi_bf(vv.x); a[0][2] += w * lo_bf(vv.y); a[0][3] += w * hi_bf(vv.y);
  }
  if (SELF) {
    const uint2 vv = *reinterpret_cast<const uint2*>(lds_q + lr * QS + (2 * HD + s * 4) * 2);
    const float w = my[t_new];
    a[1][0] += w * lo_bf(vv.x); a[1][1] += w * hi_bf(vv.x); a[1][2] += w * lo_bf(vv.y); a[1][3] += w * hi_bf(vv.y);
  }
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = ((a[0][e] + a[1][e]) + (a[2][e] + a[3][e])) * inv;
  *reinterpret_cast<uint2*>(p.out + (size_t)row * p.ld_out + h * HD + s * 4) = uint2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
}

template <typename F>
hipError_t set_lds(F* fn, size_t lds) {
  return hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

}  // namespace

const char* kmb_decode_block_check(const KmbDecodeBlock& p) {
  if (p.R <= 0) return "decode block: no rows";
  if (p.K <= 0 || (p.K % KBLK) != 0 || p.K > 4 * KBLK) return "decode block: K must be 768, 1536, 2304 or 3072";
  if (!p.in || !p.W || !p.bias || !p.out) return "decode block: missing tensor";
  if ((p.ld_in & 7) || (p.ld_out & 3)) return "decode block: row strides";
  if (((uintptr_t)p.in & 15) || ((uintptr_t)p.W & 15) || ((uintptr_t)p.bias & 15) || ((uintptr_t)p.out & 7))
    return "decode block: alignment";
  if ((p.gamma == nullptr) != (p.beta == nullptr)) return "decode block: gamma and beta come together";
  if (p.ln_out && (!p.gamma || ((uintptr_t)p.ln_out & 15))) return "decode block: ln_out needs the LayerNorm";
  if (p.kind == 0) {
    if (p.N <= 0 || (p.N & 63)) return "decode block: N must be a multiple of 64";
    if (p.residual && ((p.ld_res & 3) || ((uintptr_t)p.residual & 7))) return "decode block: residual alignment";
    if (p.act != 0 && p.act != 1) return "decode block: act";
  } else if (p.kind == 1 || p.kind == 2) {
    if (p.H <= 0 || p.Tk <= 0 || p.Tk > p.Tmax || !p.Kc || !p.Vc) return "decode block: attention arguments";
    if (p.K != KBLK) return "decode block: attention kinds take K = 768";
    if (p.N != (p.kind == 1 ? 3 : 1) * p.H * HD) return "decode block: N must be the q|k|v (self) or q (cross) rows of H heads";
    if ((p.ldc & 7) || ((uintptr_t)p.Kc & 15) || ((uintptr_t)p.Vc & 15)) return "decode block: cache alignment";
    if (p.kind == 1 && (p.kv_row || p.key_mask)) return "decode block: the self-attention cache is per row and unmasked";
    if ((size_t)RT * p.Tk * 4 > 96 * 1024) return "decode block: Tk too large for the score tile";
  } else {
    return "decode block: kind";
  }
  return nullptr;
}

hipError_t kmb_decode_block_launch(const KmbDecodeBlock& p, hipStream_t stream) {
  const int tiles = (p.R + RT - 1) / RT;
  const size_t a_bytes = (size_t)RT * (p.K + 8) * 2;
  const int nch = (p.K / 8 + 63) / 64;
  hipError_t e = hipSuccess;
  if (p.kind == 0) {
    static size_t set[7] = {0, 0, 0, 0, 0, 0, 0};
#define KMB_PROJ(NCH)                                                                                       \
  do {                                                                                                      \
    if (a_bytes > set[NCH]) { e = set_lds(decode_proj_kernel<NCH>, a_bytes); if (e != hipSuccess) return e; set[NCH] = a_bytes; } \
    hipLaunchKernelGGL((decode_proj_kernel<NCH>), dim3(p.N / 64, tiles), dim3(256), a_bytes, stream, p);    \
  } while (0)
    if (nch <= 2) KMB_PROJ(2);
    else if (nch <= 4) KMB_PROJ(4);
    else KMB_PROJ(6);
#undef KMB_PROJ
    return hipGetLastError();
  }
  const bool self = p.kind == 1;
  const size_t lds = a_bytes + (size_t)RT * ((self ? 3 * HD : HD) + 8) * 2 + (size_t)RT * p.Tk * sizeof(float);
  static size_t set_s = 0, set_c = 0;
  if (self) {
    if (lds > set_s) { e = set_lds(decode_attn_kernel<true, 2>, lds); if (e != hipSuccess) return e; set_s = lds; }
    hipLaunchKernelGGL((decode_attn_kernel<true, 2>), dim3(p.H, tiles), dim3(256), lds, stream, p);
  } else {
    if (lds > set_c) { e = set_lds(decode_attn_kernel<false, 2>, lds); if (e != hipSuccess) return e; set_c = lds; }
    hipLaunchKernelGGL((decode_attn_kernel<false, 2>), dim3(p.H, tiles), dim3(256), lds, stream, p);
  }
  return hipGetLastError();
}
