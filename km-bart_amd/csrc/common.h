// Shared device helpers for the KM-BART HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits; arithmetic is always done in f32

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (8 bf16 = 4 VGPR)
typedef __attribute__((ext_vector_type(4))) short s16x4;    // ds_read_b64_tr_b16 result
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4; // one 16-byte chunk

#define KMB_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even, NaN stays NaN: the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950
// (one instruction per two elements; the integer-arithmetic form costs ~6 VALU each and dominated the epilogues)
typedef __attribute__((ext_vector_type(2))) __bf16 kmb_bf16x2;
typedef __attribute__((ext_vector_type(2))) float kmb_f32x2;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const kmb_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, kmb_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float lo_bf(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_bf(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

__device__ __forceinline__ void unpack8(const u32x4& c, float* f) {
  f[0] = lo_bf(c[0]); f[1] = hi_bf(c[0]); f[2] = lo_bf(c[1]); f[3] = hi_bf(c[1]);
  f[4] = lo_bf(c[2]); f[5] = hi_bf(c[2]); f[6] = lo_bf(c[3]); f[7] = hi_bf(c[3]);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 c;
  c[0] = pack2bf(f[0], f[1]); c[1] = pack2bf(f[2], f[3]);
  c[2] = pack2bf(f[4], f[5]); c[3] = pack2bf(f[6], f[7]);
  return c;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf-form GeLU, the reference's "activation_function": "gelu" (config/vcg_base.json:3).
// erf via Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 round-off level): one v_rcp + one v_exp
// instead of libm's branchy erff -- the GeLU epilogue of a K=768 GEMM costs as much as its main loop otherwise.
// Returns cdf = Phi(x) and e = exp(-x^2 / 2) (shared with the derivative).
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& e) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));      // v_rcp_f32 (1 ulp), no division sequence
  e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);              // exp(-x^2 / 2) as one v_exp_f32
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float h = 0.5f * poly * t * e;                                   // 0.5 * (1 - erf(|x| / sqrt 2))
  cdf = x >= 0.f ? 1.0f - h : h;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, e;
  gelu_parts(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, e;
  gelu_parts(x, cdf, e);
  return cdf + x * 0.39894228040143268f * e;
}

// Two elements at a time: the same arithmetic written on float2 so that the multiplies / FMAs lower to v_pk_mul_f32 /
// v_pk_fma_f32 (one instruction per pair); only v_rcp_f32 / v_exp_f32 stay scalar.  Bit-identical to the scalar forms
// (every operation is the same IEEE operation in the same order; -ffp-contract=off).  The GeLU epilogue of the K = 768
// FFN GEMM is pure VALU time with the matrix pipe idle: 42 us on top of a 99 us GEMM with the scalar form.
__device__ __forceinline__ void gelu_parts2(kmb_f32x2 x, kmb_f32x2& cdf, kmb_f32x2& e) {
  const kmb_f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const kmb_f32x2 z = ax * 0.70710678118654752f;
  const kmb_f32x2 den = __builtin_elementwise_fma(kmb_f32x2{0.3275911f, 0.3275911f}, z, kmb_f32x2{1.0f, 1.0f});
  const kmb_f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const kmb_f32x2 arg = z * z * -1.4426950408889634f;
  e = kmb_f32x2{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
  kmb_f32x2 poly = __builtin_elementwise_fma(kmb_f32x2{1.061405429f, 1.061405429f}, t, kmb_f32x2{-1.453152027f, -1.453152027f});
  poly = __builtin_elementwise_fma(poly, t, kmb_f32x2{1.421413741f, 1.421413741f});
  poly = __builtin_elementwise_fma(poly, t, kmb_f32x2{-0.284496736f, -0.284496736f});
  poly = __builtin_elementwise_fma(poly, t, kmb_f32x2{0.254829592f, 0.254829592f});
  const kmb_f32x2 h = 0.5f * poly * t * e;
  cdf = kmb_f32x2{x[0] >= 0.f ? 1.0f - h[0] : h[0], x[1] >= 0.f ? 1.0f - h[1] : h[1]};
}
__device__ __forceinline__ kmb_f32x2 gelu2(kmb_f32x2 x) {
  kmb_f32x2 cdf, e;
  gelu_parts2(x, cdf, e);
  return x * cdf;
}
// GeLU and its derivative from ONE evaluation of (cdf, e): the forward FFN epilogue stores the derivative (bf16) instead
// of the pre-activation, so the backward epilogue of the fc2 data gradient is a single multiply -- the erf / exp work of
// GeLU'(u) is done once, where cdf and e are already in registers, instead of a second time per element in backward.
__device__ __forceinline__ void gelu_both2(kmb_f32x2 x, kmb_f32x2& y, kmb_f32x2& dy) {
  kmb_f32x2 cdf, e;
  gelu_parts2(x, cdf, e);
  y = x * cdf;
  dy = cdf + x * 0.39894228040143268f * e;
}
__device__ __forceinline__ kmb_f32x2 gelu_grad2(kmb_f32x2 x) {
  kmb_f32x2 cdf, e;
  gelu_parts2(x, cdf, e);
  return cdf + x * 0.39894228040143268f * e;
}

// ---- split-K slabs: a += slab[1] + slab[2] + ... (src points at this thread's element of slab 0), in slice order ----
// U slabs' loads are in flight together: the plain `for (s) a += load(s)` loop is not unrolled by hipcc (runtime trip count)
// and compiles to load, s_waitcnt vmcnt(0), add -- one dependent memory round trip per slab.  Same additions in the same
// order: identical bits.
template <int U>
__device__ __forceinline__ void add_slabs(f32x4& a, const float* __restrict__ src, size_t stride, int nslabs) {
  int s = 1;
  for (; s + U <= nslabs; s += U) {
    f32x4 b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) b[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(s + u) * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) { a[0] += b[u][0]; a[1] += b[u][1]; a[2] += b[u][2]; a[3] += b[u][3]; }
  }
  for (; s < nslabs; ++s) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + (size_t)s * stride);
    a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
  }
}
// the same for two adjacent vectors (eight columns)
template <int U>
__device__ __forceinline__ void add_slabs2(f32x4& lo, f32x4& hi, const float* __restrict__ src, size_t stride, int nslabs) {
  int s = 1;
  for (; s + U <= nslabs; s += U) {
    f32x4 bl[U], bh[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bl[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(s + u) * stride);
      bh[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(s + u) * stride + 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      lo[0] += bl[u][0]; lo[1] += bl[u][1]; lo[2] += bl[u][2]; lo[3] += bl[u][3];
      hi[0] += bh[u][0]; hi[1] += bh[u][1]; hi[2] += bh[u][2]; hi[3] += bh[u][3];
    }
  }
  for (; s < nslabs; ++s) {
    const f32x4 l2 = *reinterpret_cast<const f32x4*>(src + (size_t)s * stride);
    const f32x4 h2 = *reinterpret_cast<const f32x4*>(src + (size_t)s * stride + 4);
    lo[0] += l2[0]; lo[1] += l2[1]; lo[2] += l2[2]; lo[3] += l2[3];
    hi[0] += h2[0]; hi[1] += h2[1]; hi[2] += h2[2]; hi[3] += h2[3];
  }
}

// ---- dropout: counter-based keep decision, identical in forward epilogues and backward ----
// keep(row, col) depends only on (site_seed, row, col); site_seed = mix(seed, step, site) on host.
__device__ __forceinline__ uint32_t kmb_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// the same decision for a PAIR of columns (col even, col + 1) from the pieces a row loop can keep: rowterm = row * 0x9E3779B1u (wave-uniform),
// colterm = (col >> 1) * 0x85EBCA77u + 0x165667B1u (loop-invariant per lane) -- one hash, no multiply outside it
__device__ __forceinline__ uint32_t drop_colterm(uint32_t col) { return (col >> 1) * 0x85EBCA77u + 0x165667B1u; }
__device__ __forceinline__ void drop_keep_pair(uint32_t site_seed, uint32_t rowterm, uint32_t colterm, uint32_t thr16, bool& k0, bool& k1) {
  const uint32_t h = kmb_hash32(rowterm ^ colterm ^ site_seed);
  k0 = (h & 0xffffu) >= thr16;
  k1 = (h >> 16) >= thr16;
}
__device__ __forceinline__ bool drop_keep(uint32_t site_seed, uint32_t row, uint32_t col, uint32_t thr16) {
  // one 32-bit hash per PAIR of columns (16 random bits each): callers that walk consecutive columns share it
  const uint32_t h = kmb_hash32((row * 0x9E3779B1u) ^ ((col >> 1) * 0x85EBCA77u + 0x165667B1u) ^ site_seed);
  const uint32_t bits = (col & 1u) ? (h >> 16) : (h & 0xffffu);
  return bits >= thr16;  // P(drop) = thr16 / 65536
}
