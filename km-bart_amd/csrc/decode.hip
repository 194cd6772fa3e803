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
// A workgroup owns 16 rows (one MFMA row tile) x 64 or 128 output columns (kind 0) or one head (kinds 1, 2).  The
// LayerNorm of its 16 input rows is recomputed by every workgroup that needs them (12-48 times 16 x 768 elements: noise)
// instead of being a launch of its own; the normalised rows are kept in LDS as the MFMA activation operand and written
// to memory once (by the workgroups of column block 0) because the next projection adds them as its residual.
//
// Weights go global -> registers directly as MFMA fragments (no LDS staging: a workgroup reads each weight element
// once) from a fragment-order copy (below): the 16 x 16 x 32 MFMA sums over its 32 K slots in no particular order, so
// lane group g feeds K elements g*16 .. g*16+7 to the first MFMA of a 64-deep chunk and g*16+8 .. g*16+15 to the second,
// for the weight and the activation fragment alike.  What bounds a block is the request rate of the CU's vector memory
// pipe (~50 GB/s per CU, DESIGN.md section 4 "Generation"), so the order of requests is the design: activation rows and
// LayerNorm parameters first, then every weight fragment the registers can hold (up to three 768-deep K blocks), then
// keys / values; everything is in flight before the first wait.
//
// Time grows with the rows (every 16-row tile streams the layer's weights through L2); above 1024 rows the engine uses
// the GEMM path, whose 128-row tiles amortise the weights.
//
// Numerics follow the unfused path: q, k, v, attention output, GeLU output and the pre-LayerNorm sums are rounded to
// bf16 where that path stores them; sums are fp32.
#include "common.h"
#include "kernels.h"
#include <math.h>
#include <string.h>

// Diagnostic build only (tools/decode_stamps.py: -DKMB_DECODE_STAMP): per-workgroup s_memrealtime stamps (100 MHz) at
// the phase boundaries of the last launch of each kernel type.
#ifdef KMB_DECODE_STAMP
__device__ unsigned long long* g_dec_stamps = nullptr;
extern "C" int kmb_debug_set_decode_stamps(void* p) {
  unsigned long long* v = (unsigned long long*)p;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dec_stamps), &v, sizeof(v));
}
#define DSTAMP(type, i)                                                                                      \
  do {                                                                                                       \
    if (g_dec_stamps != nullptr && threadIdx.x == 0)                                                         \
      g_dec_stamps[((size_t)(type) * 4096 + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();       \
  } while (0)
// the resident decoder-layers kernel: 32 slots per workgroup, layer KMB_DL_STAMP_LAYER of the launch (tools/decode_resident_stamps.py)
#ifndef KMB_DL_STAMP_LAYER
#define KMB_DL_STAMP_LAYER 1
#endif
#define DLSTAMP(i)                                                                                           \
  do {                                                                                                       \
    if (g_dec_stamps != nullptr && threadIdx.x == 0 && l == KMB_DL_STAMP_LAYER)                              \
      g_dec_stamps[(size_t)blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memrealtime();                        \
  } while (0)
#else
#define DSTAMP(type, i)
#define DLSTAMP(i)
#endif

#ifndef KMB_DEC_SELF_VU
#define KMB_DEC_SELF_VU 10   // cached values of a row prefetched by the self-attention block.  20 (every value of a max_length 20 search in
                             // flight before the LayerNorm; 484 registers, no spill) measures SLOWER: 10.13-10.21 ms per generate against
                             // 10.01-10.03 (tools/gen_ab.sh, build.py --variant vu20 KMB_DEC_SELF_VU=20): 20 more requests per lane in
                             // front of the in-order return path delay the keys the scores wait for
#endif

namespace {

constexpr int RT = 16;      // rows per workgroup
constexpr int KBLK = 768;   // K block whose weight fragments are in flight together
constexpr int HD = 64;

// ---- weights in fragment order ----
// A wave-wide 16-byte-per-lane load of an MFMA weight fragment out of the row-major matrix touches 16 rows x 64 bytes:
// 16 half-used cache lines per instruction, and the vector memory pipe of a CU retires such requests at ~7 per us
// (in-kernel stamps: 24 of them took 2 us just to issue).  The decode blocks therefore read a copy of the decoder
// weights laid out in the order the fragments are consumed: fragment (16-row tile n, 64-deep chunk c, half s) is ONE
// contiguous KiB, lane l's eight elements at l*8 -- W[n*16 + (l & 15)][c*64 + (l >> 4)*16 + s*8 ...].  Packed once per
// generate() by pack_weights_kernel (99 MB for the six decoder layers of vcg_base, 0.1 ms).
struct WBlock { u32x4 w[24]; };   // one 16-column tile x 768 K

__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ void load_wblock(WBlock& f, const bf16_t* __restrict__ Wp, int K, int ntile, int k0, int lane) {
  const bf16_t* p = Wp + ((size_t)ntile * (K >> 6) + (k0 >> 6)) * 1024 + lane * 8;
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    f.w[2 * c] = *reinterpret_cast<const u32x4*>(p + c * 1024);
    f.w[2 * c + 1] = *reinterpret_cast<const u32x4*>(p + c * 1024 + 512);
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

struct PackDesc { const bf16_t* src; bf16_t* dst; int ld, N, K; int first; };   // first: index of its first 16-byte chunk
constexpr int PACK_MAX = 48;
struct PackArgs { PackDesc m[PACK_MAX]; int n; int total; };

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs a) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.total; i += gridDim.x * 256) {
    int mi = 0;
    while (mi + 1 < a.n && a.m[mi + 1].first <= i) ++mi;
    const PackDesc& d = a.m[mi];
    const int o = i - d.first;                 // chunk index in the packed matrix
    const int lane = o & 63, sub = (o >> 6) & 1, cc = o >> 7;
    const int kc = d.K >> 6;
    const int c = cc % kc, nt = cc / kc;
    const bf16_t* src = d.src + (size_t)(nt * 16 + (lane & 15)) * d.ld + c * 64 + (lane >> 4) * 16 + sub * 8;
    *reinterpret_cast<u32x4*>(d.dst + (size_t)o * 8) = *reinterpret_cast<const u32x4*>(src);
  }
}

// ---- the 16 input rows of a workgroup -> LDS (row stride a_stride bytes) ----
// Wave w stages rows 4w .. 4w+3.  The loads are issued first (rows_issue), ahead of the weight fragments in the wave's
// in-order memory queue; rows_finish normalises (gamma != null: LayerNorm with the arithmetic of ln_fwd_kernel, the
// four rows' reductions interleaved) and writes LDS; `keep` != null: the normalised rows also go to memory.
template <int NCH>   // 16-byte chunks per lane: K / 8 / 64 rounded up
struct RowRegs {
  u32x4 raw[4][NCH];
  f32x4 gb[NCH <= 2 ? NCH : 1][4];   // gamma, beta of this lane's columns (requested with the rows: behind the weight
};                                   // fragments they would wait for the whole weight stream to arrive)

template <int NCH>
__device__ __forceinline__ void rows_issue(RowRegs<NCH>& rr, const bf16_t* __restrict__ in, int ld_in, int row0, int R, int K,
                                           const float* __restrict__ gamma, const float* __restrict__ beta, int wave, int lane) {
  const int nch = K >> 3;
  if (NCH <= 2 && gamma != nullptr) {
#pragma unroll
    for (int j = 0; j < (NCH <= 2 ? NCH : 1); ++j) {
      const int c = lane + 64 * j < nch ? lane + 64 * j : 0;
      rr.gb[j][0] = *reinterpret_cast<const f32x4*>(gamma + c * 8);
      rr.gb[j][1] = *reinterpret_cast<const f32x4*>(gamma + c * 8 + 4);
      rr.gb[j][2] = *reinterpret_cast<const f32x4*>(beta + c * 8);
      rr.gb[j][3] = *reinterpret_cast<const f32x4*>(beta + c * 8 + 4);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + wave * 4 + i;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      rr.raw[i][j] = (row < R && c < nch) ? *reinterpret_cast<const u32x4*>(in + (size_t)row * ld_in + c * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  }
}

template <int NCH>
__device__ __forceinline__ void rows_finish(const RowRegs<NCH>& rr, int row0, int R, int K, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, float eps, bf16_t* __restrict__ keep, char* lds_a,
                                            int a_stride, int wave, int lane) {
  const int nch = K >> 3;
  if (NCH > 2 || gamma == nullptr) {   // the LayerNorm inputs of a decoder layer are d_model wide (NCH <= 2)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = lane + 64 * j;
        if (c < nch) *reinterpret_cast<u32x4*>(lds_a + (wave * 4 + i) * a_stride + c * 16) = rr.raw[i][j];
      }
    return;
  }
  constexpr int NL = NCH > 2 ? 1 : NCH;   // (keeps the dead instantiation small)
  float v[4][NL][8], s[4], q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      unpack8(rr.raw[i][j], v[i][j]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[i] += v[i][j][e];   // chunks past K are zero
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = wave_sum(s[i]) / (float)K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    q[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if (lane + 64 * j < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float dlt = v[i][j][e] - s[i]; q[i] += dlt * dlt; }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = rsqrtf(wave_sum(q[i]) / (float)K + eps);
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int c = lane + 64 * j;
    if (c < nch) {
      const f32x4 g0 = rr.gb[j][0], g1 = rr.gb[j][1], b0 = rr.gb[j][2], b1 = rr.gb[j][3];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int lr = wave * 4 + i, row = row0 + lr;
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (v[i][j][e] - s[i]) * q[i] * g0[e] + b0[e];
          o[4 + e] = (v[i][j][4 + e] - s[i]) * q[i] * g1[e] + b1[e];
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
// the same over rotations inside a row of 16 lanes (DPP row_ror: a register move, not a trip through the LDS crossbar)
template <int CTRL>
__device__ __forceinline__ float row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, row_ror<0x128>(v)); v = fmaxf(v, row_ror<0x124>(v)); v = fmaxf(v, row_ror<0x122>(v)); v = fmaxf(v, row_ror<0x121>(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += row_ror<0x128>(v); v += row_ror<0x124>(v); v += row_ror<0x122>(v); v += row_ror<0x121>(v);
  return v;
}

// Which (weight slice, row tile) a workgroup takes.  The `tiles` row tiles that read the same weight slice are split
// into RG groups; a (slice, group) unit runs on ONE XCD (workgroup b runs on XCD b % 8), so a slice crosses the fabric
// into RG L2s instead of all eight, and the units are dealt round-robin over the XCDs.  Launch 8 * ceil(units / 8) *
// tiles_per_group workgroups; returns false for the padding ones.
constexpr int RG = 2;
__device__ __forceinline__ bool unit_of(int nslices, int tiles, int& slice, int& tile) {
  const int tpg = (tiles + RG - 1) / RG;
  const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
  const int u = xcd + 8 * (i / tpg);
  slice = u / RG;
  tile = (u % RG) * tpg + i % tpg;
  return slice < nslices && tile < tiles;
}
inline int unit_grid(int nslices, int tiles) {
  const int tpg = (tiles + RG - 1) / RG;
  return 8 * ((nslices * RG + 7) / 8) * tpg;
}

// ------------------------------------------------------------------------------------------ kind 0: projection
// one workgroup per (64 * NTW-column block, row tile), see unit_of; wave w owns 16-column tiles w*NTW .. of the block.
// NTW = 2 (the wide fc1): one 768-deep K block only.
template <int NCH, int NTW>
__device__ __forceinline__ void decode_proj_body(const KmbDecodeBlock& p, char* smem) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int nb, rt;
  if (!unit_of(p.N / (64 * NTW), (p.R + RT - 1) / RT, nb, rt)) return;
  const int row0 = rt * RT, nt0 = (nb * 4 + wave) * NTW;   // first 16-column tile of this wave
  const int a_stride = (p.K + 8) * 2;
  const int nblk = p.K / KBLK;
  const int r = lane & 15, g = lane >> 4;
  const int row = row0 + r;
  [[maybe_unused]] const int stype = NCH == 6 ? 1 : 0;
  DSTAMP(stype, 0);
  RowRegs<NCH> rr;
  rows_issue<NCH>(rr, p.in, p.ld_in, row0, p.R, p.K, p.gamma, p.beta, wave, lane);
  // NTW = 1: up to three K blocks are requested before anything is waited for (these launches have at most one
  // workgroup per CU, so the 288 registers cost no occupancy).  NTW = 2 (fc1: two workgroups per CU have to fit, 256
  // registers each): the second tile's fragments are requested after the LayerNorm, when its registers are free.
  // (Measured and dropped at the end of round 4: a FOURTH block for fc2, requested behind the row staging into the registers the
  // staged rows have left, instead of into slot 0 after the first block's MFMAs -- 9.8 -> 10.2 us: the block is bound by the
  // number of requests its CU issues, not by when the last one is issued.)
  constexpr int NWB = NTW == 2 ? 2 : 3;
  WBlock wb[NWB];
  load_wblock(wb[0], p.W, p.K, nt0, 0, lane);
  if (NTW == 1) {
    if (nblk > 1) load_wblock(wb[1], p.W, p.K, nt0, KBLK, lane);
    if (nblk > 2) load_wblock(wb[NWB - 1], p.W, p.K, nt0, 2 * KBLK, lane);
  }
  f32x4 bias[NTW];
  uint2 res[NTW];
  auto load_bias_res = [&]() {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int col = (nt0 + t) * 16 + g * 4;
      bias[t] = *reinterpret_cast<const f32x4*>(p.bias + col);
      res[t] = (p.residual != nullptr && row < p.R) ? *reinterpret_cast<const uint2*>(p.residual + (size_t)row * p.ld_res + col) : uint2{0u, 0u};
    }
  };
  if (NTW == 1) load_bias_res();
  DSTAMP(stype, 1);
  rows_finish<NCH>(rr, row0, p.R, p.K, p.gamma, p.beta, p.eps, nb == 0 ? p.ln_out : nullptr, smem, a_stride, wave, lane);
  if (NTW == 2) {
    __builtin_amdgcn_sched_barrier(0);   // (keeps the second tile's 96 registers out of the LayerNorm's live range)
    // bias / residual of the 256-register kernel are requested here, into registers the staged rows have left: requested with
    // the first tile's fragments they were spilled while still in flight -- `s_waitcnt vmcnt(2)` in front of the LayerNorm
    load_bias_res();
    load_wblock(wb[1], p.W, p.K, nt0 + 1, 0, lane);
  }
  __syncthreads();
  DSTAMP(stype, 2);
  f32x4 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (NTW == 2) {
    mma_wblock(acc[0], wb[0], smem, a_stride, 0, lane);
    mma_wblock(acc[NTW - 1], wb[1], smem, a_stride, 0, lane);
  } else {   // K blocks 0 .. 3 (host: K <= 3072); slot of block b: b % 3
    mma_wblock(acc[0], wb[0], smem, a_stride, 0, lane);
    if (nblk > 3) load_wblock(wb[0], p.W, p.K, nt0, 3 * KBLK, lane);
    if (nblk > 1) mma_wblock(acc[0], wb[1], smem, a_stride, KBLK, lane);
    if (nblk > 2) mma_wblock(acc[0], wb[NWB - 1], smem, a_stride, 2 * KBLK, lane);
    if (nblk > 3) mma_wblock(acc[0], wb[0], smem, a_stride, 3 * KBLK, lane);
  }
  DSTAMP(stype, 3);
  if (row >= p.R) return;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int col = (nt0 + t) * 16 + g * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = acc[t][e] + bias[t][e];
    if (p.act == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
    }
    if (p.residual != nullptr) {
      v[0] += lo_bf(res[t].x); v[1] += hi_bf(res[t].x); v[2] += lo_bf(res[t].y); v[3] += hi_bf(res[t].y);
    }
    *reinterpret_cast<uint2*>(p.out + (size_t)row * p.ld_out + col) = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
  }
  DSTAMP(stype, 4);
}

template <int NCH>
__global__ __launch_bounds__(256) void decode_proj_kernel(const KmbDecodeBlock p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  decode_proj_body<NCH, 1>(p, smem);
}
// the wide fc1: 480 workgroups, two per CU -> two waves per SIMD, 256 registers each
__global__ __launch_bounds__(256, 2) void decode_proj_wide_kernel(const KmbDecodeBlock p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  decode_proj_body<2, 2>(p, smem);
}

// ------------------------------------------------------------------------------------------ kinds 1, 2: attention
// one workgroup per (head, row tile).  SELF: W rows [q | k | v] (3 x H x 64), wave w owns 16-column tiles 3w .. 3w+2 of
// the head's 192 columns; the new key / value row goes to the cache at position Tk - 1.  Cross: W rows are the q rows
// only, wave w owns tile w; keys / values are the cached projections of the encoder output of the row's batch item.
// Attention: 16 lanes per row.  Scores: 4 lanes per key (16 of the 64 dimensions each), 4 keys per step and KU steps in
// flight; values: lane s owns output elements 4s .. 4s+3, VU value rows in flight.  (One key per lane and four value
// rows in flight made the 100-key cross-attention a chain of 32 dependent L2 round trips: 16 us.)
// KVLDS (cross-attention with kv_group >= 4 beams per batch item): the 16 rows of a tile belong to at most four batch
// items, and the rows of an item attend to the SAME cached keys / values.  The workgroup copies those (<= 4 x Tk x 256
// bytes of this head) into LDS once -- whole 128-byte lines, requested right behind the weight fragments so they land
// during the projection -- instead of every row fetching its own 32-byte pieces from L2 (16 x Tk x 256 bytes in
// quarter-line requests: the request rate of the vector memory pipe made that 19 of the block's 26 us).
// With the keys / values of the tile's items in LDS the attention itself runs on the matrix cores (round 4; on the vector
// ALUs the 64-key cross-attention was 8.4 of the block's 18 us: 2 x 65 k multiply-adds per workgroup, each with a bf16 unpack):
//   scores  S^T = K_staged q^T   : one 16-key tile x 16 rows per MFMA pair, EVERY staged key against every row (<= 4 items:
//                                  the products of the other items' keys are computed and dropped -- 16 MFMAs per wave)
//   softmax per row over its own item's keys (fp32, as before); e = exp(s - max) leaves as TWO bf16 matrices e_hi + e_lo
//                                  (e_hi = bf16(e), e_lo = bf16(e - e_hi): 16 significant bits, so that the weights carry
//                                  the precision the fp32 vector path gave them), zero at the other items' keys
//   output  O^T = V_staged^T (e_hi + e_lo)^T : values read as transposed fragments (ds_read_b64_tr_b16), wave w owns
//                                  head dimensions 16 w .. 16 w + 15; scaled by 1 / sum(e) at the store.
constexpr int KV_ITEMS = 4;      // batch items a 16-row tile can touch when kv_group >= 4
constexpr int NKV = 15;          // 16-byte chunks per thread and operand: KV_ITEMS * Tk * 8 <= NKV * 256  (Tk <= 120)
constexpr int KS = 144;          // LDS row stride of a staged key row (128 + 16: conflict-free 16-row fragment reads)
__host__ __device__ constexpr int kv_pad(int Tk) { return (KV_ITEMS * Tk + 31) & ~31; }     // staged keys, padded to MFMA K steps
__host__ __device__ constexpr int pb_stride(int Tk) { return kv_pad(Tk) * 2 + 16; }          // bytes per row of e_hi / e_lo

template <bool SELF, bool KVLDS>
__global__ __launch_bounds__(256) void decode_attn_kernel(const KmbDecodeBlock p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(!(SELF && KVLDS), "the self-attention cache is per row");
  constexpr int NTW = SELF ? 3 : 1;          // 16-column tiles per wave
  constexpr int QW = SELF ? 3 * HD : HD;     // projected columns of the head
  constexpr int QS = (QW + 8) * 2;           // LDS row stride of the projected tile
  constexpr int KU = 5, VU = SELF ? KMB_DEC_SELF_VU : 10;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int h, rt;
  if (!unit_of(p.H, (p.R + RT - 1) / RT, h, rt)) return;
  const int row0 = rt * RT;
  const int d = p.H * HD;
  const int a_stride = (p.K + 8) * 2;
  char* const lds_q = smem + RT * a_stride;
  float* const sc = reinterpret_cast<float*>(lds_q + RT * QS);   // [16][Tk] scores
  // tile t (0 .. 11 | 0 .. 3) of this head covers weight rows part (q | k | v) * d + h * 64 + (t % 4) * 16
  auto tile_of = [&](int t) { return ((t >> 2) * d + h * HD) / 16 + (t & 3); };
  [[maybe_unused]] const int stype = SELF ? 2 : 3;
  DSTAMP(stype, 0);
  RowRegs<2> rr;
  rows_issue<2>(rr, p.in, p.ld_in, row0, p.R, p.K, p.gamma, p.beta, wave, lane);
  WBlock wb[NTW];   // every tile's fragments in flight at once (one workgroup per CU: the registers are free)
  f32x4 acc[NTW], bias[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) load_wblock(wb[t], p.W, p.K, tile_of(wave * NTW + t), 0, lane);
  // SELF: the first 4 * KU cached keys and VU cached values of this thread's row (all of them for max_length <= 21 / 11)
  // do not depend on the projection; fetched after it they were two exposed L2 round trips (5 of the block's 14.6 us).
  // They are requested behind the LayerNorm (below), not in front of it: in the in-order return path they follow the weight
  // fragments either way, and requested first they were 60 more live registers across the LayerNorm -- hipcc parked them in
  // accumulator registers there, and a copy of a value that is still in flight is `s_waitcnt vmcnt(0)`: the LayerNorm
  // waited for the whole weight stream (stamps: 10 of the block's 16 us before the projection could start).
  const int a_lr = threadIdx.x >> 4, a_s = threadIdx.x & 15;
  [[maybe_unused]] u32x4 pk0[SELF ? KU : 1], pk1[SELF ? KU : 1];
  [[maybe_unused]] uint2 pv[SELF ? VU : 1];
  // history index (p.hist: position t of a row lives in cache row hist[row][t]; a beam reorder permutes these rows instead of copying the
  // caches): the cache rows of the keys / values prefetched below, requested first of all so that they are back when the LayerNorm is done
  [[maybe_unused]] int hk[SELF ? KU : 1], hv[SELF ? VU : 1];
  if (SELF) {
    const int prow = row0 + a_lr < p.R ? row0 + a_lr : 0;
    const int tc = p.Tk - 1;
    const int32_t* hrow = p.hist != nullptr ? p.hist + (size_t)prow * p.Tmax : nullptr;
#pragma unroll
    // (positions past the cached ones are dummy loads of position 0: from the row's OWN cache row -- hist[.][0] may not be written yet)
    for (int u = 0; u < KU; ++u) { const int t = (a_s >> 2) + 4 * u; const int v = hrow != nullptr ? hrow[t < tc ? t : 0] : prow; hk[u] = t < tc ? v : prow; }
#pragma unroll
    for (int u = 0; u < VU; ++u) { const int v = hrow != nullptr ? hrow[u < tc ? u : 0] : prow; hv[u] = u < tc ? v : prow; }
  }
  // KVLDS: this thread's chunks of the tile's keys / values (chunk c = tid + 256 i: staged row c / 8, 16-byte piece c % 8)
  [[maybe_unused]] u32x4 kreg[KVLDS ? NKV : 1], vreg[KVLDS ? NKV : 1];
  [[maybe_unused]] int first_item = 0, kv_rows = 0;
  char* const lds_k = reinterpret_cast<char*>(sc) + (((size_t)RT * p.Tk * sizeof(float) + 15) & ~(size_t)15);
  char* const lds_v = lds_k + (size_t)KV_ITEMS * p.Tk * KS;
  float* const lds_m = reinterpret_cast<float*>(lds_v + (size_t)KV_ITEMS * p.Tk * 128);   // 0 / -inf per staged key
  [[maybe_unused]] long long mreg[2] = {1, 1};   // key mask of staged rows tid, tid + 256 (KV_ITEMS * Tk <= 512)
  if (KVLDS) {
    const int last_row = row0 + RT - 1 < p.R ? row0 + RT - 1 : p.R - 1;
    first_item = row0 / p.kv_group;
    kv_rows = (last_row / p.kv_group - first_item + 1) * p.Tk;     // staged rows: item-major, Tk per item
    int rowi = threadIdx.x >> 3;
    int li = rowi / p.Tk, t = rowi - li * p.Tk;
    const int seg = threadIdx.x & 7;
#pragma unroll
    for (int i = 0; i < NKV; ++i) {
      if (rowi < kv_rows) {   // (no request at all past the staged rows: the block is bound by the requests a CU can issue)
        const size_t off = ((size_t)(first_item + li) * p.Tmax + t) * p.ldc + h * HD + seg * 8;
        kreg[i] = *reinterpret_cast<const u32x4*>(p.Kc + off);
        vreg[i] = *reinterpret_cast<const u32x4*>(p.Vc + off);
      }
      rowi += 32; t += 32;
      while (t >= p.Tk) { t -= p.Tk; ++li; }
    }
    if (p.key_mask != nullptr) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int mr = threadIdx.x + 256 * i;
        if (mr < kv_rows) mreg[i] = p.key_mask[(size_t)(first_item + mr / p.Tk) * p.mask_ld + mr % p.Tk];
      }
    }
  }
  DSTAMP(stype, 7);   // every load of the block has been issued
  rows_finish<2>(rr, row0, p.R, p.K, p.gamma, p.beta, p.eps, h == 0 ? p.ln_out : nullptr, smem, a_stride, wave, lane);
  // the bias is requested only now, into the registers the staged rows have left: requested with the weights it was the
  // value hipcc chose to spill in the self-attention block (288 fragment registers), and a spill of a value that is still
  // in flight is `s_waitcnt vmcnt(0)` -- the block waited for its whole weight stream before it requested its keys
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int tile = wave * NTW + t;
    bias[t] = *reinterpret_cast<const f32x4*>(p.bias + (tile >> 2) * d + h * HD + (tile & 3) * 16 + (lane >> 4) * 4);
  }
  if (SELF) {
    const int prow = row0 + a_lr < p.R ? row0 + a_lr : 0;
    const int tc = p.Tk - 1;
#pragma unroll
    for (int u = 0; u < KU; ++u) {   // (clamped addresses, no branch: a conditional load's register copy waits for the load)
      const int t = (a_s >> 2) + 4 * u, tt = t < tc ? t : 0;
      const bf16_t* kr = p.Kc + ((size_t)hk[u] * p.Tmax + tt) * p.ldc + h * HD + (a_s & 3) * 16;
      pk0[u] = *reinterpret_cast<const u32x4*>(kr);
      pk1[u] = *reinterpret_cast<const u32x4*>(kr + 8);
    }
#pragma unroll
    for (int u = 0; u < VU; ++u)
      pv[u] = *reinterpret_cast<const uint2*>(p.Vc + ((size_t)hv[u] * p.Tmax + (u < tc ? u : 0)) * p.ldc + h * HD + a_s * 4);
    (void)prow;
  }
  __syncthreads();
  DSTAMP(stype, 1);
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_wblock(acc[t], wb[t], smem, a_stride, 0, lane);
  }
  {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int tile = wave * NTW + t, part = tile >> 2;
      const int col = (tile & 3) * 16 + g * 4;           // within the head
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[t][e] + bias[t][e];
      if (part == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.q_scale;
      }
      const uint2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      *reinterpret_cast<uint2*>(lds_q + r * QS + (part * HD + col) * 2) = pk;
      if (SELF && part > 0 && row0 + r < p.R) {   // append to the cache (the row's own cache row; the history index records it)
        bf16_t* dst = (part == 1 ? p.Kc : p.Vc) + ((size_t)(row0 + r) * p.Tmax + (p.Tk - 1)) * p.ldc + h * HD + col;
        *reinterpret_cast<uint2*>(dst) = pk;
        if (p.hist != nullptr && h == 0 && tile == 4 && g == 0) p.hist[(size_t)(row0 + r) * p.Tmax + (p.Tk - 1)] = row0 + r;
      }
    }
  }
  if (KVLDS) {
    int rowi = threadIdx.x >> 3;
    const int seg = threadIdx.x & 7;
#pragma unroll
    for (int i = 0; i < NKV; ++i) {
      if (rowi < kv_rows) {
        *reinterpret_cast<u32x4*>(lds_k + (size_t)rowi * KS + seg * 16) = kreg[i];
        *reinterpret_cast<u32x4*>(lds_v + (size_t)rowi * 128 + seg * 16) = vreg[i];
      }
      rowi += 32;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (threadIdx.x + 256 * i < kv_rows) lds_m[threadIdx.x + 256 * i] = mreg[i] == 0 ? -INFINITY : 0.f;
  }
  DSTAMP(stype, 2);
  __syncthreads();
  DSTAMP(stype, 3);
  if constexpr (KVLDS) {
    // ---- attention on the matrix cores (see the note above KV_ITEMS) ----
    const int r = lane & 15, g = lane >> 4;
    const int Tk = p.Tk, kpad = (kv_rows + 31) & ~31, PBS = pb_stride(Tk);
    char* const pb_hi = smem;                                               // the activation rows are done with
    char* const pb_lo = reinterpret_cast<char*>(lds_m + KV_ITEMS * Tk);
    float* const inv_s = reinterpret_cast<float*>(pb_lo + RT * PBS);
    const int t2 = 2 * Tk, t3 = 3 * Tk;
    {   // scores: lane (r, g) of key tile kt ends up with keys kt * 16 + 4 g .. + 3 of row r
      const int rrow = row0 + r < p.R ? row0 + r : p.R - 1;
      const int item_r = rrow / p.kv_group - first_item;
      const bf16x8 q0 = *reinterpret_cast<const bf16x8*>(lds_q + r * QS + g * 16);
      const bf16x8 q1 = *reinterpret_cast<const bf16x8*>(lds_q + r * QS + 64 + g * 16);
      for (int kt = wave; kt * 16 < kv_rows; kt += 4) {
        const int key = kt * 16 + r < kv_rows ? kt * 16 + r : 0;
        const char* kr = lds_k + (size_t)key * KS + g * 16;
        f32x4 sa = {0.f, 0.f, 0.f, 0.f};
        sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kr), q0, sa, 0, 0, 0);
        sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kr + 64), q1, sa, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kidx = kt * 16 + 4 * g + e;
          const int it = (kidx >= Tk) + (kidx >= t2) + (kidx >= t3);
          if (kidx < kv_rows && it == item_r) sc[(size_t)r * Tk + (kidx - it * Tk)] = sa[e] + lds_m[kidx];
        }
      }
    }
    __syncthreads();
    DSTAMP(stype, 4);
    {   // softmax: 16 lanes per row; e_hi / e_lo over the staged key index (eight keys = one 16-byte store per lane and
        // matrix), zero outside the row's item
      const int lr = threadIdx.x >> 4, s = threadIdx.x & 15;
      const int rrow = row0 + lr < p.R ? row0 + lr : p.R - 1;
      const int lo_k = (rrow / p.kv_group - first_item) * Tk, hi_k = lo_k + Tk;   // the item's staged keys
      const float* my = sc + (size_t)lr * Tk;
      float mx = -INFINITY;
      for (int t = s; t < Tk; t += 16) mx = fmaxf(mx, my[t]);
      mx = row16_max(mx);
      float l = 0.f;
      for (int c = s * 8; c < kpad; c += 128) {
        float e[8], rem[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = 0.f;
        if (c + 8 > lo_k && c < hi_k && mx != -INFINITY) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (c + i >= lo_k && c + i < hi_k) e[i] = __expf(my[c + i - lo_k] - mx);
        }
        const u32x4 hi = pack8(e);
        float back[8];
        unpack8(hi, back);
#pragma unroll
        for (int i = 0; i < 8; ++i) { l += e[i]; rem[i] = e[i] - back[i]; }
        *reinterpret_cast<u32x4*>(pb_hi + lr * PBS + c * 2) = hi;
        *reinterpret_cast<u32x4*>(pb_lo + lr * PBS + c * 2) = pack8(rem);
      }
      l = row16_sum(l);
      if (s == 0) inv_s[lr] = l > 0.f ? 1.f / l : 0.f;
    }
    __syncthreads();
    f32x4 oa = {0.f, 0.f, 0.f, 0.f};   // lane (r, g): head dimensions 16 wave + 4 g .. + 3 of row r
    for (int k0 = 0; k0 < kpad; k0 += 32) {
      bf16x8 vf;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        int krow = k0 + g * 8 + hh * 4 + (r >> 2);
        krow = krow < kv_rows ? krow : kv_rows - 1;   // its weight is zero; the row has to be finite
        const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(lds_v + (size_t)krow * 128 + wave * 32 + (r & 3) * 8));
        vf[hh * 4 + 0] = t[0]; vf[hh * 4 + 1] = t[1]; vf[hh * 4 + 2] = t[2]; vf[hh * 4 + 3] = t[3];
      }
      const bf16x8 ph = *reinterpret_cast<const bf16x8*>(pb_hi + r * PBS + (k0 + g * 8) * 2);
      const bf16x8 pl = *reinterpret_cast<const bf16x8*>(pb_lo + r * PBS + (k0 + g * 8) * 2);
      oa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, ph, oa, 0, 0, 0);
      oa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pl, oa, 0, 0, 0);
    }
    DSTAMP(stype, 5);
    if (row0 + r < p.R) {
      const float inv = inv_s[r];
      *reinterpret_cast<uint2*>(p.out + (size_t)(row0 + r) * p.ld_out + h * HD + wave * 16 + g * 4) =
          uint2{pack2bf(oa[0] * inv, oa[1] * inv), pack2bf(oa[2] * inv, oa[3] * inv)};
    }
    DSTAMP(stype, 6);
    return;
  }
  // ---- attention ----
  const int lr = threadIdx.x >> 4, s = threadIdx.x & 15;
  const int row = row0 + lr;
  if (row >= p.R) return;   // no workgroup barrier below
  const int crow = KVLDS ? row / p.kv_group : p.kv_row != nullptr ? p.kv_row[row] : row;
  const bf16_t* Kc = p.Kc + (size_t)crow * p.Tmax * p.ldc + h * HD;
  const bf16_t* Vc = p.Vc + (size_t)crow * p.Tmax * p.ldc + h * HD;
  [[maybe_unused]] const char* const my_k = lds_k + (size_t)(crow - first_item) * p.Tk * KS;
  [[maybe_unused]] const char* const my_v = lds_v + (size_t)(crow - first_item) * p.Tk * 128;
  [[maybe_unused]] const float* const my_m = lds_m + (size_t)(crow - first_item) * p.Tk;
  const int64_t* km = (!KVLDS && p.key_mask != nullptr) ? p.key_mask + (size_t)crow * p.mask_ld : nullptr;
  const int kq = s >> 2, part = s & 3;
  float qp[16];   // this lane's 16 of the row's 64 query elements
  unpack8(*reinterpret_cast<const u32x4*>(lds_q + lr * QS + part * 32), qp);
  unpack8(*reinterpret_cast<const u32x4*>(lds_q + lr * QS + part * 32 + 16), qp + 8);
  const int Tc = SELF ? p.Tk - 1 : p.Tk;   // rows that live in the cache
  float* const my = sc + (size_t)lr * p.Tk;
  float mx = -INFINITY;
  for (int t0 = 0; t0 < Tc; t0 += 4 * KU) {
    u32x4 k0[KU], k1[KU];
    long long mk[KU];
    [[maybe_unused]] float madd[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int t = t0 + kq + 4 * u;
      const bool ok = t < Tc;
      if (SELF && t0 == 0) {
        k0[u] = pk0[u];
        k1[u] = pk1[u];
      } else if (KVLDS) {
        const char* kr = my_k + (size_t)(ok ? t : 0) * KS + part * 32;
        k0[u] = *reinterpret_cast<const u32x4*>(kr);
        k1[u] = *reinterpret_cast<const u32x4*>(kr + 16);
        madd[u] = my_m[ok ? t : 0];
      } else if (SELF && p.hist != nullptr) {
        const int tt = ok ? t : 0;
        const bf16_t* kr = p.Kc + ((size_t)p.hist[(size_t)crow * p.Tmax + tt] * p.Tmax + tt) * p.ldc + h * HD + part * 16;
        k0[u] = *reinterpret_cast<const u32x4*>(kr);
        k1[u] = *reinterpret_cast<const u32x4*>(kr + 8);
      } else {
        const bf16_t* kr = Kc + (size_t)(ok ? t : 0) * p.ldc + part * 16;
        k0[u] = *reinterpret_cast<const u32x4*>(kr);
        k1[u] = *reinterpret_cast<const u32x4*>(kr + 8);
      }
      mk[u] = km != nullptr ? km[ok ? t : 0] : 1;
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int t = t0 + kq + 4 * u;
      float k8[16];
      unpack8(k0[u], k8);
      unpack8(k1[u], k8 + 8);
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) dot += qp[e] * k8[e];
      dot += __shfl_xor(dot, 1, 16);
      dot += __shfl_xor(dot, 2, 16);
      if (KVLDS) dot += madd[u];
      else if (mk[u] == 0) dot = -INFINITY;
      if (t < Tc) {
        if (part == 0) my[t] = dot;
        mx = fmaxf(mx, dot);
      }
    }
  }
  if (SELF) {   // the new key: from the projection output in LDS, 4 elements per lane
    float q4[4], k4[4];
    const uint2 qv = *reinterpret_cast<const uint2*>(lds_q + lr * QS + s * 8);
    const uint2 kv = *reinterpret_cast<const uint2*>(lds_q + lr * QS + (HD + s * 4) * 2);
    q4[0] = lo_bf(qv.x); q4[1] = hi_bf(qv.x); q4[2] = lo_bf(qv.y); q4[3] = hi_bf(qv.y);
    k4[0] = lo_bf(kv.x); k4[1] = hi_bf(kv.x); k4[2] = lo_bf(kv.y); k4[3] = hi_bf(kv.y);
    const float dot = group16_sum((q4[0] * k4[0] + q4[1] * k4[1]) + (q4[2] * k4[2] + q4[3] * k4[3]));
    if (s == 0) my[Tc] = dot;
    mx = fmaxf(mx, dot);
  }
  mx = group16_max(mx);
  DSTAMP(stype, 4);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the 16 lanes of a row are in one wave
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float l = 0.f;
  for (int t = s; t < p.Tk; t += 16) {
    const float e = (mx == -INFINITY) ? 0.f : __expf(my[t] - mx);
    my[t] = e;
    l += e;
  }
  l = group16_sum(l);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float a[2][4];   // two partial sums (independent chains) x four output elements
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) a[i][e] = 0.f;
  for (int t0 = 0; t0 < Tc; t0 += VU) {
    uint2 vv[VU];
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int t = t0 + u < Tc ? t0 + u : 0;
      if (SELF && t0 == 0) { vv[u] = pv[u]; continue; }
      vv[u] = KVLDS ? *reinterpret_cast<const uint2*>(my_v + (size_t)t * 128 + s * 8)
              : (SELF && p.hist != nullptr)
                  ? *reinterpret_cast<const uint2*>(p.Vc + ((size_t)p.hist[(size_t)crow * p.Tmax + t] * p.Tmax + t) * p.ldc + h * HD + s * 4)
                  : *reinterpret_cast<const uint2*>(Vc + (size_t)t * p.ldc + s * 4);
    }
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const float w = t0 + u < Tc ? my[t0 + u] : 0.f;
      a[u & 1][0] += w * lo_bf(vv[u].x); a[u & 1][1] += w * hi_bf(vv[u].x);
      a[u & 1][2] += w * lo_bf(vv[u].y); a[u & 1][3] += w * hi_bf(vv[u].y);
    }
  }
  if (SELF) {
    const uint2 vv = *reinterpret_cast<const uint2*>(lds_q + lr * QS + (2 * HD + s * 4) * 2);
    const float w = my[Tc];
    a[1][0] += w * lo_bf(vv.x); a[1][1] += w * hi_bf(vv.x); a[1][2] += w * lo_bf(vv.y); a[1][3] += w * hi_bf(vv.y);
  }
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (a[0][e] + a[1][e]) * inv;
  DSTAMP(stype, 5);
  *reinterpret_cast<uint2*>(p.out + (size_t)row * p.ld_out + h * HD + s * 4) = uint2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
  DSTAMP(stype, 6);
}

// ================================================================================ resident decoder-layers kernel (round 5)
// One launch runs n_layers whole decoder layers of a decode step: 12 workgroups per 16-row tile (slot j = head j / output
// columns 64 j .. / fc1 columns 256 j ..), all co-resident (12 x tiles <= 256 CUs), walking the six blocks of a layer as PHASES
// that are separated by a 12-workgroup counter barrier per row tile instead of a kernel boundary.  What that buys (DESIGN.md
// section 4 "Generation", round 4's measurement: a block costs the same for 40 rows as for 320, every launch starts on cold
// L2s): the weight fragments of phase p + 1 are requested right after phase p has published its tile, so they stream in while
// the barrier's hand-off latency passes; the XCD's L2 stays warm across phases and layers (slot-major block placement: the 20
// row tiles that read one weight slice sit on one XCD); 36 kernel boundaries per step become 6 / n_layers.
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md "Valid forms", row 1): a tile is published with
// write-through (sc1) stores of whole 128-byte lines out of an LDS staging image, every storing wave drains (s_waitcnt
// vmcnt(0)), the workgroup's barrier, then ONE lane adds 1 to the phase's counter of the row tile (agent-scope atomic); a
// consumer's lane 0 polls that counter with sc1 loads until it reads 12, the workgroup's barrier, and EVERY load of handed-off
// bytes is an sc1 buffer load (they bypass the CU's L1, which no other CU's store ever refreshes).  Results do not depend on
// block placement; the counters are zeroed by a memset node in front of the launch (kmb_decode_layers_launch); every spin is
// bounded and sets status bit 8 when it gives up (the step's results are then garbage and the host raises).
// Write-after-read on the exchange buffers needs no extra synchronisation: a workgroup overwrites buffer X in phase p + 2 only
// after the barrier of phase p + 1, which every reader of X's previous contents reaches after its reads.
// Arithmetic: the phases are the bodies of the six-launch blocks above (same LayerNorm, same MFMA order, same bf16
// roundings): the logits are bit-identical to KMB_GEN_FUSED=1 (tests/test_decode_fused_gpu.py).
constexpr int DL_SLOTS = 12;          // workgroups per row tile: d_model = 768 = 12 heads = 12 x 64 output columns
constexpr unsigned DL_SPIN_LIMIT = 400000u;

__device__ __forceinline__ u32x4 ld16_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16));
}

// the 16 input rows of the tile (K elements each) -> registers, through sc1 loads
template <int NCH>
__device__ __forceinline__ void rows_issue_sc1(RowRegs<NCH>& rr, __amdgpu_buffer_rsrc_t rs, int ld_in, int row0, int R, int K,
                                               const float* __restrict__ gamma, const float* __restrict__ beta, int wave, int lane) {
  const int nch = K >> 3;
  if (NCH <= 2 && gamma != nullptr) {
#pragma unroll
    for (int j = 0; j < (NCH <= 2 ? NCH : 1); ++j) {
      const int c = lane + 64 * j < nch ? lane + 64 * j : 0;
      rr.gb[j][0] = *reinterpret_cast<const f32x4*>(gamma + c * 8);
      rr.gb[j][1] = *reinterpret_cast<const f32x4*>(gamma + c * 8 + 4);
      rr.gb[j][2] = *reinterpret_cast<const f32x4*>(beta + c * 8);
      rr.gb[j][3] = *reinterpret_cast<const f32x4*>(beta + c * 8 + 4);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + wave * 4 + i;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      const bool ok = row < R && c < nch;
      const u32x4 v = ld16_sc1(rs, ok ? ((unsigned)row * (unsigned)ld_in + (unsigned)c * 8u) * 2u : 0u);   // clamped, no branch around the load
      rr.raw[i][j] = ok ? v : u32x4{0u, 0u, 0u, 0u};
    }
  }
}

// publish this workgroup's [16 x NC] bf16 tile (staged in LDS, row stride NC * 2 bytes) as columns col0 .. of rows row0 ..,
// whole 128-byte lines per store instruction, write-through; then drain, barrier, and one lane signals the counter
template <int NC>
__device__ __forceinline__ void publish_tile(const char* lds_out, __amdgpu_buffer_rsrc_t rd, int ld_out, int row0, int R, int col0,
                                             unsigned* ctr, int tid) {
  const int r = tid >> 4, seg = tid & 15;
  __syncthreads();   // the staging image is complete
  if (row0 + r < R) {
    const unsigned base = ((unsigned)(row0 + r) * (unsigned)ld_out + (unsigned)col0) * 2u;
    if constexpr (NC == 64) {
      const uint2 v = *reinterpret_cast<const uint2*>(lds_out + r * 128 + seg * 8);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, v), rd, (int)(base + seg * 8), 0, 16);
    } else {
#pragma unroll
      for (int j = 0; j < NC / 128; ++j) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(lds_out + r * (NC * 2) + (seg + 16 * j) * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rd, (int)(base + (seg + 16 * j) * 16), 0, 16);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wait until all DL_SLOTS workgroups of the row tile have signalled `ctr`; bounded
__device__ __forceinline__ void group_wait(unsigned* ctr, int32_t* status) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)DL_SLOTS) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > DL_SPIN_LIMIT) { atomicOr(status, 8); break; }
    }
  }
  __syncthreads();
}

// kind-0 epilogue of one 16-column tile into the LDS staging image (row stride ost bytes): + bias [+ GeLU] [+ residual from LDS]
template <bool GELU>
__device__ __forceinline__ void proj_epilogue_lds(const f32x4& acc, const f32x4& bias, const char* lds_res, int res_stride, int res_col,
                                                  char* lds_out, int ost, int out_col, int lane) {
  const int r = lane & 15, g = lane >> 4;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = acc[e] + bias[e];
  if (GELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
  }
  if (lds_res != nullptr) {
    const uint2 res = *reinterpret_cast<const uint2*>(lds_res + r * res_stride + (res_col + g * 4) * 2);
    v[0] += lo_bf(res.x); v[1] += hi_bf(res.x); v[2] += lo_bf(res.y); v[3] += hi_bf(res.y);
  }
  *reinterpret_cast<uint2*>(lds_out + r * ost + (out_col + g * 4) * 2) = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
}

// rows_finish<2> for the resident kernel: the same arithmetic (bit for bit), two rows at a time -- half the temporaries; the kernel
// holds up to 256 registers of prefetched weight fragments across this point, and what the LayerNorm spills comes back through
// scratch loads (measured: 5.2 us of a layer's 9 us first phase).  No `keep` output: the normalised rows stay in LDS.
__device__ __forceinline__ void rows_finish_lean(const RowRegs<2>& rr, int row0, int R, const float* __restrict__ gamma, float eps,
                                                 char* lds_a, int a_stride, int wave, int lane) {
  constexpr int K = KBLK, nch = K >> 3;
  if (gamma == nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = lane + 64 * j;
        if (c < nch) *reinterpret_cast<u32x4*>(lds_a + (wave * 4 + i) * a_stride + c * 16) = rr.raw[i][j];
      }
    return;
  }
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    float v[2][2][8], s[2], q[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      s[i] = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        unpack8(rr.raw[hf * 2 + i][j], v[i][j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[i] += v[i][j][e];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) s[i] = wave_sum(s[i]) / (float)K;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      q[i] = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (lane + 64 * j < nch) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float dlt = v[i][j][e] - s[i]; q[i] += dlt * dlt; }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) q[i] = rsqrtf(wave_sum(q[i]) / (float)K + eps);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        const f32x4 g0 = rr.gb[j][0], g1 = rr.gb[j][1], b0 = rr.gb[j][2], b1 = rr.gb[j][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int lr = wave * 4 + hf * 2 + i;
          float o[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (v[i][j][e] - s[i]) * q[i] * g0[e] + b0[e];
            o[4 + e] = (v[i][j][4 + e] - s[i]) * q[i] * g1[e] + b1[e];
          }
          *reinterpret_cast<u32x4*>(lds_a + lr * a_stride + c * 16) = pack8(o);
        }
      }
    }
  }
}

// L2 prefetch by the workgroups that share an XCD's L2 (MI355X_MICROARCH.md: a CU takes in ~125 GB/s from its XCD's L2 but only 30-60
// GB/s from the Infinity Cache, bytes in flight / 2 us): the `nparts` workgroups of a slot that sit on one XCD each touch THEIR share
// of the slice all of them are about to stream -- one 4-byte LDS-DMA per 128-byte line into a dummy LDS word (no register
// destination: nothing for a late arrival to overwrite) -- one phase ahead of its use.  Speed only: nothing reads the dummy.
__device__ __forceinline__ void touch_share(const void* base, unsigned bytes, int part, int nparts, char* lds_dummy, int tid) {
  const unsigned lines = bytes >> 7;
  const unsigned lo = (unsigned)(((unsigned long long)lines * (unsigned)part) / (unsigned)nparts);
  const unsigned hi = (unsigned)(((unsigned long long)lines * (unsigned)(part + 1)) / (unsigned)nparts);
  char* const dst = lds_dummy + (tid >> 6) * 256;   // wave-uniform base; lane l lands at + 4 l
  for (unsigned l0 = lo; l0 < hi; l0 += 256) {
    const unsigned ln = l0 + (unsigned)tid;
    if (ln < hi)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)base + (size_t)ln * 128),
                                       (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
  }
}

struct DLLds {   // byte offsets into the dynamic LDS image (kmb_decode_layers_lds computes the same)
  int a, q, sc, x, k, v, m, pbh, pbl, inv, out, dummy, total;
};
__host__ __device__ inline DLLds dl_lds(int Tmax, int S, int F) {
  DLLds L;
  const int a_bytes = RT * (KBLK + 8) * 2;
  L.a = 0;                                          // the phase's normalised input rows: MFMA operand of the LayerNorm phases, residual of the next one
  L.q = a_bytes;                                    // projected q | k | v tile of the attention phases
  L.sc = L.q + RT * (3 * HD + 8) * 2;               // fp32 scores [16][Tk]
  const int tk = Tmax > S ? Tmax : S;
  L.x = L.sc + ((RT * tk * 4 + 255) & ~255);        // X region: o rows (out-projections) | staged cross keys / values ... | fc2's 16 x F input rows
  L.k = L.x;
  L.v = L.k + KV_ITEMS * S * 128;                   // (keys: 128-byte rows with a chunk swizzle, filled by LDS-DMA -- not the KS-padded image of the block kernels)
  L.m = L.v + KV_ITEMS * S * 128;
  L.pbh = L.m + ((KV_ITEMS * S * 4 + 15) & ~15);
  L.pbl = L.pbh + RT * pb_stride(S);
  L.inv = L.pbl + RT * pb_stride(S);
  int end = L.inv + RT * 4;
  const int ob = L.x + a_bytes, hb = L.x + RT * (F + 8) * 2;
  if (ob > end) end = ob;
  if (hb > end) end = hb;
  L.out = (end + 255) & ~255;                       // staging image of the tile being published (16 x 64 or 16 x F / 12 columns)
  L.dummy = L.out + RT * (F / DL_SLOTS > 64 ? F / DL_SLOTS : 64) * 2;   // landing words of the L2 touches (never read)
  L.total = L.dummy + 1024;
  return L;
}

template <int NT5, int NKVT>   // NT5 = ffn width / 768: fc1's 16-column tiles per wave, fc2's 768-deep K blocks; NKVT: 16-byte chunks
__global__ __launch_bounds__(256) void decode_layers_kernel(const KmbDecodeLayers a) {   // per thread of the staged cross keys (4 * S * 8 <= 256 NKVT)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles = (a.R + RT - 1) / RT;
  // block -> (slot, row tile): slots 0 .. 7 live on XCD = slot (blocks b and b + 8 share an XCD: the tiles that stream one weight
  // slice share an L2), slots 8 .. 11 take the remaining blocks.  Placement is a speed matter only.
  // (part, nparts): this workgroup's index among the workgroups of its slot on its XCD (the L2 touches are shared among them)
  int slot, tile, part, nparts;
  {
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3, half0 = (tiles + 1) / 2;
    if (i < tiles) { slot = x; tile = i; part = i; nparts = tiles; }
    else {   // slot 8 + x / 2: its first half0 tiles on the even XCD of the pair, the rest on the odd one (padding blocks exit)
      const int j = i - tiles;
      slot = 8 + (x >> 1);
      tile = (x & 1) ? half0 + j : j;
      part = j; nparts = (x & 1) ? tiles - half0 : half0;
      if (j >= nparts) return;
    }
  }
  if (slot >= DL_SLOTS || tile >= tiles) return;
  constexpr int F = NT5 * KBLK, nt5 = NT5;
  const int row0 = tile * RT, d = KBLK, S = a.S, Tk = a.Tk;
  const DLLds L = dl_lds(a.Tmax, S, F);
  const int a_stride = (KBLK + 8) * 2;
  char* const lds_a = smem + L.a;
  char* const lds_q = smem + L.q;
  float* const sc = reinterpret_cast<float*>(smem + L.sc);
  char* const lds_x = smem + L.x;
  char* const lds_out = smem + L.out;
  char* const lds_dummy = smem + L.dummy;
  constexpr unsigned TILE_B = KBLK * 32;   // bytes of one 16-column weight tile x 768 K in fragment order
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.x_in, 0, a.R * d * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)a.o, 0, a.R * d * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc((void*)a.z, 0, a.R * d * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)a.hh, 0, a.R * F * 2, 0x00020000);
  const int h = slot;
  auto tile_of = [&](int t) { return ((t >> 2) * d + h * HD) / 16 + (t & 3); };   // self q|k|v tile t (0 .. 11) of head h
  WBlock w[NT5 > 3 ? NT5 : 3];
  // ---- the first layer's self-attention weights (its input rows are requested at the top of the loop)
  // (two of a wave's three tiles: 192 registers; the accumulator half of the register file holds 256 and the LayerNorm needs the rest)
#pragma unroll
  for (int t = 0; t < 2; ++t) load_wblock(w[t], a.L[0].Wqkv, d, tile_of(wave * 3 + t), 0, (int)(threadIdx.x & 63));

  for (int l = 0;; ++l) {   // (left by a `break` in front of the next layer's prefetch: a conditional prefetch would keep the old rows live through the whole body)
    const KmbDecodeLayerP& P = a.L[l];
    unsigned* const bar = a.bars + ((size_t)l * 6) * tiles + tile;   // counter of phase p: bar[p * tiles]
    // every per-lane address below derives from `tid`, which is made opaque once per layer: hipcc otherwise hoists ~100 lane-
    // constant address computations out of the loop and SPILLS them -- and a scratch reload is a vector-memory load whose wait
    // also drains the weight fragments in flight (cdna_hip_programming.md, "a lane-constant address hoisted to kernel entry")
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, r = lane & 15, g = lane >> 4;
    // ================================================================= phase 1: [LayerNorm ->] q|k|v of head h -> cache append -> attention
    {
      // the layer's input rows: the launch's input, or the previous layer's output behind its last barrier (the weights of this
      // phase were requested before that wait)
      DLSTAMP(0);
      if (l > 0) group_wait(bar - tiles, a.status);
      DLSTAMP(1);
      // history index of the self-attention cache (a.hist; the same for every layer): the cache rows of the prefetched keys / values
      constexpr int KU = 5, VU = 10;
      const int a_lr = tid >> 4, a_s = tid & 15;
      int hk[KU], hv[VU];
      {
        const int prow = row0 + a_lr < a.R ? row0 + a_lr : 0, tc = Tk - 1;
        const int32_t* hrow = a.hist != nullptr ? a.hist + (size_t)prow * a.Tmax : nullptr;
#pragma unroll
        for (int u = 0; u < KU; ++u) { const int t = (a_s >> 2) + 4 * u; const int v = hrow != nullptr ? hrow[t < tc ? t : 0] : prow; hk[u] = t < tc ? v : prow; }
#pragma unroll
        for (int u = 0; u < VU; ++u) { const int v = hrow != nullptr ? hrow[u < tc ? u : 0] : prow; hv[u] = u < tc ? v : prow; }
      }
      {
        RowRegs<2> rr;
        rows_issue_sc1<2>(rr, l == 0 ? rs_in : rs_z, d, row0, a.R, d, P.lnin_g, P.lnin_b, wave, lane);
#ifdef KMB_DECODE_STAMP
        DLSTAMP(25);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DLSTAMP(26);
#endif
        rows_finish_lean(rr, row0, a.R, P.lnin_g, a.eps, lds_a, a_stride, wave, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_wblock(w[2], P.Wqkv, d, tile_of(wave * 3 + 2), 0, lane);   // the third tile: behind the LayerNorm, lands under the first two tiles' MFMAs
      f32x4 bias[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int tl = wave * 3 + t;
        bias[t] = *reinterpret_cast<const f32x4*>(P.bqkv + (tl >> 2) * d + h * HD + (tl & 3) * 16 + g * 4);
      }
      u32x4 pk0[KU], pk1[KU];
      uint2 pv[VU];
      {
        const int tc = Tk - 1;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int t = (a_s >> 2) + 4 * u, tt = t < tc ? t : 0;
          const bf16_t* kr = P.Kc + ((size_t)hk[u] * a.Tmax + tt) * d + h * HD + (a_s & 3) * 16;
          pk0[u] = *reinterpret_cast<const u32x4*>(kr);
          pk1[u] = *reinterpret_cast<const u32x4*>(kr + 8);
        }
#pragma unroll
        for (int u = 0; u < VU; ++u) pv[u] = *reinterpret_cast<const uint2*>(P.Vc + ((size_t)hv[u] * a.Tmax + (u < tc ? u : 0)) * d + h * HD + a_s * 4);
      }
      DLSTAMP(2);
      __syncthreads();
      {
        // L2 touches for what comes next (issued behind the barrier that follows the LayerNorm: they fly under the MFMAs and the attention; in front of it the rows' wait would wait for them too): the slot's out-projection weights,
        // and the cross-attention keys / values / of this tile's batch items (phase 3; this workgroup's own 128-byte pieces)
        touch_share(P.Wo + (size_t)slot * 4 * (TILE_B / 2), 4 * TILE_B, part, nparts, lds_dummy, tid);
        {
          const int last_row_ = row0 + RT - 1 < a.R ? row0 + RT - 1 : a.R - 1;
          const int first_item_ = row0 / a.kv_group, kv_rows_ = (last_row_ / a.kv_group - first_item_ + 1) * S;
          char* const dst = lds_dummy + (tid >> 6) * 256;
          for (int rw = tid; rw < kv_rows_; rw += 256) {
            const size_t off = ((size_t)first_item_ * S + rw) * a.ldc + h * HD;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(P.cK + off), (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(P.cV + off), (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
          }
        }
      }
      constexpr int QS = (3 * HD + 8) * 2;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mma_wblock(acc, w[t], lds_a, a_stride, 0, lane);
        const int tl = wave * 3 + t, part = tl >> 2;
        const int col = (tl & 3) * 16 + g * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[e] + bias[t][e];
        if (part == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= a.q_scale;
        }
        const uint2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *reinterpret_cast<uint2*>(lds_q + r * QS + (part * HD + col) * 2) = pk;
        if (part > 0 && row0 + r < a.R) {   // append to the cache
          bf16_t* dst = (part == 1 ? P.Kc : P.Vc) + ((size_t)(row0 + r) * a.Tmax + (Tk - 1)) * d + h * HD + col;
          *reinterpret_cast<uint2*>(dst) = pk;
          if (a.hist != nullptr && l == 0 && h == 0 && tl == 4 && g == 0) a.hist[(size_t)(row0 + r) * a.Tmax + (Tk - 1)] = row0 + r;
        }
      }
      __syncthreads();
      DLSTAMP(3);
      // ---- attention over the cache + the new key / value (16 lanes per row), as decode_attn_kernel<SELF>
      const int lr = a_lr, s = a_s;
      const int row = row0 + lr < a.R ? row0 + lr : 0;   // (rows past R compute on row 0's cache and are not published)
      const bf16_t* Kc = P.Kc + (size_t)row * a.Tmax * d + h * HD;
      const bf16_t* Vc = P.Vc + (size_t)row * a.Tmax * d + h * HD;
      const int kq = s >> 2, part = s & 3;
      float qp[16];
      unpack8(*reinterpret_cast<const u32x4*>(lds_q + lr * QS + part * 32), qp);
      unpack8(*reinterpret_cast<const u32x4*>(lds_q + lr * QS + part * 32 + 16), qp + 8);
      const int Tc = Tk - 1;
      float* const my = sc + (size_t)lr * Tk;
      float mx = -INFINITY;
      for (int t0 = 0; t0 < Tc; t0 += 4 * KU) {
        u32x4 k0[KU], k1[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int t = t0 + kq + 4 * u;
          const bool ok = t < Tc;
          if (t0 == 0) { k0[u] = pk0[u]; k1[u] = pk1[u]; }
          else if (a.hist != nullptr) {
            const int tt = ok ? t : 0;
            const bf16_t* kr = P.Kc + ((size_t)a.hist[(size_t)row * a.Tmax + tt] * a.Tmax + tt) * d + h * HD + part * 16;
            k0[u] = *reinterpret_cast<const u32x4*>(kr);
            k1[u] = *reinterpret_cast<const u32x4*>(kr + 8);
          } else {
            const bf16_t* kr = Kc + (size_t)(ok ? t : 0) * d + part * 16;
            k0[u] = *reinterpret_cast<const u32x4*>(kr);
            k1[u] = *reinterpret_cast<const u32x4*>(kr + 8);
          }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int t = t0 + kq + 4 * u;
          float k8[16];
          unpack8(k0[u], k8);
          unpack8(k1[u], k8 + 8);
          float dot = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) dot += qp[e] * k8[e];
          dot += __shfl_xor(dot, 1, 16);
          dot += __shfl_xor(dot, 2, 16);
          if (t < Tc) {
            if (part == 0) my[t] = dot;
            mx = fmaxf(mx, dot);
          }
        }
      }
      {   // the new key: from the projection output in LDS, 4 elements per lane
        float q4[4], k4[4];
        const uint2 qv = *reinterpret_cast<const uint2*>(lds_q + lr * QS + s * 8);
        const uint2 kv = *reinterpret_cast<const uint2*>(lds_q + lr * QS + (HD + s * 4) * 2);
        q4[0] = lo_bf(qv.x); q4[1] = hi_bf(qv.x); q4[2] = lo_bf(qv.y); q4[3] = hi_bf(qv.y);
        k4[0] = lo_bf(kv.x); k4[1] = hi_bf(kv.x); k4[2] = lo_bf(kv.y); k4[3] = hi_bf(kv.y);
        const float dot = group16_sum((q4[0] * k4[0] + q4[1] * k4[1]) + (q4[2] * k4[2] + q4[3] * k4[3]));
        if (s == 0) my[Tc] = dot;
        mx = fmaxf(mx, dot);
      }
      mx = group16_max(mx);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float lsum = 0.f;
      for (int t = s; t < Tk; t += 16) {
        const float e = (mx == -INFINITY) ? 0.f : __expf(my[t] - mx);
        my[t] = e;
        lsum += e;
      }
      lsum = group16_sum(lsum);
      const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float ac[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) ac[i][e] = 0.f;
      for (int t0 = 0; t0 < Tc; t0 += VU) {
        uint2 vv[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) {
          const int t = t0 + u < Tc ? t0 + u : 0;
          if (t0 == 0) { vv[u] = pv[u]; continue; }
          vv[u] = a.hist != nullptr ? *reinterpret_cast<const uint2*>(P.Vc + ((size_t)a.hist[(size_t)row * a.Tmax + t] * a.Tmax + t) * d + h * HD + s * 4)
                                    : *reinterpret_cast<const uint2*>(Vc + (size_t)t * d + s * 4);
        }
#pragma unroll
        for (int u = 0; u < VU; ++u) {
          const float wgt = t0 + u < Tc ? my[t0 + u] : 0.f;
          ac[u & 1][0] += wgt * lo_bf(vv[u].x); ac[u & 1][1] += wgt * hi_bf(vv[u].x);
          ac[u & 1][2] += wgt * lo_bf(vv[u].y); ac[u & 1][3] += wgt * hi_bf(vv[u].y);
        }
      }
      {
        const uint2 vv = *reinterpret_cast<const uint2*>(lds_q + lr * QS + (2 * HD + s * 4) * 2);
        const float wgt = my[Tc];
        ac[1][0] += wgt * lo_bf(vv.x); ac[1][1] += wgt * hi_bf(vv.x); ac[1][2] += wgt * lo_bf(vv.y); ac[1][3] += wgt * hi_bf(vv.y);
      }
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (ac[0][e] + ac[1][e]) * inv;
      *reinterpret_cast<uint2*>(lds_out + lr * 128 + s * 8) = uint2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
      DLSTAMP(4);
      publish_tile<64>(lds_out, rs_o, d, row0, a.R, h * HD, bar + 0 * tiles, tid);
      DLSTAMP(5);
    }
    // ================================================================= phase 2: self-attention output projection + residual
    // requested behind the publish: this phase's weight tile, then the cross-attention keys / values / mask of the tile's batch
    // items (they do not depend on this step at all), which phase 3 stages in LDS
    __builtin_amdgcn_sched_barrier(0);
    load_wblock(w[0], P.Wo, d, slot * 4 + wave, 0, lane);
    f32x4 bias1 = *reinterpret_cast<const f32x4*>(P.bo + (slot * 4 + wave) * 16 + g * 4);
    group_wait(bar + 0 * tiles, a.status);
    DLSTAMP(6);
    {
      RowRegs<2> ro;
      rows_issue_sc1<2>(ro, rs_o, d, row0, a.R, d, nullptr, nullptr, wave, lane);
      rows_finish_lean(ro, row0, a.R, nullptr, a.eps, lds_x, a_stride, wave, lane);
      __syncthreads();
      touch_share(P.Wcq + (size_t)slot * 4 * (TILE_B / 2), 4 * TILE_B, part, nparts, lds_dummy, tid);
      DLSTAMP(7);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      mma_wblock(acc, w[0], lds_x, a_stride, 0, lane);
      proj_epilogue_lds<false>(acc, bias1, lds_a, a_stride, slot * 64 + wave * 16, lds_out, 128, wave * 16, lane);
      DLSTAMP(8);
      publish_tile<64>(lds_out, rs_z, d, row0, a.R, slot * 64, bar + 1 * tiles, tid);
      DLSTAMP(9);
    }
    // ================================================================= phase 3: LayerNorm -> q of head h -> cross-attention (matrix cores)
    __builtin_amdgcn_sched_barrier(0);
    load_wblock(w[0], P.Wcq, d, (h * HD) / 16 + wave, 0, lane);
    // ... and the cross-attention keys / values / mask of the tile's batch items (they do not depend on this step at all): they
    // land while the barrier's hand-off passes and are staged in LDS below
    // The keys / values go global -> LDS directly (LDS-DMA, 16 bytes per lane, no staging registers: the 64 registers they took
    // were spilled, and a scratch reload is a memory round trip): a wave instruction fills eight 128-byte rows; the key image is
    // swizzled at the SOURCE (row's chunk c holds the key's chunk c ^ (row & 7): conflict-free ds_read_b128 fragments), the value
    // image is linear (transposed reads).  The o rows of phase 2 in this region are done with (every wave is past its MFMAs).
    long long mreg[2] = {1, 1};
    const int last_row = row0 + RT - 1 < a.R ? row0 + RT - 1 : a.R - 1;
    const int first_item = row0 / a.kv_group;
    const int kv_rows = (last_row / a.kv_group - first_item + 1) * S;
    {
      char* const lds_k = smem + L.k;
      char* const lds_v = smem + L.v;
      int rowi = tid >> 3;
      int li = rowi / S, t = rowi - li * S;
      const int seg = tid & 7;
#pragma unroll
      for (int i = 0; i < NKVT; ++i) {
        if (rowi < kv_rows) {
          const size_t off = ((size_t)(first_item + li) * S + t) * a.ldc + h * HD;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(P.cK + off + ((seg ^ (rowi & 7)) * 8)),
                                           (__attribute__((address_space(3))) void*)(lds_k + (wave * 8 + 32 * i) * 128), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(P.cV + off + seg * 8),
                                           (__attribute__((address_space(3))) void*)(lds_v + (wave * 8 + 32 * i) * 128), 16, 0, 0);
        }
        rowi += 32; t += 32;
        while (t >= S) { t -= S; ++li; }
      }
      if (a.key_mask != nullptr) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int mr = tid + 256 * i;
          if (mr < kv_rows) mreg[i] = a.key_mask[(size_t)(first_item + mr / S) * a.mask_ld + mr % S];
        }
      }
    }
    {
      const f32x4 biasq = *reinterpret_cast<const f32x4*>(P.bcq + h * HD + wave * 16 + g * 4);
      group_wait(bar + 1 * tiles, a.status);
      DLSTAMP(10);
      RowRegs<2> rz;
      rows_issue_sc1<2>(rz, rs_z, d, row0, a.R, d, P.ln1_g, P.ln1_b, wave, lane);
      char* const lds_k = smem + L.k;
      char* const lds_v = smem + L.v;
      float* const lds_m = reinterpret_cast<float*>(smem + L.m);
      {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (tid + 256 * i < kv_rows) lds_m[tid + 256 * i] = mreg[i] == 0 ? -INFINITY : 0.f;
      }
      rows_finish_lean(rz, row0, a.R, P.ln1_g, a.eps, lds_a, a_stride, wave, lane);
      DLSTAMP(11);
      __syncthreads();
      touch_share(P.Wco + (size_t)slot * 4 * (TILE_B / 2), 4 * TILE_B, part, nparts, lds_dummy, tid);
      constexpr int QS = (HD + 8) * 2;
      {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mma_wblock(acc, w[0], lds_a, a_stride, 0, lane);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc[e] + biasq[e]) * a.q_scale;
        *reinterpret_cast<uint2*>(lds_q + r * QS + (wave * 16 + g * 4) * 2) = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      }
      __syncthreads();
      const int kpad = (kv_rows + 31) & ~31, PBS = pb_stride(S);
      char* const pb_hi = smem + L.pbh;
      char* const pb_lo = smem + L.pbl;
      float* const inv_s = reinterpret_cast<float*>(smem + L.inv);
      const int t2 = 2 * S, t3 = 3 * S;
      {
        const int rrow = row0 + r < a.R ? row0 + r : a.R - 1;
        const int item_r = rrow / a.kv_group - first_item;
        const bf16x8 q0 = *reinterpret_cast<const bf16x8*>(lds_q + r * QS + g * 16);
        const bf16x8 q1 = *reinterpret_cast<const bf16x8*>(lds_q + r * QS + 64 + g * 16);
        for (int kt = wave; kt * 16 < kv_rows; kt += 4) {
          const int key = kt * 16 + r < kv_rows ? kt * 16 + r : 0;
          const char* kr = lds_k + (size_t)key * 128;
          const int c0 = (g ^ (key & 7)) * 16;          // chunks g and g + 4 of the key, through the image's swizzle
          f32x4 sa = {0.f, 0.f, 0.f, 0.f};
          sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kr + c0), q0, sa, 0, 0, 0);
          sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kr + (c0 ^ 64)), q1, sa, 0, 0, 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kidx = kt * 16 + 4 * g + e;
            const int it = (kidx >= S) + (kidx >= t2) + (kidx >= t3);
            if (kidx < kv_rows && it == item_r) sc[(size_t)r * S + (kidx - it * S)] = sa[e] + lds_m[kidx];
          }
        }
      }
      __syncthreads();
      {
        const int lr = tid >> 4, s = tid & 15;
        const int rrow = row0 + lr < a.R ? row0 + lr : a.R - 1;
        const int lo_k = (rrow / a.kv_group - first_item) * S, hi_k = lo_k + S;
        const float* my = sc + (size_t)lr * S;
        float mx = -INFINITY;
        for (int t = s; t < S; t += 16) mx = fmaxf(mx, my[t]);
        mx = row16_max(mx);
        float lsum = 0.f;
        for (int c = s * 8; c < kpad; c += 128) {
          float e[8], rem[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) e[i] = 0.f;
          if (c + 8 > lo_k && c < hi_k && mx != -INFINITY) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
              if (c + i >= lo_k && c + i < hi_k) e[i] = __expf(my[c + i - lo_k] - mx);
          }
          const u32x4 hi = pack8(e);
          float back[8];
          unpack8(hi, back);
#pragma unroll
          for (int i = 0; i < 8; ++i) { lsum += e[i]; rem[i] = e[i] - back[i]; }
          *reinterpret_cast<u32x4*>(pb_hi + lr * PBS + c * 2) = hi;
          *reinterpret_cast<u32x4*>(pb_lo + lr * PBS + c * 2) = pack8(rem);
        }
        lsum = row16_sum(lsum);
        if (s == 0) inv_s[lr] = lsum > 0.f ? 1.f / lsum : 0.f;
      }
      __syncthreads();
      f32x4 oa = {0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < kpad; k0 += 32) {
        bf16x8 vf;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          int krow = k0 + g * 8 + hh * 4 + (r >> 2);
          krow = krow < kv_rows ? krow : kv_rows - 1;
          const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(lds_v + (size_t)krow * 128 + wave * 32 + (r & 3) * 8));
          vf[hh * 4 + 0] = t[0]; vf[hh * 4 + 1] = t[1]; vf[hh * 4 + 2] = t[2]; vf[hh * 4 + 3] = t[3];
        }
        const bf16x8 ph = *reinterpret_cast<const bf16x8*>(pb_hi + r * PBS + (k0 + g * 8) * 2);
        const bf16x8 pl = *reinterpret_cast<const bf16x8*>(pb_lo + r * PBS + (k0 + g * 8) * 2);
        oa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, ph, oa, 0, 0, 0);
        oa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pl, oa, 0, 0, 0);
      }
      {
        const float inv = inv_s[r];
        *reinterpret_cast<uint2*>(lds_out + r * 128 + (wave * 16 + g * 4) * 2) =
            uint2{pack2bf(oa[0] * inv, oa[1] * inv), pack2bf(oa[2] * inv, oa[3] * inv)};
      }
      DLSTAMP(12);
      publish_tile<64>(lds_out, rs_o, d, row0, a.R, h * HD, bar + 2 * tiles, tid);
      DLSTAMP(13);
    }
    // ================================================================= phase 4: cross-attention output projection + residual
    __builtin_amdgcn_sched_barrier(0);
    load_wblock(w[0], P.Wco, d, slot * 4 + wave, 0, lane);
    {
      const f32x4 bias2 = *reinterpret_cast<const f32x4*>(P.bco + (slot * 4 + wave) * 16 + g * 4);
      group_wait(bar + 2 * tiles, a.status);
      DLSTAMP(14);
      RowRegs<2> ro;
      rows_issue_sc1<2>(ro, rs_o, d, row0, a.R, d, nullptr, nullptr, wave, lane);
      rows_finish_lean(ro, row0, a.R, nullptr, a.eps, lds_x, a_stride, wave, lane);
      __syncthreads();
      touch_share(P.W1 + (size_t)slot * 4 * NT5 * (TILE_B / 2), 4 * NT5 * TILE_B, part, nparts, lds_dummy, tid);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      mma_wblock(acc, w[0], lds_x, a_stride, 0, lane);
      proj_epilogue_lds<false>(acc, bias2, lds_a, a_stride, slot * 64 + wave * 16, lds_out, 128, wave * 16, lane);
      DLSTAMP(15);
      publish_tile<64>(lds_out, rs_z, d, row0, a.R, slot * 64, bar + 3 * tiles, tid);
      DLSTAMP(16);
    }
    // ================================================================= phase 5: LayerNorm -> fc1 + GeLU (F / 12 columns of this slot)
    __builtin_amdgcn_sched_barrier(0);
    constexpr int ncol5 = F / DL_SLOTS;             // 64 nt5 columns; wave w owns 16-column tiles (slot * 4 + w) * nt5 ..
    // (half of the slot's weight tiles before the wait, the rest behind the LayerNorm, when its registers are free: all of
    //  them + the staged rows + the LayerNorm's temporaries do not fit 512 registers)
    constexpr int NPRE = NT5 > 1 ? NT5 / 2 : 1;
#pragma unroll
    for (int t = 0; t < NPRE; ++t) load_wblock(w[t], P.W1, d, (slot * 4 + wave) * nt5 + t, 0, lane);
    {
      group_wait(bar + 3 * tiles, a.status);
      DLSTAMP(17);
      {
        RowRegs<2> rz;
        rows_issue_sc1<2>(rz, rs_z, d, row0, a.R, d, P.ln2_g, P.ln2_b, wave, lane);
        rows_finish_lean(rz, row0, a.R, P.ln2_g, a.eps, lds_a, a_stride, wave, lane);
      }
      DLSTAMP(18);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = NPRE; t < NT5; ++t) load_wblock(w[t], P.W1, d, (slot * 4 + wave) * nt5 + t, 0, lane);
      f32x4 bias5[NT5];
#pragma unroll
      for (int t = 0; t < NT5; ++t) bias5[t] = *reinterpret_cast<const f32x4*>(P.b1 + ((slot * 4 + wave) * nt5 + t) * 16 + g * 4);
      __syncthreads();
      touch_share(P.W2 + (size_t)slot * 4 * NT5 * (TILE_B / 2), 4 * NT5 * TILE_B, part, nparts, lds_dummy, tid);
#pragma unroll
      for (int t = 0; t < NT5; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mma_wblock(acc, w[t], lds_a, a_stride, 0, lane);
        proj_epilogue_lds<true>(acc, bias5[t], nullptr, 0, 0, lds_out, ncol5 * 2, (wave * nt5 + t) * 16, lane);
      }
      DLSTAMP(19);
      publish_tile<ncol5>(lds_out, rs_h, F, row0, a.R, slot * ncol5, bar + 4 * tiles, tid);
      DLSTAMP(20);
    }
    // ================================================================= phase 6: fc2 (K = F) + residual -> the layer's pre-LayerNorm output
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < NT5; ++b) load_wblock(w[b], P.W2, F, slot * 4 + wave, b * KBLK, lane);
    {
      const f32x4 bias6 = *reinterpret_cast<const f32x4*>(P.b2 + (slot * 4 + wave) * 16 + g * 4);
      group_wait(bar + 4 * tiles, a.status);
      DLSTAMP(21);
      constexpr int h_stride = (F + 8) * 2;
      {   // 16 rows x F hidden activations -> LDS (wave w: rows 4 w ..), 16-byte sc1 loads, F / 8 chunks per row
        constexpr int nch = F >> 3, NJ = (nch + 63) / 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rowl = wave * 4 + i, row = row0 + rowl;
          u32x4 v[NJ];
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int c = lane + 64 * j;
            v[j] = ld16_sc1(rs_h, (row < a.R && c < nch) ? ((unsigned)row * (unsigned)F + (unsigned)c * 8u) * 2u : 0u);
          }
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int c = lane + 64 * j;
            if (c < nch) *reinterpret_cast<u32x4*>(lds_x + rowl * h_stride + c * 16) = row < a.R ? v[j] : u32x4{0u, 0u, 0u, 0u};
          }
        }
      }
      __syncthreads();
      if (l + 1 < a.n_layers) {   // the next layer's q | k | v tiles of this head: three 4-tile regions
        const KmbDecodeLayerP& Pn = a.L[l + 1];
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
          touch_share(Pn.Wqkv + (size_t)((pt * d + h * HD) / 16) * (TILE_B / 2), 4 * TILE_B, part, nparts, lds_dummy, tid);
      }
      DLSTAMP(22);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < NT5; ++b) mma_wblock(acc, w[b], lds_x, h_stride, b * KBLK, lane);
      proj_epilogue_lds<false>(acc, bias6, lds_a, a_stride, slot * 64 + wave * 16, lds_out, 128, wave * 16, lane);
      DLSTAMP(23);
      publish_tile<64>(lds_out, rs_z, d, row0, a.R, slot * 64, bar + 5 * tiles, tid);
      DLSTAMP(24);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (l + 1 >= a.n_layers) break;
    {   // the next layer's self-attention weights (its input rows follow behind this layer's last barrier, at the top of the loop)
      const KmbDecodeLayerP& Pn = a.L[l + 1];
#pragma unroll
      for (int t = 0; t < 2; ++t) load_wblock(w[t], Pn.Wqkv, d, tile_of(wave * 3 + t), 0, lane);
    }
  }
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
  if (p.gamma && p.K != KBLK) return "decode block: the LayerNorm input must be 768 wide";
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
    if (p.kind == 1 && (p.kv_row || p.key_mask || p.kv_group)) return "decode block: the self-attention cache is per row and unmasked";
    if (p.kv_group < 0) return "decode block: kv_group";
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
    static size_t set[4] = {0, 0, 0, 0};
#define KMB_PROJ(SLOT, KERNEL, NTW)                                                                                   \
  do {                                                                                                                \
    if (a_bytes > set[SLOT]) { e = set_lds(KERNEL, a_bytes); if (e != hipSuccess) return e; set[SLOT] = a_bytes; }     \
    hipLaunchKernelGGL(KERNEL, dim3(unit_grid(p.N / (64 * NTW), tiles)), dim3(256), a_bytes, stream, p);              \
  } while (0)
    if (nch <= 2 && (p.N % 128) == 0 && p.N >= 1536) KMB_PROJ(0, decode_proj_wide_kernel, 2);   // half as many LayerNorm prologues
    else if (nch <= 2) KMB_PROJ(1, decode_proj_kernel<2>, 1);
    else if (nch <= 4) KMB_PROJ(2, decode_proj_kernel<4>, 1);
    else KMB_PROJ(3, decode_proj_kernel<6>, 1);
#undef KMB_PROJ
    return hipGetLastError();
  }
  const bool self = p.kind == 1;
  const size_t lds = a_bytes + (size_t)RT * ((self ? 3 * HD : HD) + 8) * 2 + (((size_t)RT * p.Tk * sizeof(float) + 15) & ~(size_t)15);
  const size_t lds_kv = lds + (size_t)KV_ITEMS * p.Tk * (KS + 128 + sizeof(float)) + (size_t)RT * pb_stride(p.Tk) + RT * sizeof(float);
  static size_t set_s = 0, set_c = 0, set_l = 0;
  if (self) {
    if (lds > set_s) { e = set_lds(decode_attn_kernel<true, false>, lds); if (e != hipSuccess) return e; set_s = lds; }
    hipLaunchKernelGGL((decode_attn_kernel<true, false>), dim3(unit_grid(p.H, tiles)), dim3(256), lds, stream, p);
  } else if (p.kv_group >= 4 && KV_ITEMS * p.Tk * 8 <= NKV * 256 && lds_kv <= 160 * 1024) {
    if (lds_kv > set_l) { e = set_lds(decode_attn_kernel<false, true>, lds_kv); if (e != hipSuccess) return e; set_l = lds_kv; }
    hipLaunchKernelGGL((decode_attn_kernel<false, true>), dim3(unit_grid(p.H, tiles)), dim3(256), lds_kv, stream, p);
  } else {
    if (lds > set_c) { e = set_lds(decode_attn_kernel<false, false>, lds); if (e != hipSuccess) return e; set_c = lds; }
    hipLaunchKernelGGL((decode_attn_kernel<false, false>), dim3(unit_grid(p.H, tiles)), dim3(256), lds, stream, p);
  }
  return hipGetLastError();
}

// packed[i] <- fragment-order copy of W[i] ([N, K] row-major with row stride ld; N % 16 == 0, K % 64 == 0), all in ONE
// launch (n <= 48 matrices)
hipError_t kmb_decode_pack_launch(const bf16_t* const* W, const int* ld, const int* N, const int* K, bf16_t* const* packed, int n,
                                  hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (n > PACK_MAX) return hipErrorInvalidValue;
  PackArgs a;
  memset(&a, 0, sizeof(a));
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    if ((N[i] & 15) || (K[i] & 63) || (ld[i] & 7) || ((uintptr_t)W[i] & 15) || ((uintptr_t)packed[i] & 15)) return hipErrorInvalidValue;
    a.m[i] = PackDesc{W[i], packed[i], ld[i], N[i], K[i], (int)total};
    total += (long long)N[i] * K[i] / 8;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
  }
  a.n = n;
  a.total = (int)total;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// ---- resident decoder-layers kernel: host side ----
size_t kmb_decode_layers_lds(int Tmax, int S, int F) { return (size_t)dl_lds(Tmax, S, F).total; }
size_t kmb_decode_layers_bar_words(int R, int n_layers) { return (size_t)n_layers * 6 * ((R + RT - 1) / RT); }

const char* kmb_decode_layers_check(const KmbDecodeLayers& a) {
  const int tiles = (a.R + RT - 1) / RT;
  if (a.n_layers < 1 || a.n_layers > KMB_DL_MAX_LAYERS) return "decode layers: 1 .. 6 layers per launch";
  if (a.H != DL_SLOTS) return "decode layers: d_model must be 768 (12 heads)";
  if (a.F % KBLK || a.F < KBLK || a.F > 4 * KBLK || (a.F != KBLK && a.F != 2 * KBLK && a.F != 4 * KBLK)) return "decode layers: ffn width must be 768, 1536 or 3072";
  if (8 * (tiles + (tiles + 1) / 2) > 256) return "decode layers: more row tiles than co-resident workgroups (R <= 336)";
  if (a.kv_group < 4 || KV_ITEMS * a.S * 8 > NKV * 256) return "decode layers: needs >= 4 rows per batch item and S <= 120";
  if (a.Tk < 1 || a.Tk > a.Tmax) return "decode layers: Tk";
  if (kmb_decode_layers_lds(a.Tmax, a.S, a.F) > 160 * 1024) return "decode layers: LDS image does not fit";
  if (!a.x_in || !a.o || !a.z || !a.hh || !a.bars || !a.status) return "decode layers: missing buffer";
  return nullptr;
}

hipError_t kmb_decode_layers_launch(const KmbDecodeLayers& a, hipStream_t stream) {
  const int tiles = (a.R + RT - 1) / RT;
  const size_t lds = kmb_decode_layers_lds(a.Tmax, a.S, a.F);
  const bool small_s = KV_ITEMS * a.S * 8 <= 8 * 256;   // S <= 64: eight chunks of keys / values per thread instead of fifteen
  void (*kern)(const KmbDecodeLayers) = a.F == 4 * KBLK ? (small_s ? decode_layers_kernel<4, 8> : decode_layers_kernel<4, NKV>)
                                        : a.F == 2 * KBLK ? decode_layers_kernel<2, NKV> : decode_layers_kernel<1, NKV>;
  const int ki = a.F == 4 * KBLK ? (small_s ? 3 : 2) : a.F == 2 * KBLK ? 1 : 0;
  static size_t set[4] = {0, 0, 0, 0};
  static int resident[4] = {-1, -1, -1, -1};   // co-residency is what the group barriers rest on: ask once per kernel, refuse loudly
  hipError_t e;
  if (lds > set[ki]) {
    e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    set[ki] = lds; resident[ki] = -1;
  }
  if (resident[ki] < 0) {
    int per_cu = 0, cus = 0, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, set[ki]) != hipSuccess) per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    resident[ki] = per_cu * cus;
  }
  const int grid = 8 * (tiles + (tiles + 1) / 2);   // slots 0 .. 7: one XCD each; slots 8 .. 11: an XCD pair each (a few padding blocks exit at once)
  if (grid > resident[ki]) return hipErrorCooperativeLaunchTooLarge;
  e = hipMemsetAsync(a.bars, 0, kmb_decode_layers_bar_words(a.R, a.n_layers) * sizeof(unsigned), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
  return hipGetLastError();
}
