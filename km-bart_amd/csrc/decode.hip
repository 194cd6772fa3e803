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

// (The resident decoder-layers kernel of round 5 -- all layers of a decode step in one launch behind 12-workgroup counter barriers, bit-identical to
//  the blocks above and 17 % slower -- lives in tools/experiments/decode_resident.hip since round 6: `python km-bart_amd/build.py --variant resident`
//  builds a library with it, KMB_GEN_FUSED=2 selects it there.  That file includes this one for the device helpers above.)
#ifdef KMB_DECODE_DEVICE_ONLY
}  // namespace
#else
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

#endif  // KMB_DECODE_DEVICE_ONLY
