// Fused scaled-dot-product attention for KM-BART's tiny tiles (head_dim = 64; S <= 64..256).
//
// Semantics (transformers 3.0.2 SelfAttention as used by the reference, src/model/modules.py:84,
// src/model/model.py:35): q already carries the head_dim**-0.5 scale (applied by the projection
// epilogue), scores get -inf at padded keys (key_mask == 0) and above the diagonal (causal),
// softmax in fp32, attention_dropout = 0.
//
// Forward : grid (B*H, ceil(Tq/64)), 4 waves, wave w owns 16 query rows; K/V tiles of 64 keys are
//           staged once per workgroup in LDS; online softmax over key tiles.
// Backward: grid (B*H); loops key tiles (outer) and query tiles (inner); P is recomputed from the
//           saved log-sum-exp; dK/dV live in registers across the inner loop, dQ in LDS (fp32).
// One LDS image per tile serves both MFMA operand shapes: row reads (ds_read_b128) and
// transposed reads (ds_read_b64_tr_b16); the XOR swizzle below makes both conflict-free.
#include <cstdlib>
#include "common.h"
#include "diag.h"
#include "kernels.h"

// Diagnostic build only (tools/attn_bwd_stamps.py: -DKMB_ATTN_STAMP): s_memrealtime stamps (100 MHz) of thread 0 at the phase
// boundaries of every item a workgroup of attn_bwd_small_kernel works through (up to 32 items x 16 slots per workgroup).
#ifdef KMB_ATTN_STAMP
__device__ unsigned long long* g_attn_stamps = nullptr;
extern "C" int kmb_debug_set_attn_stamps(void* p) {
  unsigned long long* v = (unsigned long long*)p;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &v, sizeof(v));
}
#define ASTAMP(i)                                                                                           \
  do {                                                                                                      \
    if (g_attn_stamps != nullptr && threadIdx.x == 0 && astamp_iter < 32)                                   \
      g_attn_stamps[((size_t)blockIdx.x * 32 + astamp_iter) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define ASTAMP(i)
#endif

namespace {

constexpr int HD = 64;          // head dim
constexpr int TILE_BYTES = 64 * 128;

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
// byte offset of 16-byte chunk `c16` (0..7) of row `row` in a [rows][64] bf16 tile
__device__ __forceinline__ int tile_off(int row, int c16) {
  return row * 128 + (((c16 >> 1) ^ swz(row)) << 5) + ((c16 & 1) << 4);
}
__device__ __forceinline__ int elem_off(int row, int col) { return tile_off(row, col >> 3) + ((col & 7) << 1); }

// A/B fragment whose 16 "rows" are tile rows and whose k runs along the 64 columns
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int row16, int kk, int r, int g) {
  return *reinterpret_cast<const bf16x8*>(tile + tile_off(row16 * 16 + r, kk * 4 + g));
}
// A/B fragment whose 16 "rows" are tile COLUMNS col16*16.. and whose k runs along tile rows
__device__ __forceinline__ bf16x8 frag_cols(const char* tile, int col16, int kk, int r, int g) {
  bf16x8 out;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int krow = kk * 32 + g * 8 + hh * 4 + (r >> 2);
    const int off = krow * 128 + ((col16 ^ swz(krow)) << 5) + ((r & 3) << 3);
    const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off));
    out[hh * 4 + 0] = t[0]; out[hh * 4 + 1] = t[1]; out[hh * 4 + 2] = t[2]; out[hh * 4 + 3] = t[3];
  }
  return out;
}

// stage a [64][64] bf16 tile: rows t0.. of X (row stride ld), clamped to the last valid row
__device__ __forceinline__ void stage_tile(char* tile, const bf16_t* __restrict__ X, size_t base_row, int t0, int T,
                                           int ld, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;
    const int row = id >> 3, c = id & 7;
    int t = t0 + row;
    t = t < T ? t : T - 1;
    const u32x4 v = *reinterpret_cast<const u32x4*>(X + (base_row + t) * (size_t)ld + c * 8);
    *reinterpret_cast<u32x4*>(tile + tile_off(row, c)) = v;
  }
}

__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64)); v = fmaxf(v, __shfl_xor(v, 8, 64));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}

// ------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void attn_fwd_kernel(const KmbAttn p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES + 4 * 2048];
  char* Ks = smem;
  char* Vs = smem + TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  char* Ps = smem + 2 * TILE_BYTES + wave * 2048;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int q0 = blockIdx.y * 64 + wave * 16;
  const bf16_t* Qh = p.Q + h * HD;
  const bf16_t* Kh = p.K + h * HD;
  const bf16_t* Vh = p.V + h * HD;

  bf16x8 qf[2];
  {
    int q = q0 + r;
    q = q < p.Tq ? q : p.Tq - 1;
    const bf16_t* qrow = Qh + ((size_t)b * p.Tq + q) * p.ldq;
    qf[0] = *reinterpret_cast<const bf16x8*>(qrow + g * 8);
    qf[1] = *reinterpret_cast<const bf16x8*>(qrow + 32 + g * 8);
  }
  float m_run[4], l_run[4];
  f32x4 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { m_run[i] = -INFINITY; l_run[i] = 0.f; o[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nkt = (p.Tk + 63) / 64;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    stage_tile(Ks, Kh, (size_t)b * p.Tk, kt * 64, p.Tk, p.ldk, tid);
    stage_tile(Vs, Vh, (size_t)b * p.Tk, kt * 64, p.Tk, p.ldv, tid);
    __syncthreads();
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[kk], frag_rows(Ks, j, kk, r, g), s[j], 0, 0, 0);
    // masks
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = kt * 64 + j * 16 + r;
      bool kv = key < p.Tk;
      if (kv && p.key_mask != nullptr) kv = p.key_mask[(size_t)b * p.Tk + key] != 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int qi = q0 + g * 4 + q;
        const bool ok = kv && (!p.causal || key <= qi);
        s[j][q] = ok ? s[j][q] : -INFINITY;
      }
    }
    // online softmax (row = g*4 + q, spread over the 16 lanes of the group)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float mx = fmaxf(fmaxf(s[0][q], s[1][q]), fmaxf(s[2][q], s[3][q]));
      mx = group16_max(mx);
      const float m_new = fmaxf(m_run[q], mx);
      const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run[q] - m_new);
      float lsum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = (m_new == -INFINITY) ? 0.f : __expf(s[j][q] - m_new);
        s[j][q] = pv;
        lsum += pv;
      }
      l_run[q] = l_run[q] * alpha + lsum;
      m_run[q] = m_new;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j][q] *= alpha;
    }
    // P (C layout) -> LDS -> A layout
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<bf16_t*>(Ps + elem_off(g * 4 + q, j * 16 + r)) = f2bf(s[j][q]);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 pf = frag_rows(Ps, 0, kk, r, g);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, frag_cols(Vs, j, kk, r, g), o[j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float l = group16_sum(l_run[q]);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    const int qi = q0 + g * 4 + q;
    if (qi < p.Tq) {
      bf16_t* orow = p.O + ((size_t)b * p.Tq + qi) * p.ldo + h * HD;
#pragma unroll
      for (int j = 0; j < 4; ++j) orow[j * 16 + r] = f2bf(o[j][q] * inv);
      if (r == 0 && p.lse != nullptr)
        p.lse[((size_t)b * p.H + h) * p.Tq + qi] = l > 0.f ? m_run[q] + __logf(l) : -INFINITY;
    }
  }
}

// ------------------------------------------ forward, single-tile shapes (Tq <= 64 and Tk <= 64: every training shape)
// attn_fwd_kernel runs one (batch, head) item per workgroup: load -> barrier -> QK^T -> softmax -> PV -> store, four workgroups
// per CU and nothing in flight while an item is computed (3.5 / 2.5 / 3.2 TB/s of algorithmic bytes on the encoder self /
// decoder self / cross shapes at b = 1024).  Here a workgroup is persistent over items like attn_bwd_small_kernel: the global
// loads of item i+1 (Q, K, V: six 16-byte chunks per thread, in registers) are issued before item i is computed and written
// to LDS after it.  Same arithmetic, operation for operation, as one key tile of attn_fwd_kernel (whose online-softmax
// rescale is exp(-inf - m) = 0 on the first tile): identical bits.  Waves without query rows (Tq = 32: waves 2, 3) only
// stage.
// mk: the key-mask words of this lane's four keys (16 j + lane % 16).  They travel with the item's operands: read where they
// are used -- after the next item's loads have been issued -- each of them queued behind those loads in the in-order return
// path, and the compute of item i waited for the operands of item i + 1 (rounds 2-3: the software pipeline hid nothing
// whenever a padding mask was given).
struct FwdRegs { u32x4 q[2], k[2], v[2]; long long mk; };   // mk: the key-mask word of key `tid` of the tile (threads 0 .. 63 only)

// PACK (Tq, Tk <= 32, even H): two heads of a batch item per tile, as in the backward (bwd_load_item)
template <bool PACK>
__device__ __forceinline__ void fwd_load_item(const KmbAttn& p, int item, int tid, FwdRegs& x) {
  const int HH = PACK ? (p.H >> 1) : p.H;
  const int b = item / HH, h0 = PACK ? 2 * (item % HH) : item % HH;
  // The mask word FIRST, and 32-bit byte offsets as in bwd_load_item (the launcher sends tensors that reach 4 GB to the general kernel).  Late
  // round 5: with 64-bit per-lane addresses the one-head kernel sat at 128 registers with the mask's lane address spilled, and its reload in
  // front of the mask load -- the LAST load of this function -- came with `s_waitcnt vmcnt(0)`: the first wave waited there for the Q / K / V
  // loads it had just issued for the NEXT item before it started the current one, and the other three waited for it at the next barrier.
  if (p.key_mask != nullptr && tid < 64) {   // one load per tile (first wave; key = tid, PACK: keys 32 .. 63 are the second head's = the same batch item's)
    const int key = PACK ? (tid & 31) : tid;
    x.mk = p.key_mask[(uint32_t)b * (uint32_t)p.Tk + (uint32_t)(key < p.Tk ? key : p.Tk - 1)];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;
    const int row = PACK ? ((id >> 3) & 31) : (id >> 3), c = id & 7;
    const int h = PACK ? h0 + i : h0;
    const int tq = row < p.Tq ? row : p.Tq - 1, tk = row < p.Tk ? row : p.Tk - 1;
    const uint32_t rq = (uint32_t)b * (uint32_t)p.Tq + (uint32_t)tq, rk = (uint32_t)b * (uint32_t)p.Tk + (uint32_t)tk;
    const uint32_t hc = (uint32_t)(h * HD + c * 8);
    auto at = [](const bf16_t* base, uint32_t elem) { return reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) + elem * 2u); };
    // rows 32 .. 63 (i = 1) of a tile that holds at most 32 rows are never fetched: their LDS rows stay zero
    if (PACK || i == 0 || p.Tq > 32) x.q[i] = *at(p.Q, rq * (uint32_t)p.ldq + hc);
    if (PACK || i == 0 || p.Tk > 32) {
      x.k[i] = *at(p.K, rk * (uint32_t)p.ldk + hc);
      x.v[i] = *at(p.V, rk * (uint32_t)p.ldv + hc);
    }
  }
}

template <bool PACK>
__global__ __launch_bounds__(256, 4) void attn_fwd_small_kernel(const KmbAttn p) {
  __shared__ __attribute__((aligned(16))) char smem[3 * TILE_BYTES + 4 * 2048];
  __shared__ float msk_s[64];   // key-mask flags of the tile's 64 keys
  char* Qs = smem;
  char* Ks = smem + TILE_BYTES;
  char* Vs = smem + 2 * TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  char* Ps = smem + 3 * TILE_BYTES + wave * 2048;
  const int HH = PACK ? (p.H >> 1) : p.H;
  const int nitems = p.B * HH;
  const int q0 = wave * 16;
  const bool lse_vec = (p.Tq & 3) == 0 && ((uintptr_t)p.lse & 15) == 0;   // a lane's four log-sum-exps are one aligned 16-byte piece
  int item = blockIdx.x;
  if (item >= nitems) return;
  FwdRegs x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {   // (masked keys multiply a probability of exactly 0: their rows must be finite, not just ignored)
    const u32x4 z = {0u, 0u, 0u, 0u};
    x.q[i] = z; x.k[i] = z; x.v[i] = z;
  }
  x.mk = 1;
  fwd_load_item<PACK>(p, item, tid, x);
  for (; item < nitems; item += gridDim.x) {
    const int b = item / HH, h = PACK ? 2 * (item % HH) : item % HH;   // PACK: the first of the tile's two heads
    __syncthreads();   // everyone is done with the previous item's LDS images
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // (the never-fetched halves are rewritten with their zeros: LDS stores are not the limit)
      const int id = tid + 256 * i;
      const int row = id >> 3, c = id & 7;
      *reinterpret_cast<u32x4*>(Qs + tile_off(row, c)) = x.q[i];
      *reinterpret_cast<u32x4*>(Ks + tile_off(row, c)) = x.k[i];
      *reinterpret_cast<u32x4*>(Vs + tile_off(row, c)) = x.v[i];
    }
    if (tid < 64) msk_s[tid] = x.mk != 0 ? 1.f : 0.f;
    __syncthreads();
    bool key_on[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) key_on[j] = msk_s[j * 16 + r] != 0.f;
    const int nxt = item + (int)gridDim.x;   // the next item's loads go out now and land while this one is computed
    if (nxt < nitems) fwd_load_item<PACK>(p, nxt, tid, x);
    if ((PACK ? (q0 & 31) : q0) >= p.Tq) continue;   // wave-uniform: this wave has no query rows (both barriers are at the loop head)
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 qf = frag_rows(Qs, wave, kk, r, g);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, frag_rows(Ks, j, kk, r, g), s[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = j * 16 + r;
      const int keyl = PACK ? (key & 31) : key;
      const bool kv = keyl < p.Tk && key_on[j];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int qi = q0 + g * 4 + q;
        const bool ok = kv && (!p.causal || keyl <= (PACK ? (qi & 31) : qi)) && (!PACK || (key >> 5) == (qi >> 5));   // PACK: own head's keys only
        s[j][q] = ok ? s[j][q] : -INFINITY;
      }
    }
    float m_run[4], l_run[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float mx = fmaxf(fmaxf(s[0][q], s[1][q]), fmaxf(s[2][q], s[3][q]));
      mx = group16_max(mx);
      float lsum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = (mx == -INFINITY) ? 0.f : __expf(s[j][q] - mx);
        s[j][q] = pv;
        lsum += pv;
      }
      l_run[q] = lsum;
      m_run[q] = mx;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<bf16_t*>(Ps + elem_off(g * 4 + q, j * 16 + r)) = f2bf(s[j][q]);
    // Ps is this wave's own: its LDS operations execute in order, a wave-level fence keeps the compiler from moving the reads up
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x4 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 pf = frag_rows(Ps, 0, kk, r, g);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, frag_cols(Vs, j, kk, r, g), o[j], 0, 0, 0);
    }
    // O leaves through the wave's own P image (free now): whole 128-byte rows, 16 bytes per lane -- two store instructions per wave where the
    // accumulator layout (a lane holds ONE column of four rows per MFMA tile) needed sixteen 2-byte ones (round 5, as in the backward)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the P fragments have been read
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x4 lv;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float l = group16_sum(l_run[q]);
      const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16_t*>(Ps + elem_off(g * 4 + q, j * 16 + r)) = f2bf(o[j][q] * inv);
      lv[q] = l > 0.f ? m_run[q] + __logf(l) : -INFINITY;
    }
    if (r == 0 && p.lse != nullptr) {   // the log-sum-exps of this lane's four rows (consecutive queries of one head): ONE 16-byte store where the
      const int qt = q0 + g * 4;        // layout allows it instead of four 4-byte ones (a wave issued 4 + 2 stores per item, now 1 + 2)
      const int qi = PACK ? (qt & 31) : qt, hq = PACK ? h + (qt >> 5) : h;   // the first query inside its head, and that head
      float* dst = p.lse + ((size_t)b * p.H + hq) * p.Tq + qi;
      if (lse_vec) {
        if (qi < p.Tq) *reinterpret_cast<f32x4*>(dst) = lv;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (qi + q < p.Tq) dst[q] = lv[q];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = (lane >> 3) + 8 * i, c = lane & 7;   // row of the wave's 16, 16-byte chunk of its 128 bytes
      const int qt = q0 + lr;
      const int qi = PACK ? (qt & 31) : qt, hq = PACK ? h + (qt >> 5) : h;
      const u32x4 v = *reinterpret_cast<const u32x4*>(Ps + tile_off(lr, c));
      if (qi < p.Tq) *reinterpret_cast<u32x4*>(p.O + ((size_t)b * p.Tq + qi) * p.ldo + hq * HD + c * 8) = v;
    }
  }
}

// ----------------------------------------------------------------- backward
__global__ __launch_bounds__(256) void attn_bwd_kernel(const KmbAttn p, int nqt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* dOs = smem + TILE_BYTES;
  char* Ks = smem + 2 * TILE_BYTES;
  char* Vs = smem + 3 * TILE_BYTES;
  char* Ps = smem + 4 * TILE_BYTES;
  char* dSs = smem + 5 * TILE_BYTES;
  float* lse_s = reinterpret_cast<float*>(smem + 6 * TILE_BYTES);
  float* del_s = lse_s + 64;
  float* colk = del_s + 64;   // [4 waves][64] column sums of dK over each wave's keys (bias-gradient partials)
  float* colv = colk + 256;   // [4 waves][64] same for dV
  float* dQacc = colv + 256;  // [nqt*64][64] fp32
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const bf16_t* Qh = p.Q + h * HD;
  const bf16_t* Kh = p.K + h * HD;
  const bf16_t* Vh = p.V + h * HD;
  const bf16_t* Oh = p.O + h * HD;
  const bf16_t* dOh = p.dO + h * HD;

  const int nkt = (p.Tk + 63) / 64;
  for (int i = tid; i < 512; i += 256) colk[i] = 0.f;  // colk and colv are adjacent
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    stage_tile(Ks, Kh, (size_t)b * p.Tk, kt * 64, p.Tk, p.ldk, tid);
    stage_tile(Vs, Vh, (size_t)b * p.Tk, kt * 64, p.Tk, p.ldv, tid);
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { dk[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    for (int qt = 0; qt < nqt; ++qt) {
      __syncthreads();
      stage_tile(Qs, Qh, (size_t)b * p.Tq, qt * 64, p.Tq, p.ldq, tid);
      stage_tile(dOs, dOh, (size_t)b * p.Tq, qt * 64, p.Tq, p.lddo, tid);
      {  // delta = rowsum(dO * O), lse: wave w owns rows w*16..w*16+15 of the tile
        int q = qt * 64 + wave * 16 + r;
        q = q < p.Tq ? q : p.Tq - 1;
        const bf16_t* orow = Oh + ((size_t)b * p.Tq + q) * p.ldo;
        const bf16_t* drow = dOh + ((size_t)b * p.Tq + q) * p.lddo;
        float a8[8], b8[8], acc = 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          unpack8(*reinterpret_cast<const u32x4*>(orow + half * 32 + g * 8), a8);
          unpack8(*reinterpret_cast<const u32x4*>(drow + half * 32 + g * 8), b8);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc += a8[e] * b8[e];
        }
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        if (g == 0) {
          del_s[wave * 16 + r] = acc;
          lse_s[wave * 16 + r] = p.lse[((size_t)b * p.H + h) * p.Tq + q];
        }
      }
      __syncthreads();
      // S = Q K^T and dP = dO V^T for this wave's 16 query rows x 64 keys
      f32x4 s[4], dp[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 qf = frag_rows(Qs, wave, kk, r, g);
        const bf16x8 df = frag_rows(dOs, wave, kk, r, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, frag_rows(Ks, j, kk, r, g), s[j], 0, 0, 0);
          dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, frag_rows(Vs, j, kk, r, g), dp[j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = kt * 64 + j * 16 + r;
        bool kv = key < p.Tk;
        if (kv && p.key_mask != nullptr) kv = p.key_mask[(size_t)b * p.Tk + key] != 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int lrow = wave * 16 + g * 4 + q;
          const int qi = qt * 64 + lrow;
          const bool ok = kv && qi < p.Tq && (!p.causal || key <= qi);
          const float lse = lse_s[lrow];
          const float pv = (ok && lse != -INFINITY) ? __expf(s[j][q] - lse) : 0.f;
          const float ds = pv * (dp[j][q] - del_s[lrow]);
          *reinterpret_cast<bf16_t*>(Ps + elem_off(lrow, j * 16 + r)) = f2bf(pv);
          *reinterpret_cast<bf16_t*>(dSs + elem_off(lrow, j * 16 + r)) = f2bf(ds);
        }
      }
      __syncthreads();
      // dQ[16 rows of this wave] += dS K ; dV[16 keys of this wave] += P^T dO ; dK += dS^T Q
      f32x4 dq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) dq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 dsf = frag_rows(dSs, wave, kk, r, g);
        const bf16x8 ptf = frag_cols(Ps, wave, kk, r, g);
        const bf16x8 dstf = frag_cols(dSs, wave, kk, r, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dq[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, frag_cols(Ks, j, kk, r, g), dq[j], 0, 0, 0);
          // operands swapped: the accumulators hold the TRANSPOSED 16x16 tiles -- lane (r, g) has key r of this wave and
          // the four consecutive feature columns 16 j + 4 g .. +3, so dK / dV leave as 8-byte stores (4 + 4 store
          // instructions per wave instead of 32 + 32 two-byte ones; a store instruction costs the wave ~60 cycles)
          dv[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(dOs, j, kk, r, g), ptf, dv[j], 0, 0, 0);
          dk[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Qs, j, kk, r, g), dstf, dk[j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float* a = dQacc + (size_t)(qt * 64 + wave * 16 + g * 4 + q) * 64 + j * 16 + r;
          *a = (kt == 0) ? dq[j][q] : (*a + dq[j][q]);
        }
    }
    // this wave's 16 keys of dK / dV (transposed tiles: key = lane & 15, columns 16 j + 4 g + q)
    const int key = kt * 64 + wave * 16 + r;
    const bool key_ok = key < p.Tk;
    if (p.dk_colsum != nullptr) {  // uniform branch: column sums over the keys = over the 16 r-lanes of a DPP row
      auto row16_sum = [](float v) {   // inclusive prefix sums by row_shr 1, 2, 4, 8 (zeros shifted in): lane 15 = total
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));
        return v;
      };
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float sk = row16_sum(key_ok ? dk[j][q] : 0.f);
          const float sv = row16_sum(key_ok ? dv[j][q] : 0.f);
          if (r == 15) {   // this wave's own partial row: plain read-modify-write, no atomics, deterministic
            colk[wave * 64 + j * 16 + g * 4 + q] += sk;
            colv[wave * 64 + j * 16 + g * 4 + q] += sv;
          }
        }
    }
    if (key_ok) {
      typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
      bf16_t* kr = p.dK + ((size_t)b * p.Tk + key) * p.lddk + h * HD + g * 4;
      bf16_t* vr = p.dV + ((size_t)b * p.Tk + key) * p.lddv + h * HD + g * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<u32x2*>(kr + j * 16) = u32x2{pack2bf(dk[j][0], dk[j][1]), pack2bf(dk[j][2], dk[j][3])};
        *reinterpret_cast<u32x2*>(vr + j * 16) = u32x2{pack2bf(dv[j][0], dv[j][1]), pack2bf(dv[j][2], dv[j][3])};
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < p.Tq * 8; i += 256) {
    const int q = i >> 3, c = i & 7;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = dQacc[(size_t)q * 64 + c * 8 + e] * p.dq_scale;
    *reinterpret_cast<u32x4*>(p.dQ + ((size_t)b * p.Tq + q) * p.lddq + h * HD + c * 8) = pack8(v);
  }
  if (p.dq_colsum != nullptr && tid < 64) {
    float sq = 0.f;
    for (int q = 0; q < p.Tq; ++q) sq += dQacc[(size_t)q * 64 + tid] * p.dq_scale;
    p.dq_colsum[(size_t)b * p.ld_colsum + h * HD + tid] = sq;
  }
  if (p.dk_colsum != nullptr && tid < 64) {
    p.dk_colsum[(size_t)b * p.ld_colsum + h * HD + tid] = (colk[tid] + colk[64 + tid]) + (colk[128 + tid] + colk[192 + tid]);
    p.dv_colsum[(size_t)b * p.ld_colsum + h * HD + tid] = (colv[tid] + colv[64 + tid]) + (colv[128 + tid] + colv[192 + tid]);
  }
}

// ------------------------------------------ backward, single-tile shapes (Tq <= 64 and Tk <= 64: every training shape)
// The general kernel above is latency-bound there: one (batch, head) item per workgroup, load -> barrier -> compute ->
// store with two workgroups per CU and nothing in flight while the MFMAs run (124 us for 6144 items = 40 % of the HBM
// rate).  Here a workgroup is PERSISTENT over items and software-pipelined: the global loads of item i+1 (Q, dO, K, V,
// O: ten 16-byte chunks per thread, in registers) are issued before item i is computed and written to LDS after it,
// so the HBM latency hides under the MFMA / softmax-gradient work; with one key tile the dQ accumulator in LDS (16 KB)
// and its read-modify-write pass are gone -- dQ leaves from registers as transposed MFMA tiles (8-byte stores), like
// dK / dV; delta = rowsum(dO * O) comes from the staged registers (no second global read of dO).
// Same arithmetic as attn_bwd_kernel up to the summation order of delta and of the column sums (fp32, last bit).
struct BwdRegs { u32x4 q[2], d[2], k[2], v[2]; float lse; long long mk; };   // lse, mk: row / key `tid` of the tile, threads 0 .. 63 only   // lse, mk: see FwdRegs

// PACK (Tq, Tk <= 32, even H: the decoder's self-attention): ONE 64 x 64 tile carries TWO heads of a batch item -- rows 0 .. 31 head 2 hp,
// rows 32 .. 63 head 2 hp + 1 -- where a tile per head was half empty: half the items, the same work per item.
template <bool PACK>
__device__ __forceinline__ void bwd_load_item(const KmbAttn& p, int item, int tid, BwdRegs& x) {
  const int HH = PACK ? (p.H >> 1) : p.H;
  const int b = item / HH, h0 = PACK ? 2 * (item % HH) : item % HH;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;
    const int row = PACK ? ((id >> 3) & 31) : (id >> 3), c = id & 7;   // PACK: i = 1 is the second head's rows 0 .. 31
    const int h = PACK ? h0 + i : h0;
    const int tq = row < p.Tq ? row : p.Tq - 1, tk = row < p.Tk ? row : p.Tk - 1;
    // 32-bit BYTE offsets from the tensors' bases (the launcher sends shapes whose tensors reach 4 GB to the general kernel):
    // 64-bit multiplies are quarter-rate, and there were forty-five of them per item and wave here
    const uint32_t rq = (uint32_t)b * (uint32_t)p.Tq + (uint32_t)tq, rk = (uint32_t)b * (uint32_t)p.Tk + (uint32_t)tk;
    const uint32_t hc = (uint32_t)(h * HD + c * 8);
    auto at = [](const bf16_t* base, uint32_t elem) { return reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) + elem * 2u); };
    // rows 32 .. 63 (i = 1) of a tile that holds at most 32 rows are never fetched (the decoder's 32-token shapes: five of a self-attention
    // item's ten tile loads, three of a cross-attention item's): their registers keep the zeros they were initialised with -- finite, so that
    // the zero probabilities / score gradients of those rows multiply to zero -- and so do their LDS rows
    if (PACK || i == 0 || p.Tq > 32) {
      x.q[i] = *at(p.Q, rq * (uint32_t)p.ldq + hc);
      x.d[i] = *at(p.dO, rq * (uint32_t)p.lddo + hc);
    }
    if (PACK || i == 0 || p.Tk > 32) {
      x.k[i] = *at(p.K, rk * (uint32_t)p.ldk + hc);
      x.v[i] = *at(p.V, rk * (uint32_t)p.ldv + hc);
    }
  }
  // the tile's 64 log-sum-exps and 64 key-mask words: ONE load each, by the first wave only (row / key = tid; PACK: rows / keys 32 .. 63 are the
  // second head's).  Every thread used to load its row's word (2 loads) and its four keys' words (4 loads): sixteen loads a wave and item kept the
  // wave from issuing for 4 us -- in-kernel stamps, profiles/r05_attn_bwd_stamps.txt -- fourteen for 1.1
  if (tid < 64) {
    const int rl = PACK ? (tid & 31) : tid, hh = PACK ? h0 + (tid >> 5) : h0;
    const int tq = rl < p.Tq ? rl : p.Tq - 1, tk = rl < p.Tk ? rl : p.Tk - 1;
    x.lse = p.lse[((uint32_t)b * (uint32_t)p.H + (uint32_t)hh) * (uint32_t)p.Tq + (uint32_t)tq];
    if (p.key_mask != nullptr) x.mk = p.key_mask[(uint32_t)b * (uint32_t)p.Tk + (uint32_t)tk];
  }
}

template <bool PACK>
__global__ __launch_bounds__(256, 3) void attn_bwd_small_kernel(const KmbAttn p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* dOs = smem + TILE_BYTES;
  char* Ks = smem + 2 * TILE_BYTES;
  char* Vs = smem + 3 * TILE_BYTES;
  char* Ps = smem + 4 * TILE_BYTES;
  char* dSs = smem + 5 * TILE_BYTES;
  float* lse_s = reinterpret_cast<float*>(smem + 6 * TILE_BYTES);
  float* del_s = lse_s + 64;
  float* colq = del_s + 64;    // [4 waves][64] column sums of dQ / dK / dV over each wave's rows (bias-gradient partials)
  float* colk = colq + 256;
  float* colv = colk + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int HH = PACK ? (p.H >> 1) : p.H;   // items per batch element
  const int nitems = p.B * HH;
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
  auto row16_sum = [](float v) {   // inclusive prefix sums by row_shr 1, 2, 4, 8 (zeros shifted in): lane 15 = total
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));
    return v;
  };
  int item = blockIdx.x;
  if (item >= nitems) return;
  BwdRegs x;
#pragma unroll
  for (int i = 0; i < 2; ++i) x.q[i] = x.d[i] = x.k[i] = x.v[i] = u32x4{0u, 0u, 0u, 0u};
  x.lse = 0.f;
  x.mk = 1;
  bwd_load_item<PACK>(p, item, tid, x);
  [[maybe_unused]] int astamp_iter = -1;
  int prev_b = -1, prev_h = 0;   // the item whose column-sum partials (colq / colk / colv) are complete but not yet written
  auto write_colsums = [&]() {   // 64 threads: the four waves' partials of the previous item -> its row of the bias-gradient partials
    const size_t o = (size_t)prev_b * p.ld_colsum + prev_h * HD + tid;
    if (PACK) {   // waves 0, 1 hold the first head's rows, waves 2, 3 the second head's ((a + b) + (0 + 0) == a + b: the unpacked sums' bits)
      if (p.dq_colsum != nullptr) { p.dq_colsum[o] = colq[tid] + colq[64 + tid]; p.dq_colsum[o + HD] = colq[128 + tid] + colq[192 + tid]; }
      p.dk_colsum[o] = colk[tid] + colk[64 + tid]; p.dk_colsum[o + HD] = colk[128 + tid] + colk[192 + tid];
      p.dv_colsum[o] = colv[tid] + colv[64 + tid]; p.dv_colsum[o + HD] = colv[128 + tid] + colv[192 + tid];
      return;
    }
    if (p.dq_colsum != nullptr) p.dq_colsum[o] = (colq[tid] + colq[64 + tid]) + (colq[128 + tid] + colq[192 + tid]);
    p.dk_colsum[o] = (colk[tid] + colk[64 + tid]) + (colk[128 + tid] + colk[192 + tid]);
    p.dv_colsum[o] = (colv[tid] + colv[64 + tid]) + (colv[128 + tid] + colv[192 + tid]);
  };
  for (; item < nitems; item += gridDim.x) {
    const int b = item / HH, h = PACK ? 2 * (item % HH) : item % HH;   // PACK: h = the first of the tile's two heads
    ++astamp_iter;
    ASTAMP(0);
    __syncthreads();   // everyone is done with the previous item's LDS images (and its column-sum partials are complete)
    ASTAMP(1);
    // the previous item's column sums leave behind THIS barrier (round 5: a barrier of their own per item was 0.7 of its 16 us);
    // the partials are next written at the end of this iteration, two barriers further on
    if (p.dk_colsum != nullptr && prev_b >= 0 && tid < 64) write_colsums();
    // ---- staged registers -> LDS ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + 256 * i;
      const int row = id >> 3, c = id & 7;
      *reinterpret_cast<u32x4*>(Qs + tile_off(row, c)) = x.q[i];
      *reinterpret_cast<u32x4*>(dOs + tile_off(row, c)) = x.d[i];
      *reinterpret_cast<u32x4*>(Ks + tile_off(row, c)) = x.k[i];
      *reinterpret_cast<u32x4*>(Vs + tile_off(row, c)) = x.v[i];
    }
    if (tid < 64) {
      lse_s[tid] = x.lse;
      del_s[tid] = x.mk != 0 ? 1.f : 0.f;   // (the delta slots: free since delta comes from registers) key-mask flags
    }
    ASTAMP(2);
    __syncthreads();
    ASTAMP(3);
    bool key_on[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) key_on[j] = del_s[j * 16 + r] != 0.f;
    // ---- next item's loads go out now and land while this item is computed ----
    const int nxt = item + (int)gridDim.x;
    if (nxt < nitems) bwd_load_item<PACK>(p, nxt, tid, x);
    ASTAMP(4);
    // ---- S = Q K^T and dP = dO V^T for this wave's 16 query rows x 64 keys ----
    f32x4 s4[4], dp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s4[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 qf = frag_rows(Qs, wave, kk, r, g);
      const bf16x8 df = frag_rows(dOs, wave, kk, r, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, frag_rows(Ks, j, kk, r, g), s4[j], 0, 0, 0);
        dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, frag_rows(Vs, j, kk, r, g), dp[j], 0, 0, 0);
      }
    }
    // P and dS go to LDS TRANSPOSED ([key][query]): a lane holds one key (r) and four consecutive query rows (4 g ..), i.e. eight
    // contiguous bytes of the transposed image -- 8 stores of 8 bytes per lane where the [query][key] image took 32 of 2 bytes -- and
    // the readers below swap roles (the operand that read rows reads columns and vice versa: same fragments, same bits)
    const float lse4[4] = {lse_s[wave * 16 + g * 4], lse_s[wave * 16 + g * 4 + 1], lse_s[wave * 16 + g * 4 + 2], lse_s[wave * 16 + g * 4 + 3]};
    // delta[q] = sum over the keys of P[q][k] * dP[q][k] (= rowsum(dO * O), the softmax backward's correction term) from the probabilities
    // and dP this wave holds in registers -- the saved output O is not read at all (an eighth of an item's bytes, two of its sixteen loads)
    float del4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = j * 16 + r;
      const int keyl = PACK ? (key & 31) : key;                 // the key's position inside its head
      const bool kv = keyl < p.Tk && key_on[j];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int lrow = wave * 16 + g * 4 + q;
        const int rowl = PACK ? (lrow & 31) : lrow;
        const bool ok = kv && rowl < p.Tq && (!p.causal || keyl <= rowl) && (!PACK || (key >> 5) == (lrow >> 5));   // PACK: a query only sees its own head's keys
        const float pvq = (ok && lse4[q] != -INFINITY) ? __expf(s4[j][q] - lse4[q]) : 0.f;
        s4[j][q] = pvq;
        del4[q] += pvq * dp[j][q];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // the row's keys sit in the 16 lanes of a DPP row: four rotate-and-add steps leave the total in every lane
      float v = del4[q];
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
      del4[q] = v;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = j * 16 + r;
      float ds[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) ds[q] = s4[j][q] * (dp[j][q] - del4[q]);
      const int off = elem_off(key, wave * 16 + g * 4);
      *reinterpret_cast<uint2*>(Ps + off) = uint2{pack2bf(s4[j][0], s4[j][1]), pack2bf(s4[j][2], s4[j][3])};
      *reinterpret_cast<uint2*>(dSs + off) = uint2{pack2bf(ds[0], ds[1]), pack2bf(ds[2], ds[3])};
    }
    ASTAMP(5);
    __syncthreads();
    ASTAMP(6);
    // ---- dQ^T, dV^T, dK^T tiles: lane (r, g) holds row r of this wave's 16 and the columns 16 j + 4 g .. + 3 ----
    f32x4 dq[4], dk[4], dv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { dq[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 dsf = frag_cols(dSs, wave, kk, r, g);    // rows = this wave's query rows (image columns), k = keys (image rows)
      const bf16x8 ptf = frag_rows(Ps, wave, kk, r, g);     // rows = this wave's keys (image rows), k = query rows
      const bf16x8 dstf = frag_rows(dSs, wave, kk, r, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dq[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Ks, j, kk, r, g), dsf, dq[j], 0, 0, 0);
        dv[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(dOs, j, kk, r, g), ptf, dv[j], 0, 0, 0);
        dk[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Qs, j, kk, r, g), dstf, dk[j], 0, 0, 0);
      }
    }
    ASTAMP(7);
    const int row = wave * 16 + r;   // query row of dQ, key row of dK / dV
    const int rowl = PACK ? (row & 31) : row;   // the row inside its head
    const bool q_ok = rowl < p.Tq, k_ok = rowl < p.Tk;
    if (p.dk_colsum != nullptr) {   // uniform branch
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float sq = row16_sum(q_ok ? dq[j][q] * p.dq_scale : 0.f);
          const float sk = row16_sum(k_ok ? dk[j][q] : 0.f);
          const float sv = row16_sum(k_ok ? dv[j][q] : 0.f);
          if (r == 15) {
            colq[wave * 64 + j * 16 + g * 4 + q] = sq;
            colk[wave * 64 + j * 16 + g * 4 + q] = sk;
            colv[wave * 64 + j * 16 + g * 4 + q] = sv;
          }
        }
    }
    ASTAMP(8);
    // dQ / dK / dV leave through this wave's 2 KB of the V image (free since the barrier behind the S / dP phase, and 16 rows x 64 columns are
    // exactly one result tile of the wave): a lane holds four consecutive columns of a row per MFMA tile, i.e. 8-byte pieces of rows 4608 bytes
    // apart -- twelve store instructions per wave and item, sixteen 32-byte segments each; staged, the wave stores whole 128-byte rows, 16 bytes per
    // lane: six instructions.  (The first attempt at this -- through the Q / K / V images behind an extra workgroup barrier, at a time when sixteen
    // loads per wave were what the memory pipe waited for -- measured nothing; with eight loads per wave the stores are the longest phase left.)
    auto out_tile = [&](const f32x4 (&t)[4], float scale, bf16_t* base, uint32_t ld, int T) {
      const int srow = wave * 16 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<uint2*>(Vs + elem_off(srow, j * 16 + g * 4)) = uint2{pack2bf(t[j][0] * scale, t[j][1] * scale), pack2bf(t[j][2] * scale, t[j][3] * scale)};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the rows are this wave's own: no workgroup barrier
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int lrow = wave * 16 + (lane >> 3) + 8 * i, c = lane & 7;
        const int rl = PACK ? (lrow & 31) : lrow, hr = PACK ? h + (lrow >> 5) : h;
        const u32x4 v = *reinterpret_cast<const u32x4*>(Vs + tile_off(lrow, c));
        if (rl < T)
          *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(base) + (((uint32_t)b * (uint32_t)T + (uint32_t)rl) * ld + (uint32_t)(hr * HD + c * 8)) * 2u) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the next tile overwrites the rows just read)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    out_tile(dq, p.dq_scale, p.dQ, (uint32_t)p.lddq, p.Tq);
    out_tile(dk, 1.f, p.dK, (uint32_t)p.lddk, p.Tk);
    out_tile(dv, 1.f, p.dV, (uint32_t)p.lddv, p.Tk);
    ASTAMP(9);
    prev_b = b; prev_h = h;
  }
  if (p.dk_colsum != nullptr && prev_b >= 0) {   // the last item's column sums
    __syncthreads();
    if (tid < 64) write_colsums();
  }
}

// ------------------------------------------------- single-query decode step
// one wave per (row, head): scores over the cached keys, softmax, weighted sum of cached values
__global__ __launch_bounds__(256) void attn_decode_kernel(const KmbAttnDecode p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + wave;
  float* sc = reinterpret_cast<float*>(smem) + (size_t)wave * p.Tk;
  if (item >= p.R * p.H) return;  // whole wave exits together; no block barrier is used below
  const int row = item / p.H, h = item % p.H;
  const int crow = p.kv_row != nullptr ? p.kv_row[row] : row;
  const int mrow = p.mask_row != nullptr ? p.mask_row[row] : row;
  const int HDm = p.ldc;
  const bf16_t* Kc = p.Kc + (size_t)crow * p.Tmax * HDm + h * HD;
  const bf16_t* Vc = p.Vc + (size_t)crow * p.Tmax * HDm + h * HD;
  float q[HD];
  {
    const bf16_t* qrow = p.Q + (size_t)row * p.ldq + h * HD;
#pragma unroll
    for (int c = 0; c < 8; ++c) unpack8(*reinterpret_cast<const u32x4*>(qrow + c * 8), q + c * 8);
  }
  // decode self-attention: the newest key / value (position Tk - 1) comes from the projection output, not the cache;
  // this wave also appends it (one element per lane) -- the separate append launches are gone
  const int t_new = p.new_k != nullptr ? p.Tk - 1 : -1;
  const bf16_t* knew = p.new_k != nullptr ? p.new_k + (size_t)row * p.ld_new + h * HD : nullptr;
  const bf16_t* vnew = p.new_k != nullptr ? p.new_v + (size_t)row * p.ld_new + h * HD : nullptr;
  if (t_new >= 0) {
    p.Kw[((size_t)crow * p.Tmax + t_new) * HDm + h * HD + lane] = knew[lane];
    p.Vw[((size_t)crow * p.Tmax + t_new) * HDm + h * HD + lane] = vnew[lane];
    if (p.hist != nullptr && h == 0 && lane == 0) p.hist[(size_t)row * p.Tmax + t_new] = crow;
  }
  // history index (decode self-attention after beam reorders): position t of this row lives in cache row hist[row][t]
  const int32_t* hrow = p.hist != nullptr ? p.hist + (size_t)row * p.Tmax : nullptr;
  float mx = -INFINITY;
  for (int t = lane; t < p.Tk; t += 64) {
    float s = 0.f;
    const bf16_t* krow = (t == t_new) ? knew
                         : hrow != nullptr ? p.Kc + ((size_t)hrow[t] * p.Tmax + t) * HDm + h * HD : Kc + (size_t)t * HDm;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float k8[8];
      unpack8(*reinterpret_cast<const u32x4*>(krow + c * 8), k8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += q[c * 8 + e] * k8[e];
    }
    if (p.key_mask != nullptr && p.key_mask[(size_t)mrow * p.mask_ld + t] == 0) s = -INFINITY;
    sc[t] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float l = 0.f;
  for (int t = lane; t < p.Tk; t += 64) {
    const float e = (mx == -INFINITY) ? 0.f : __expf(sc[t] - mx);
    sc[t] = e;
    l += e;
  }
  l = wave_sum(l);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // four independent partial sums, eight value rows in flight: a single dependent chain over Tk rows made this
  // launch-sized kernel latency-bound (21 us for 64 keys)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int Tc = t_new >= 0 ? p.Tk - 1 : p.Tk;   // rows that live in the cache
  int t = 0;
  auto vrow = [&](int tt) -> const bf16_t* {
    return hrow != nullptr ? p.Vc + ((size_t)hrow[tt] * p.Tmax + tt) * HDm + h * HD : Vc + (size_t)tt * HDm;
  };
#pragma unroll 2
  for (; t + 4 <= Tc; t += 4) {
    a0 += sc[t] * bf2f(vrow(t)[lane]);
    a1 += sc[t + 1] * bf2f(vrow(t + 1)[lane]);
    a2 += sc[t + 2] * bf2f(vrow(t + 2)[lane]);
    a3 += sc[t + 3] * bf2f(vrow(t + 3)[lane]);
  }
  for (; t < Tc; ++t) a0 += sc[t] * bf2f(vrow(t)[lane]);
  if (t_new >= 0) a1 += sc[t_new] * bf2f(vnew[lane]);
  const float acc = (a0 + a1) + (a2 + a3);
  p.O[(size_t)row * p.ldo + h * HD + lane] = f2bf(acc * inv);
}

}  // namespace

const char* kmb_attn_check(const KmbAttn& p, int backward) {
  if (p.B <= 0 || p.H <= 0 || p.Tq <= 0 || p.Tk <= 0) return "attention: empty problem";
  if ((p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.ldo & 7)) return "attention: row strides must be multiples of 8";
  if (((uintptr_t)p.Q & 15) || ((uintptr_t)p.K & 15) || ((uintptr_t)p.V & 15) || ((uintptr_t)p.O & 15))
    return "attention: pointers must be 16-byte aligned";
  if (backward) {
    if (!p.dO || !p.dQ || !p.dK || !p.dV || !p.lse) return "attention backward: missing tensor";
    if ((p.lddo & 7) || (p.lddq & 7) || (p.lddk & 7) || (p.lddv & 7)) return "attention backward: row strides";
    if (((uintptr_t)p.dO & 15) || ((uintptr_t)p.dQ & 15)) return "attention backward: alignment";
    if (p.Tq > 384) return "attention backward: Tq > 384 is not supported (dQ accumulator lives in LDS)";
  }
  return nullptr;
}

hipError_t kmb_attn_fwd_launch(const KmbAttn& p, hipStream_t stream) {
  static const bool small_ok = !(KMB_DIAG_ENV("KMB_ATTN_FWD_SMALL") && KMB_DIAG_ENV("KMB_ATTN_FWD_SMALL")[0] == '0');
  // (the single-tile kernel's 32-bit byte offsets: every tensor below 4 GB)
  int ld_max = p.ldq;
  for (int l : {p.ldk, p.ldv, p.ldo}) ld_max = l > ld_max ? l : ld_max;
  const bool fits32 = (size_t)p.B * (size_t)(p.Tq > p.Tk ? p.Tq : p.Tk) * (size_t)ld_max * 2 < ((size_t)1 << 32);
  if (small_ok && p.Tq <= 64 && p.Tk <= 64 && p.B * p.H >= 1024 && fits32) {   // one query tile, one key tile, enough items to pipeline
    static const bool pack_ok = !(KMB_DIAG_ENV("KMB_ATTN_PACK") && KMB_DIAG_ENV("KMB_ATTN_PACK")[0] == '0');
    const bool pack = pack_ok && p.Tq <= 32 && p.Tk <= 32 && (p.H & 1) == 0;   // two heads per tile (32-token self-attention)
    const int items = pack ? p.B * (p.H >> 1) : p.B * p.H;
    if (pack) hipLaunchKernelGGL(attn_fwd_small_kernel<true>, dim3(items < 1024 ? items : 1024), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(attn_fwd_small_kernel<false>, dim3(items < 1024 ? items : 1024), dim3(256), 0, stream, p);   // four workgroups per CU
    return hipGetLastError();
  }
  dim3 grid(p.B * p.H, (p.Tq + 63) / 64), block(256);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, block, 0, stream, p);
  return hipGetLastError();
}

hipError_t kmb_attn_bwd_launch(const KmbAttn& p, hipStream_t stream) {
  const int nqt = (p.Tq + 63) / 64;
  static const bool small_ok = !(KMB_DIAG_ENV("KMB_ATTN_BWD_SMALL") && KMB_DIAG_ENV("KMB_ATTN_BWD_SMALL")[0] == '0');
  // (its 32-bit byte offsets: every tensor below 4 GB)
  const size_t rows_max = (size_t)p.B * (size_t)(p.Tq > p.Tk ? p.Tq : p.Tk);
  int ld_max = p.ldq;
  for (int l : {p.ldk, p.ldv, p.ldo, p.lddo, p.lddq, p.lddk, p.lddv}) ld_max = l > ld_max ? l : ld_max;
  const bool fits32 = rows_max * (size_t)ld_max * 2 < ((size_t)1 << 32);
  if (small_ok && p.Tq <= 64 && p.Tk <= 64 && fits32) {   // one query tile, one key tile: the persistent, software-pipelined form
    const size_t lds_s = 6 * TILE_BYTES + (128 + 768) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_small_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_small_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
      if (e != hipSuccess) return e;
      attr_set = true;
    }
    // two heads per tile for the 32-token self-attention shapes (KMB_ATTN_PACK=0 in the diagnostic build: one head per tile, as before round 5)
    static const bool pack_ok = !(KMB_DIAG_ENV("KMB_ATTN_PACK") && KMB_DIAG_ENV("KMB_ATTN_PACK")[0] == '0');
    const bool pack = pack_ok && p.Tq <= 32 && p.Tk <= 32 && (p.H & 1) == 0;
    const int items = pack ? p.B * (p.H >> 1) : p.B * p.H;
    const int grid = items < 768 ? items : 768;   // three workgroups per CU (165 VGPRs, 52.6 KB of LDS each)
    if (pack) hipLaunchKernelGGL(attn_bwd_small_kernel<true>, dim3(grid), dim3(256), lds_s, stream, p);
    else hipLaunchKernelGGL(attn_bwd_small_kernel<false>, dim3(grid), dim3(256), lds_s, stream, p);
    return hipGetLastError();
  }
  const size_t lds = 6 * TILE_BYTES + (128 + 512) * sizeof(float) + (size_t)nqt * 64 * 64 * sizeof(float);
  static size_t lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    lds_set = lds;
  }
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(p.B * p.H), dim3(256), lds, stream, p, nqt);
  return hipGetLastError();
}

hipError_t kmb_attn_decode_launch(const KmbAttnDecode& p, hipStream_t stream) {
  const int items = p.R * p.H;
  if (items <= 0) return hipSuccess;
  const size_t lds = (size_t)4 * p.Tk * sizeof(float);
  hipLaunchKernelGGL(attn_decode_kernel, dim3((items + 3) / 4), dim3(256), lds, stream, p);
  return hipGetLastError();
}
