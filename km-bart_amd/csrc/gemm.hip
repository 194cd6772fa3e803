// bf16 MFMA GEMM for gfx950 with fused epilogues.
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) )
//
// A is logical [M,K], B is logical [N,K] (a torch Linear weight is [out,in] = [N,K]).
// Each operand is either K-contiguous ("KC": X[m*ld + k]) or M/N-contiguous (X[k*ld + m]).
//   forward  Y  = X  W^T      : A=X  (KC)       B=W   (KC)
//   dgrad    dX = dY W        : A=dY (KC)       B=W   (N-contig, i.e. W[k'=n][n'=k])
//   wgrad    dW = dY^T X      : A=dY (M-contig) B=X   (N-contig), reduction over tokens
// Tile: 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16.
// LDS: two 32 KB stages (A tile + B tile), XOR-swizzled so the fragment reads are conflict-free:
//   KC tile      [128][64]  : 16-B chunk c of row r stored at chunk c ^ ((r>>1)&7)   (ds_read_b128)
//   non-KC tile  [64][128]  : 32-B chunk c of k-row r stored at chunk c ^ f(r),
//                             f(r) = (r&3) | ((r>>3)&1)<<2                           (ds_read_b64_tr_b16)
// The epilogue round-trips the fp32 accumulators through LDS so that bias / GeLU / dropout /
// residual math and the stores run row-major with 16-byte accesses.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <tuple>
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM * BK + BN * BK) * 2;  // 32 KB
constexpr int EPI_LD = BN + 4;                        // fp32 staging row stride
constexpr int CS256 = 16 * 128 * 4, CS512 = 32 * 128 * 4;  // column-sum scratch behind the fp32 staging (bias gradients)
constexpr int EPI_BYTES = BM * EPI_LD * 4;
constexpr int LDS_BYTES = ((EPI_BYTES > 2 * STAGE_BYTES) ? EPI_BYTES : 2 * STAGE_BYTES) + CS256;  // 74 KB: two per CU

// Workgroups are dealt round-robin over the 8 XCDs (each with a private L2): give every XCD a contiguous range
// of tile ids so that neighbouring tiles -- which share an A row panel -- hit the same L2 (bijective for any grid).
// Measured need: rocprof FETCH_SIZE showed the A panel fetched ~6x its size without the remap.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

__device__ __forceinline__ int swz_nkc(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// ---- global -> register staging of one 128x64 (KC) or 64x128 (non-KC) tile: 4 chunks / thread ----
template <bool KC>
__device__ __forceinline__ void load_tile(const bf16_t* __restrict__ X, int ld, int r0, int R, int k0, int K,
                                          int tid, u32x4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (KC) {
      int row = r0 + (id >> 3);
      const int k = k0 + (id & 7) * 8;
      row = row < R ? row : R - 1;  // rows past the edge are never stored: clamp to stay in bounds
      if (k < K) v = *reinterpret_cast<const u32x4*>(X + (size_t)row * ld + k);
    } else {
      const int k = k0 + (id >> 4);
      int m = r0 + (id & 15) * 8;
      const int mlast = ((R - 1) >> 3) << 3;
      m = m < R ? m : mlast;
      if (k < K) v = *reinterpret_cast<const u32x4*>(X + (size_t)k * ld + m);
    }
    regs[i] = v;
  }
}

template <bool KC>
__device__ __forceinline__ void store_tile(char* lds, int tid, const u32x4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i;
    int off;
    if (KC) {
      const int row = id >> 3, c = id & 7;
      off = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    } else {
      const int krow = id >> 4, piece = id & 15;
      off = krow * 256 + (((piece >> 1) ^ swz_nkc(krow)) << 5) + ((piece & 1) << 4);
    }
    *reinterpret_cast<u32x4*>(lds + off) = regs[i];
  }
}

// fragment for MFMA 16x16x32: 16 rows (r = lane&15) x 32 k (8 per lane group g = lane>>4)
template <bool KC>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int rowtile16, int kk, int r, int g) {
  if (KC) {
    const int row = rowtile16 * 16 + r;
    const int c = kk * 4 + g;
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
  } else {
    bf16x8 out;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int krow = kk * 32 + g * 8 + hh * 4 + (r >> 2);
      const int off = krow * 256 + ((rowtile16 ^ swz_nkc(krow)) << 5) + ((r & 3) << 3);
      const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(lds + off));
      out[hh * 4 + 0] = t[0]; out[hh * 4 + 1] = t[1]; out[hh * 4 + 2] = t[2]; out[hh * 4 + 3] = t[3];
    }
    return out;
  }
}

// NIT row-iterations over a staging image of row stride LD floats; SWZ: 16-column groups XOR-swapped by
// ((row >> 2) & 1) instead of padded rows; TILE_ROWS: rows of C covered by one call (column-sum contract)
template <int NT, bool HOIST = true, int NIT = 8, int LD = EPI_LD, bool SWZ = false, int TILE_ROWS = (NT == 512 ? 256 : 128)>
__device__ __forceinline__ void gemm_epilogue_phase2(const KmbGemm& p, const float* ef, float* cs, int tid, int row0,
                                                     int col0, int slice);

template <int NT>
__device__ __forceinline__ void gemm_epilogue(const KmbGemm& p, char* smem, f32x4 (&acc)[4][4], int tid, int wm, int wn,
                                              int r, int g, int row0, int col0, int slice) {
  // ---- epilogue phase 1: accumulators -> LDS fp32 [rows][EPI_LD] ----
  float* ef = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        ef[(wm * 64 + i * 16 + g * 4 + q) * EPI_LD + wn * 64 + j * 16 + r] = acc[i][j][q];
  __syncthreads();
  gemm_epilogue_phase2<NT>(p, ef, reinterpret_cast<float*>(smem + EPI_BYTES * (NT / 256)), tid, row0, col0, slice);
}

// cs: [NT/16][128] fp32 scratch (only touched when p.colsum != nullptr)
template <int NT, bool HOIST, int NIT, int LD, bool SWZ, int TILE_ROWS>
__device__ __forceinline__ void gemm_epilogue_phase2(const KmbGemm& p, const float* ef, float* cs, int tid, int row0,
                                                     int col0, int slice) {
  constexpr int RPP = NT / 16;  // rows per pass
  constexpr int NH = HOIST ? NIT : 1;
  auto eoff = [&](int lrow, int c) { return lrow * LD + (SWZ ? (c ^ (((lrow >> 2) & 1) << 4)) : c); };
  // ---- phase 2: row-major math + 16-byte stores ----
  const int c8 = (tid & 15) * 8;
  const int gcol = col0 + c8;
  if (gcol >= p.N && p.colsum == nullptr) return;
  const int nvalid = gcol >= p.N ? 0 : ((p.N - gcol) < 8 ? (p.N - gcol) : 8);
  float csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
  if (p.split_k > 1) {  // raw partial sums of this K slice -> slab[slice][M][N]
    float* slab = p.slab + (size_t)slice * p.M * p.N;
    for (int it = 0; it < NIT; ++it) {
      const int lrow = (tid >> 4) + RPP * it;
      const int grow = row0 + lrow;
      if (grow >= p.M) break;
      float* o = slab + (size_t)grow * p.N + gcol;
      if (nvalid == 8 && (p.N & 3) == 0) {
        *reinterpret_cast<f32x4*>(o) = *reinterpret_cast<const f32x4*>(ef + eoff(lrow, c8));
        *reinterpret_cast<f32x4*>(o + 4) = *reinterpret_cast<const f32x4*>(ef + eoff(lrow, c8) + 4);
      } else {
        for (int e = 0; e < nvalid; ++e) o[e] = ef[eoff(lrow, c8) + e];
      }
    }
    return;
  }
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = (p.bias != nullptr && e < nvalid) ? p.bias[gcol + e] : 0.f;

  // ---- gather everything the 8 row-iterations need first (LDS reads and global residual / aux loads are then in
  //      flight together instead of one dependent round trip per iteration) ----
  const bool full8 = nvalid == 8;
  f32x4 vlo[NH], vhi[NH];
  u32x4 resv[NH], auxv[NH];
  auto gather = [&](int it, int slot) {
    const int lrow = (tid >> 4) + RPP * it;
    const int grow = row0 + lrow;
    const bool ok = grow < p.M && nvalid > 0;
    vlo[slot] = *reinterpret_cast<const f32x4*>(ef + eoff(lrow, c8));
    vhi[slot] = *reinterpret_cast<const f32x4*>(ef + eoff(lrow, c8) + 4);
    resv[slot] = u32x4{0u, 0u, 0u, 0u};
    auxv[slot] = u32x4{0u, 0u, 0u, 0u};
    if (ok && full8) {
      if (p.residual != nullptr) resv[slot] = *reinterpret_cast<const u32x4*>(p.residual + (size_t)grow * p.ld_res + gcol);
      if (p.act == 2 || p.act == 4) auxv[slot] = *reinterpret_cast<const u32x4*>(p.aux + (size_t)grow * p.ld_aux + gcol);
    }
  };
  if (HOIST) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) gather(it, it);
  }
  auto process = [&](int it0, int it) -> bool {
    const int lrow = (tid >> 4) + RPP * it0;
    const int grow = row0 + lrow;
    if (grow >= p.M || nvalid == 0) return false;
    float v[8];
    v[0] = vlo[it][0]; v[1] = vlo[it][1]; v[2] = vlo[it][2]; v[3] = vlo[it][3];
    v[4] = vhi[it][0]; v[5] = vhi[it][1]; v[6] = vhi[it][2]; v[7] = vhi[it][3];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] += bias8[e];
      if (gcol + e < p.col_scale_n) v[e] *= p.col_scale;
    }
    if (p.act == 1) {
      if (p.preact != nullptr) {
        if (full8) {
          *reinterpret_cast<u32x4*>(p.preact + (size_t)grow * p.ld_preact + gcol) = pack8(v);
        } else {
          for (int e = 0; e < nvalid; ++e) p.preact[(size_t)grow * p.ld_preact + gcol + e] = f2bf(v[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
    } else if (p.act == 2 || p.act == 4) {
      float u[8];
      if (full8) {
        unpack8(auxv[it], u);
      } else {
        for (int e = 0; e < 8; ++e) u[e] = e < nvalid ? bf2f(p.aux[(size_t)grow * p.ld_aux + gcol + e]) : 0.f;
      }
      if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_f(u[e]);
      } else {  // tanh'(.) = 1 - y^2, aux = y
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= (1.f - u[e] * u[e]);
      }
    } else if (p.act == 3) {  // BartClassificationHead: tanh
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
    }
    if (p.drop_thr16 != 0u) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        v[e] = drop_keep(p.drop_seed, (uint32_t)grow, (uint32_t)(gcol + e), p.drop_thr16) ? v[e] * p.drop_scale : 0.f;
    }
    if (p.residual != nullptr) {
      float rr[8];
      if (full8) {
        unpack8(resv[it], rr);
      } else {
        for (int e = 0; e < 8; ++e) rr[e] = e < nvalid ? bf2f(p.residual[(size_t)grow * p.ld_res + gcol + e]) : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rr[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] += v[e];
    if (p.tile_order & 256) {  // ablation build path (tools/gemm_ablate.py): keep the math alive, drop the stores
      float keep = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) keep += v[e];
      if (keep == 1.2345e30f) p.out_bf16[0] = 0;
      return true;
    }
    if (p.out_bf16 != nullptr) {
      if (full8) {
        *reinterpret_cast<u32x4*>(p.out_bf16 + (size_t)grow * p.ld_out_bf16 + gcol) = pack8(v);
      } else {
        for (int e = 0; e < nvalid; ++e) p.out_bf16[(size_t)grow * p.ld_out_bf16 + gcol + e] = f2bf(v[e]);
      }
    }
    if (p.out_f32 != nullptr) {
      float* o = p.out_f32 + (size_t)grow * p.ld_out_f32 + gcol;
      const bool vec = full8 && ((p.ld_out_f32 & 3) == 0);
      if (p.beta != 0.f) {
        for (int e = 0; e < nvalid; ++e) v[e] += p.beta * o[e];
      }
      if (vec) {
        *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
        for (int e = 0; e < nvalid; ++e) o[e] = v[e];
      }
    }
    return true;
  };
  if (HOIST) {
#pragma unroll
    for (int it0 = 0; it0 < NIT; ++it0)
      if (!process(it0, it0)) break;
  } else {
#pragma unroll 2
    for (int it0 = 0; it0 < NIT; ++it0) {
      gather(it0, 0);
      if (!process(it0, 0)) break;
    }
  }
  if (p.colsum != nullptr) {
    // column sums of this tile's stored values: reduce the row-lanes through LDS, one partial row per 128 rows
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[(tid >> 4) * 128 + c8 + e] = csum[e];
    __syncthreads();
    if (tid < 128 && col0 + tid < p.N) {
      float t = 0.f;
#pragma unroll 8
      for (int k = 0; k < RPP; ++k) t += cs[k * 128 + tid];
      // contract: one partial row per 64 rows of C; this tile fills its first row and zeroes the rest
      const int prow = row0 >> 6;
      p.colsum[(size_t)prow * p.N + col0 + tid] = t;
      for (int k = 1; k < TILE_ROWS / 64; ++k)
        if (row0 + 64 * k < p.M) p.colsum[(size_t)(prow + k) * p.N + col0 + tid] = 0.f;
    }
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const KmbGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int nsl = p.split_k > 1 ? p.split_k : 1;
  const int bid = (p.tile_order & 1) ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int tile = bid / nsl, slice = bid % nsl;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int row0 = tm * BM, col0 = tn * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt_all = (p.K + BK - 1) / BK;
  const int t_begin = (int)((long)nt_all * slice / nsl), t_end = (int)((long)nt_all * (slice + 1) / nsl);
  const int nt = t_end - t_begin;
  u32x4 ra[4], rb[4];
  load_tile<A_KC>(p.A, p.lda, row0, p.M, t_begin * BK, p.K, tid, ra);
  load_tile<B_KC>(p.B, p.ldb, col0, p.N, t_begin * BK, p.K, tid, rb);
  store_tile<A_KC>(smem, tid, ra);
  store_tile<B_KC>(smem + BM * BK * 2, tid, rb);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * STAGE_BYTES;
    char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
    const bool more = (t + 1) < nt;
    if (more) {
      load_tile<A_KC>(p.A, p.lda, row0, p.M, (t_begin + t + 1) * BK, p.K, tid, ra);
      load_tile<B_KC>(p.B, p.ldb, col0, p.N, (t_begin + t + 1) * BK, p.K, tid, rb);
    }
    const char* la = cur;
    const char* lb = cur + BM * BK * 2;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag<A_KC>(la, wm * 4 + i, kk, r, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag<B_KC>(lb, wn * 4 + j, kk, r, g);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      store_tile<A_KC>(nxt, tid, ra);
      store_tile<B_KC>(nxt + BM * BK * 2, tid, rb);
    }
    __syncthreads();
  }

  gemm_epilogue<256>(p, smem, acc, tid, wm, wn, r, g, row0, col0, slice);
}

// ------------------------------------------------------------------------------------------
// v2: same tile / fragment maps, but the global->LDS staging is LDS-DMA (global_load_lds, 16 B per
// lane): no staging VGPRs and no ds_write pass.  A wave instruction writes 1 KiB of LDS linearly
// (base + lane*16), so the XOR swizzle is applied to the per-lane SOURCE address instead: lane i of
// piece p lands on physical chunk (i % 8 or i % 16) of its row and therefore fetches the LOGICAL
// chunk that the swizzle maps there.  Requires K % 64 == 0 (no zero-fill on this path).
template <bool KC>
__device__ __forceinline__ void glds_tile(char* lds_tile, const bf16_t* __restrict__ X, int ld, int r0, int R, int k0,
                                          int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;  // 16 pieces of 1 KiB per 16 KiB tile
    const bf16_t* src;
    if (KC) {
      const int row = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      int grow = r0 + row;
      grow = grow < R ? grow : R - 1;
      src = X + (size_t)grow * ld + k0 + c * 8;
    } else {
      const int krow = piece * 4 + (lane >> 4);
      const int ps = lane & 15;
      const int c32 = (ps >> 1) ^ swz_nkc(krow);
      int m = r0 + c32 * 16 + (ps & 1) * 8;
      const int mlast = ((R - 1) >> 3) << 3;
      m = m < R ? m : mlast;
      src = X + (size_t)(k0 + krow) * ld + m;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024), 16, 0, 0);
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_kernel_v2(const KmbGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nsl = p.split_k > 1 ? p.split_k : 1;
  const int bid = (p.tile_order & 1) ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int tile = bid / nsl, slice = bid % nsl;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int row0 = tm * BM, col0 = tn * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt_all = p.K / BK;
  const int t_begin = (int)((long)nt_all * slice / nsl), t_end = (int)((long)nt_all * (slice + 1) / nsl);
  const int nt = t_end - t_begin;
  glds_tile<A_KC>(smem, p.A, p.lda, row0, p.M, t_begin * BK, wave, lane);
  glds_tile<B_KC>(smem + BM * BK * 2, p.B, p.ldb, col0, p.N, t_begin * BK, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int nt_run = (p.tile_order & 512) ? 1 : nt;  // ablation: one K step only
  for (int t = 0; t < nt_run; ++t) {
    char* cur = smem + (t & 1) * STAGE_BYTES;
    char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
    if (t + 1 < nt) {
      glds_tile<A_KC>(nxt, p.A, p.lda, row0, p.M, (t_begin + t + 1) * BK, wave, lane);
      glds_tile<B_KC>(nxt + BM * BK * 2, p.B, p.ldb, col0, p.N, (t_begin + t + 1) * BK, wave, lane);
    }
    const char* la = cur;
    const char* lb = cur + BM * BK * 2;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag<A_KC>(la, wm * 4 + i, kk, r, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag<B_KC>(lb, wn * 4 + j, kk, r, g);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  gemm_epilogue<256>(p, smem, acc, tid, wm, wn, r, g, row0, col0, slice);
}


// ------------------------------------------------------------------------------------------
// LDS-DMA staging / fragment reads for tiles that are 128 or 256 rows (columns) tall; used by v4.
// (A 256x128 three-stage variant with counted vmcnt was measured 10-25 % slower than v2 on every training
//  shape -- one workgroup per CU loses the cross-workgroup overlap -- and was removed.)

template <bool KC, int ROWS>
__device__ __forceinline__ void glds_tile3(char* lds_tile, const bf16_t* __restrict__ X, int ld, int r0, int R, int k0,
                                           int wave, int lane) {
  constexpr int PIECES = ROWS / 8;   // 1 KiB pieces per tile
  constexpr int PER_WAVE = PIECES / 8;
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int piece = wave * PER_WAVE + i;
    const bf16_t* src;
    if (KC) {
      const int row = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      int grow = r0 + row;
      grow = grow < R ? grow : R - 1;
      src = X + (size_t)grow * ld + k0 + c * 8;
    } else {
      constexpr int LPR = ROWS / 8;            // lanes (16-byte slots) per k-row
      const int krow = piece * (64 / LPR) + lane / LPR;
      const int ps = lane % LPR;
      const int c32 = (ps >> 1) ^ swz_nkc(krow);
      int m = r0 + c32 * 16 + (ps & 1) * 8;
      const int mlast = ((R - 1) >> 3) << 3;
      m = m < R ? m : mlast;
      src = X + (size_t)(k0 + krow) * ld + m;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024), 16, 0, 0);
  }
}

template <bool KC, int ROWS>
__device__ __forceinline__ bf16x8 read_frag3(const char* lds, int rowtile16, int kk, int r, int g) {
  if (KC) {
    const int row = rowtile16 * 16 + r;
    const int c = kk * 4 + g;
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
  } else {
    bf16x8 out;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int krow = kk * 32 + g * 8 + hh * 4 + (r >> 2);
      const int off = krow * (ROWS * 2) + ((rowtile16 ^ swz_nkc(krow)) << 5) + ((r & 3) << 3);
      const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(lds + off));
      out[hh * 4 + 0] = t[0]; out[hh * 4 + 1] = t[1]; out[hh * 4 + 2] = t[2]; out[hh * 4 + 3] = t[3];
    }
    return out;
  }
}

// ------------------------------------------------------------------------------------------
// v4: 256x256x64 tile, 512 threads = 8 waves (2x4), each wave 128x64 (8x4 MFMA tiles): 12 fragment reads per
// 32 MFMAs instead of 8 per 16 -- the LDS read traffic per MFMA is what bounds the 64x64-per-wave kernels.
// Two 64 KB stages filled by LDS-DMA; epilogue in two column halves (fp32 staging does not fit otherwise).
constexpr int BM4 = 256, BN4 = 256;
constexpr int ST4 = (BM4 + BN4) * BK * 2;   // 64 KB
constexpr int LDS4 = ((2 * ST4 > 2 * EPI_BYTES) ? 2 * ST4 : 2 * EPI_BYTES) + CS512;

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v4(const KmbGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, g = lane >> 4;
  const int tiles_n = (p.N + BN4 - 1) / BN4;
  const int nsl = p.split_k > 1 ? p.split_k : 1;
  const int bid = (p.tile_order & 1) ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int tile = bid / nsl, slice = bid % nsl;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int row0 = tm * BM4, col0 = tn * BN4;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt_all = p.K / BK;
  const int t_begin = (int)((long)nt_all * slice / nsl), t_end = (int)((long)nt_all * (slice + 1) / nsl);
  const int nt = t_end - t_begin;
  constexpr int A_BYTES = BM4 * BK * 2;

  glds_tile3<A_KC, BM4>(smem, p.A, p.lda, row0, p.M, t_begin * BK, wave, lane);
  glds_tile3<B_KC, BN4>(smem + A_BYTES, p.B, p.ldb, col0, p.N, t_begin * BK, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * ST4;
    char* nxt = smem + ((t + 1) & 1) * ST4;
    if (t + 1 < nt) {
      glds_tile3<A_KC, BM4>(nxt, p.A, p.lda, row0, p.M, (t_begin + t + 1) * BK, wave, lane);
      glds_tile3<B_KC, BN4>(nxt + A_BYTES, p.B, p.ldb, col0, p.N, (t_begin + t + 1) * BK, wave, lane);
    }
    const char* la = cur;
    const char* lb = cur + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[8], fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag3<B_KC, BN4>(lb, wn * 4 + j, kk, r, g);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = read_frag3<A_KC, BM4>(la, wm * 8 + i, kk, r, g);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // epilogue: two passes over the column halves; in pass h the waves with (wn >> 1) == h stage their accumulators
  float* ef = reinterpret_cast<float*>(smem);
  for (int h = 0; h < 2; ++h) {
    if ((wn >> 1) == h) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            ef[(wm * 128 + i * 16 + g * 4 + q) * EPI_LD + (wn & 1) * 64 + j * 16 + r] = acc[i][j][q];
    }
    __syncthreads();
    gemm_epilogue_phase2<512, false>(p, ef, reinterpret_cast<float*>(smem + 2 * EPI_BYTES), tid, row0, col0 + h * 128, slice);
    __syncthreads();
  }
}


// Variants that were built, verified bit-identical and then REMOVED because they measured slower on every training
// shape (MI355X, b=256): a 256x128 three-stage ring with counted vmcnt (one workgroup per CU: -10..25 %), a
// transposed-block MFMA with a register epilogue and 8-byte stores (-20 %: the stores are issue-bound), and several
// tiles per workgroup with the next tile's first K step prefetched under the epilogue (-5..30 %: fewer independent
// workgroups to overlap).  What did pay: LDS-DMA staging, hoisted epilogue loads, hardware bf16 conversion, split-K
// for the weight gradients, per-shape choice between the 128x128 and 256x256 tiles and the XCD tile order.

}  // namespace

const char* kmb_gemm_check(const KmbGemm& p) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return "gemm: empty problem";
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) return "gemm: operand not 16-byte aligned";
  if ((p.lda & 7) || (p.ldb & 7)) return "gemm: leading dimension must be a multiple of 8";
  if (p.a_kc && (p.K & 7)) return "gemm: K must be a multiple of 8 for a K-contiguous A";
  if (p.b_kc && (p.K & 7)) return "gemm: K must be a multiple of 8 for a K-contiguous B";
  if (!p.a_kc && (((p.M + 7) & ~7) > p.lda)) return "gemm: M-contiguous A needs lda >= roundup8(M)";
  if (!p.b_kc && (((p.N + 7) & ~7) > p.ldb)) return "gemm: N-contiguous B needs ldb >= roundup8(N)";
  if (p.out_bf16 && ((p.ld_out_bf16 & 7) || ((uintptr_t)p.out_bf16 & 15))) return "gemm: bf16 output alignment";
  if (p.residual && ((p.ld_res & 7) || ((uintptr_t)p.residual & 15))) return "gemm: residual alignment";
  if (p.aux && ((p.ld_aux & 7) || ((uintptr_t)p.aux & 15))) return "gemm: aux alignment";
  if (p.preact && ((p.ld_preact & 7) || ((uintptr_t)p.preact & 15))) return "gemm: preact alignment";
  if (p.out_f32 && ((uintptr_t)p.out_f32 & 15)) return "gemm: f32 output alignment";
  if ((p.act == 2 || p.act == 4) && !p.aux) return "gemm: derivative epilogue needs aux";
  if (!p.a_kc && p.b_kc) return "gemm: (M-contiguous A, K-contiguous B) is not instantiated";
  if (p.colsum && p.split_k > 1) return "gemm: column sums are not available with split-K";
  if (p.split_k > 1) {
    if (!p.slab || ((uintptr_t)p.slab & 15)) return "gemm: split-K needs a 16-byte aligned slab";
    if (p.split_k > (p.K + BK - 1) / BK) return "gemm: more K slices than K steps";
  }
  return nullptr;
}

namespace {

hipError_t launch_variant(int variant, const KmbGemm& p, hipStream_t stream) {
  const int nsl = p.split_k > 1 ? p.split_k : 1;
  if (variant == 4) {
    const int tiles = ((p.M + BM4 - 1) / BM4) * ((p.N + BN4 - 1) / BN4);
    dim3 grid(tiles * nsl), block(512);
    if (p.a_kc && p.b_kc) hipLaunchKernelGGL((gemm_kernel_v4<true, true>), grid, block, LDS4, stream, p);
    else if (p.a_kc) hipLaunchKernelGGL((gemm_kernel_v4<true, false>), grid, block, LDS4, stream, p);
    else hipLaunchKernelGGL((gemm_kernel_v4<false, false>), grid, block, LDS4, stream, p);
  } else {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    dim3 grid(tiles * nsl), block(256);
    if (variant == 2) {
      if (p.a_kc && p.b_kc) hipLaunchKernelGGL((gemm_kernel_v2<true, true>), grid, block, LDS_BYTES, stream, p);
      else if (p.a_kc) hipLaunchKernelGGL((gemm_kernel_v2<true, false>), grid, block, LDS_BYTES, stream, p);
      else hipLaunchKernelGGL((gemm_kernel_v2<false, false>), grid, block, LDS_BYTES, stream, p);
    } else {
      if (p.a_kc && p.b_kc) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, LDS_BYTES, stream, p);
      else if (p.a_kc) hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, LDS_BYTES, stream, p);
      else hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, LDS_BYTES, stream, p);
    }
  }
  return hipGetLastError();
}

struct TuneKey {
  int akc, bkc, M, N, K, split, act;
  bool operator<(const TuneKey& o) const {
    return std::tie(akc, bkc, M, N, K, split, act) < std::tie(o.akc, o.bkc, o.M, o.N, o.K, o.split, o.act);
  }
};
std::map<TuneKey, int> g_best;

bool writes_an_input(const KmbGemm& p) {
  const void* outs[3] = {p.out_bf16, p.out_f32, p.preact};
  const void* ins[4] = {p.A, p.B, p.residual, p.aux};
  for (const void* o : outs)
    if (o)
      for (const void* i : ins)
        if (i == o) return true;
  return p.beta != 0.f;
}

}  // namespace

// Every variant computes bit-identical results (same per-element accumulation order), so the choice is pure
// speed: the first launch of a new shape times the eligible variants on the real operands (measure, don't guess).
hipError_t kmb_gemm_launch(const KmbGemm& p, hipStream_t stream) {
  static int forced = -1, autotune = 1, verbose = 0;
  if (forced < 0) {
    const char* ev = getenv("KMB_GEMM_VARIANT");
    forced = ev ? atoi(ev) : 0;
    const char* ea = getenv("KMB_GEMM_AUTOTUNE");
    if (ea && ea[0] == '0') autotune = 0;
    verbose = getenv("KMB_GEMM_VERBOSE") != nullptr;
    (void)hipFuncSetAttribute((const void*)gemm_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v2<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v2<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v2<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v4<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v4<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v4<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
  }
  const bool dma_ok = (p.K % BK) == 0;           // LDS-DMA variants have no K-edge zero fill
  const bool big = dma_ok && p.M > 128;
  if (!dma_ok) return launch_variant(1, p, stream);
  if (forced) {
    int v = forced;
    if (v == 3 || (v == 4 && !(big && p.N > 128))) v = 2;
    if (v != 1 && v != 2 && v != 4) v = 2;
    return launch_variant(v, p, stream);
  }
  if (!big || p.N <= 128) return launch_variant(2, p, stream);
  const TuneKey key{p.a_kc, p.b_kc, p.M, p.N, p.K, p.split_k, p.act};
  auto it = g_best.find(key);
  if (it == g_best.end()) {
    if (!autotune || writes_an_input(p)) return launch_variant(2, p, stream);
    const int cands[4] = {2, 2 + 16, 4, 4 + 16};   // variant | (tile_order << 4)
    float best_ms = 1e30f;
    int best = 2;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return launch_variant(2, p, stream);
    for (int c : cands) {
      KmbGemm q = p;
      q.tile_order = c >> 4;
      hipError_t e = launch_variant(c & 15, q, stream);  // warm
      if (e != hipSuccess) return e;
      (void)hipEventRecord(e0, stream);
      for (int rep = 0; rep < 3; ++rep) (void)launch_variant(c & 15, q, stream);
      (void)hipEventRecord(e1, stream);
      if (hipEventSynchronize(e1) != hipSuccess) return hipGetLastError();
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (verbose) fprintf(stderr, "[kmb gemm tune] akc=%d bkc=%d M=%d N=%d K=%d split=%d act=%d v%d order%d %.1f us\n",
                           p.a_kc, p.b_kc, p.M, p.N, p.K, p.split_k, p.act, c & 15, c >> 4, ms / 3 * 1e3);
      if (ms < best_ms) { best_ms = ms; best = c; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    it = g_best.emplace(key, best).first;
  }
  KmbGemm q = p;
  q.tile_order = it->second >> 4;
  return launch_variant(it->second & 15, q, stream);
}
