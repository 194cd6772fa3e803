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
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>
#include <type_traits>
#include "common.h"
#include "diag.h"
#include "kernels.h"

// ---- diagnostic build only (tools/gemm_stamps.py compiles this file with -DKMB_GEMM_STAMP into a separate library):
// per-workgroup s_memrealtime stamps (100 MHz) + HW_ID / XCC_ID, to read prologue / loop / epilogue / dispatch-gap
// times off the real kernel.  The product library has none of this.
#ifdef KMB_GEMM_STAMP
#ifndef KMB_STAMP_SLOTS
#define KMB_STAMP_SLOTS 8   // tools/gemm_clock.py builds with 12: slots 8 / 9 = the persistent kernel's s_memtime / s_memrealtime spans
#endif
__device__ unsigned long long* g_kmb_stamps = nullptr;  // [grid][KMB_STAMP_SLOTS]
extern "C" int kmb_debug_set_stamps(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_kmb_stamps), &p, sizeof(p));
}
#define KMB_STAMP(i)                                                                                   \
  do {                                                                                                 \
    if (g_kmb_stamps != nullptr && threadIdx.x == 0)                                                   \
      g_kmb_stamps[(size_t)blockIdx.x * KMB_STAMP_SLOTS + (i)] = __builtin_amdgcn_s_memrealtime();                   \
  } while (0)
#define KMB_STAMP_ID()                                                                                 \
  do {                                                                                                 \
    if (g_kmb_stamps != nullptr && threadIdx.x == 0)                                                   \
      g_kmb_stamps[(size_t)blockIdx.x * KMB_STAMP_SLOTS + 7] =                                                       \
          (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) |                                      \
          ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32);                               \
  } while (0)
#define KMB_WAIT_BEGIN() const uint64_t kmb_w0 = __builtin_amdgcn_s_memrealtime()
#define KMB_NOW() __builtin_amdgcn_s_memrealtime()
#define KMB_WAIT_END(acc) (acc) += __builtin_amdgcn_s_memrealtime() - kmb_w0
#define KMB_STAMP_VALUE(i, v)                                                                          \
  do {                                                                                                 \
    if (g_kmb_stamps != nullptr && threadIdx.x == 0) g_kmb_stamps[(size_t)blockIdx.x * KMB_STAMP_SLOTS + (i)] = (v); \
  } while (0)
#else
#define KMB_STAMP(i)
#define KMB_STAMP_ID()
#define KMB_WAIT_BEGIN()
#define KMB_NOW() uint64_t(0)
#define KMB_WAIT_END(acc)
#define KMB_STAMP_VALUE(i, v)
#endif

// KMB_PLAIN_STORES (experiment builds, build.py --variant): default-policy stores in the persistent kernels' epilogues
// instead of non-temporal ones
#ifdef KMB_PLAIN_STORES
#define KMB_NT_STORE(v, p) (*(p) = (v))
#else
#define KMB_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif

// The L2 touch of the persistent kernels: a load whose result nobody reads.  Round 5: it is a 4-byte LDS-DMA into a dummy LDS word
// of the issuing wave (`lds`: 256 bytes that nothing reads while a touch can be in flight -- the wave's epilogue staging image,
// idle during the K loop) -- NO register destination.  Rounds 2-4 gave it a register: "=v" (a fresh value per touch: the allocator
// reused the register while the load was in flight -- wrong bits), one "+v" web (split under pressure: memory fault), then v255 with
// __attribute__((amdgpu_num_vgpr(255))), on the belief that the allocator then never hands out v255.  It does (found by grepping the ISA of every kernel with a touch for other uses of v255,
// round 5: the eight-wave and the two-workgroup kernels are compiled with all 256 registers, v255 among them -- the attribute does not
// cap a kernel whose budget waves_per_eu fixes); what kept the results right was the ORDER of the counted waits (a touch is older
// than the pieces the next wait leaves outstanding), not the register.  A touch still counts as one vector-memory operation, so the
// kernels' counted waits are unchanged; results are bit-identical (tests/test_gemm_variants_gpu.py).
// (inline asm, not __builtin_amdgcn_global_load_lds: the builtin spends eight scalar instructions per touch on turning the generic
//  LDS pointer into M0 -- measured -0.5...-1 % of a step; here M0 is saved, set from a 32-bit LDS address kept in a scalar
//  register and restored inside ONE statement, as the guide's glds16_asm recipe does.  `lds_u32`: kmb_lds_addr(ptr), wave-uniform.)
#define KMB_L2_TOUCH(voff, sbase, lds_u32)                                                                                    \
  do {                                                                                                                        \
    unsigned kmb_m0_keep_;                                                                                                    \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"       \
                 : "=&s"(kmb_m0_keep_) : "v"(voff), "s"(sbase), "s"(lds_u32) : "memory");                                     \
  } while (0)
__device__ __forceinline__ unsigned kmb_lds_addr(const void* p) {   // the 32-bit LDS address of a (generic) pointer into shared memory, in a scalar register
  return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) const void*)p);
}
namespace {

int g_shared_device = 0;   // kmb_gemm_shared_device(): other kernels (RCCL) hold CUs while the GEMMs run

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM * BK + BN * BK) * 2;  // 32 KB
constexpr int EPI_LD = BN + 4;                        // fp32 staging row stride
constexpr int CS256 = 16 * 128 * 4, CS512 = 32 * 128 * 4;  // column-sum scratch behind the fp32 staging (bias gradients)
constexpr int EPI_BYTES = BM * EPI_LD * 4;
constexpr int LDS_BYTES = ((EPI_BYTES > 2 * STAGE_BYTES) ? EPI_BYTES : 2 * STAGE_BYTES) + CS256;  // 74 KB: two per CU

// Workgroups are dealt round-robin over the 8 XCDs (each with a private L2): give every XCD a contiguous range
// of tile ids so that neighbouring tiles -- which share an A row panel -- hit the same L2 (bijective for any grid).
// Measured need: rocprof FETCH_SIZE showed the A panel fetched ~6x its size without the remap.
// split_order_note -- split-K launches (weight gradients: few output tiles, all tokens to reduce over): block ids
// enumerate (tile, slice) pairs slice-minor, so with or without the remap every XCD works on every K slice and its L2
// pulls in both operands over their whole length: a 3072x768x32768 weight gradient fetched 660 MB for 251 MB of
// operands (rocprofv3 FETCH_SIZE; the 768-wide operand 8x, once per XCD).  tile_order bit 2 enumerates them slice-MAJOR
// (all tiles of slice 0, then slice 1, ...); together with the remap's contiguous ranges an XCD then works on one or
// two K slices and reads only those token ranges of the operands.  Results do not depend on the enumeration.
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
// The transposing LDS reads (`ds_read_b64_tr_b16`, the token-major operand of the data- and weight-gradient layouts) as INLINE ASM (late round 5).
// hipcc puts `s_waitcnt vmcnt(0)` in front of the first __builtin_amdgcn_ds_read_tr16_b64 behind an LDS-DMA issue: the intrinsic carries no memory
// operand, so the wait-count pass assumes it may read what the DMA is writing.  The K loops issue the NEXT stage's DMA pieces and then read fragments
// of the CURRENT one -- so in every kernel with a token-major operand the stage requested a moment ago was waited for at once: prefetch distance
// zero, two such stalls per K step in the weight-gradient kernel (found on the ISA: one counted wait per step in the K-contiguous kernels, two or
// three vmcnt(0) in the others; their matrix pipes were busy 29-40 % against 42 %).  As asm the compiler sees neither an LDS read (no wait in
// front) nor its result's latency (no wait before the use): the consumers' wait is an explicit `s_waitcnt lgkmcnt(0)` at the top of every
// sub-phase (KMB_TR_SYNC, fenced by sched_barriers: fragments are always consumed one sub-phase after they are requested) and behind the K loop of
// the persistent kernels (the next tile's first fragments live across the epilogue).  Sound only if no instruction names such a register between
// the read and the wait -- a property of the compiled code: tools/gemm_tr_asm_hazards.py walks the ISA's control-flow graph, and
// tests/test_cabi_cpu.py::test_gemm_asm_transposing_reads_are_waited_for runs it on every build.  Same arithmetic in the same order: outputs
// bit-identical to the intrinsic's (tools/gemm_tr_asm_ab.py: md5 per shape).  Kernels: v7 (+ the grouped weight gradients), v8, v11 with 256-wide
// tiles (four and eight waves) -- the ones compared on the GPU when this went in (round 5); v7d, the 128- / 192-wide v11 tiles and gemm_lean.hip:
// KMB_TR_ALL (round 6; gemm_lean.hip's L2 touch moved off v255 first -- with asm reads the allocator hands that register out).
// -DKMB_TR_BUILTIN: the intrinsic everywhere (A/B builds).  profiles/r05_gemm_transposing_reads_asm.md.
// Round 6: EVERY kernel (KMB_TR_ALL: also the four-stage kernel, the 128- / 192-wide persistent tiles and gemm_lean.hip) -- one GPU call compared
// the md5 of seven shapes x nine launch variants between the intrinsic build, round 5's partial build and this one: all identical and stable
// (profiles/r06_gemm_transposing_reads_all_variants.txt; the four-stage kernel's lone workgroups -33 %, the 128- / 192-wide weight gradients -12...-21 %).
#ifndef KMB_TR_BUILTIN
constexpr bool KMB_TR_ALL = true;
#else
constexpr bool KMB_TR_ALL = false;
#endif
#ifndef KMB_TR_BUILTIN
__device__ __forceinline__ s16x4 kmb_tr_read_asm(const char* ptr) {
  s16x4 t;
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)ptr;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(t) : "v"(a));
  return t;
}
// the same read at `addr` (32-bit LDS address in a register) + a compile-time byte offset in the instruction's offset field: one address
// register serves every (kk, hh) of a fragment column (gemm_lean.hip: without it each of the 16 reads of a stage kept its own hoisted address)
template <int OFF>
__device__ __forceinline__ s16x4 kmb_tr_read_asm_off(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  s16x4 t;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"(OFF));
  return t;
}
#define KMB_TR_SYNC()                                   \
  do {                                                  \
    __builtin_amdgcn_sched_barrier(0);                  \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0); the builtin, not asm: hipcc's own wait tracking then knows the LDS queue is empty (as asm it re-waited, lgkmcnt(0), in front of the next use of a plain fragment read -- right behind the asm reads just issued) */ \
    __builtin_amdgcn_sched_barrier(0);                  \
  } while (0)
#else
#define KMB_TR_SYNC() do { } while (0)
#endif
template <bool KC, bool ASM = false>
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
#ifndef KMB_TR_BUILTIN
      const s16x4 t = ASM ? kmb_tr_read_asm(lds + off) : __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
#else
      const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(lds + off));
#endif
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
  if (KMB_DIAG_BIT(p.tile_order, 512)) {   // epilogue ablation (diagnostic build only): keep the accumulators alive, write nothing
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) keep += acc[i][j][0];
    if (keep == 1.2345e30f) p.out_bf16[0] = 0;
    return;
  }
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
  KMB_STAMP(3);
  gemm_epilogue_phase2<NT>(p, ef, reinterpret_cast<float*>(smem + EPI_BYTES * (NT / 256)), tid, row0, col0, slice);
}

// cs: [NT/16][128] fp32 scratch (only touched when p.colsum != nullptr)
// The body is instantiated per epilogue class (ACT: 0 none / 1 GeLU / 2 GeLU' / -1 = read p.act at run time (tanh
// heads); RES: residual add; CS: column sums) and entered through ONE uniform branch: with every option tested per
// element inside the 8x-unrolled row loop the epilogue was ~90 KB of code walked once per tile -- instruction fetch,
// not the stores, made it 4.4 us of a 17 us tile (in-kernel stamps, tools/gemm_stamps.py).
// WAVE: one wave runs the body on its private 16-row staging chunk (v11); the column sums are carried across chunks
// in csum_io and finished by the caller.
template <int NT, bool HOIST, int NIT, int LD, bool SWZ, int TILE_ROWS, int ACT, bool RES, bool CS, bool FAST,
          bool WAVE = false, int WCOLS = 128>
__device__ __forceinline__ void gemm_epilogue_body(const KmbGemm& p, const float* ef, float* cs, int tid, int row0,
                                                   int col0, float* csum_io = nullptr) {
  constexpr int RPP = NT / 16;  // rows per pass
  constexpr int NH = HOIST ? NIT : 1;
  constexpr bool AUX = (ACT == 2 || ACT < 0);
  auto eoff = [&](int lrow, int c) {
    return lrow * LD + (WAVE ? (c ^ ((lrow & 7) << 3)) : SWZ ? (c ^ (((lrow >> 2) & 1) << 4)) : c);
  };
  const int act = ACT < 0 ? p.act : ACT;
  const bool res_on = RES && (ACT >= 0 || p.residual != nullptr);  // the run-time class checks its pointers
  const bool cs_on = CS && (ACT >= 0 || p.colsum != nullptr);
  const int c8 = (tid & 15) * 8;
  const int gcol = col0 + c8;
  if (WAVE && WCOLS < 128 && c8 >= WCOLS) return;   // a 96-column wave block: the last four column-lanes have no data
  // FAST: the whole tile lies inside C (a workgroup-uniform fact): no per-row / per-column edge handling at all
  if (!FAST && gcol >= p.N && !cs_on) return;
  const int nvalid = FAST ? 8 : (gcol >= p.N ? 0 : ((p.N - gcol) < 8 ? (p.N - gcol) : 8));
  const bool full8 = FAST || nvalid == 8;
  float csum[8], bias8[8], scale8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    csum[e] = WAVE ? csum_io[e] : 0.f;
    bias8[e] = (p.bias != nullptr && e < nvalid) ? p.bias[gcol + e] : 0.f;
    scale8[e] = (gcol + e < p.col_scale_n) ? p.col_scale : 1.0f;  // (v + bias) * scale, the reference's order
  }

  // ---- gather everything the row-iterations need first (LDS reads and global residual / aux loads are then in
  //      flight together instead of one dependent round trip per iteration) ----
  f32x4 vlo[NH], vhi[NH];
  u32x4 resv[NH], auxv[NH];
  auto gather = [&](int it, int slot) {
    const int lrow = (tid >> 4) + RPP * it;
    const int grow = row0 + lrow;
    const bool ok = FAST || (grow < p.M && full8);
    vlo[slot] = *reinterpret_cast<const f32x4*>(ef + eoff(lrow, c8));
    vhi[slot] = *reinterpret_cast<const f32x4*>(ef + eoff(lrow, c8) + 4);
    if (RES) resv[slot] = (ok && res_on) ? *reinterpret_cast<const u32x4*>(p.residual + (size_t)grow * p.ld_res + gcol) : u32x4{0u, 0u, 0u, 0u};
    if (AUX) auxv[slot] = (ok && (act == 2 || act == 4)) ? *reinterpret_cast<const u32x4*>(p.aux + (size_t)grow * p.ld_aux + gcol)
                                                         : u32x4{0u, 0u, 0u, 0u};
  };
  if (HOIST) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) gather(it, it);
  }
  const bool drop = p.drop_thr16 != 0u;
  const bool skip_stores = KMB_DIAG_BIT(p.tile_order, 256);  // ablation (diagnostic build only, tools/gemm_ablate.py): keep the math, drop the stores
  auto process = [&](int it0, int it) -> bool {
    const int lrow = (tid >> 4) + RPP * it0;
    const int grow = row0 + lrow;
    if (!FAST && (grow >= p.M || nvalid == 0)) return false;
    float v[8];
    v[0] = vlo[it][0]; v[1] = vlo[it][1]; v[2] = vlo[it][2]; v[3] = vlo[it][3];
    v[4] = vhi[it][0]; v[5] = vhi[it][1]; v[6] = vhi[it][2]; v[7] = vhi[it][3];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (v[e] + bias8[e]) * scale8[e];
    if (act == 1) {
      if (p.preact != nullptr) {   // GeLU and GeLU' from one evaluation; the DERIVATIVE is what backward needs (act 2)
        float dv[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          kmb_f32x2 y, dy;
          gelu_both2(kmb_f32x2{v[e], v[e + 1]}, y, dy);
          v[e] = y[0]; v[e + 1] = y[1];
          dv[e] = dy[0]; dv[e + 1] = dy[1];
        }
        if (full8) {
          *reinterpret_cast<u32x4*>(p.preact + (size_t)grow * p.ld_preact + gcol) = pack8(dv);
        } else {
          for (int e = 0; e < nvalid; ++e) p.preact[(size_t)grow * p.ld_preact + gcol + e] = f2bf(dv[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const kmb_f32x2 y = gelu2(kmb_f32x2{v[e], v[e + 1]});
          v[e] = y[0]; v[e + 1] = y[1];
        }
      }
    } else if (AUX && (act == 2 || act == 4)) {
      float u[8];
      if (full8) {
        unpack8(auxv[it], u);
      } else {
        for (int e = 0; e < 8; ++e) u[e] = e < nvalid ? bf2f(p.aux[(size_t)grow * p.ld_aux + gcol + e]) : 0.f;
      }
      if (act == 2) {   // aux = GeLU'(pre-activation), stored by the forward epilogue (act 1 with preact)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= u[e];
      } else {  // tanh'(.) = 1 - y^2, aux = y
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= (1.f - u[e] * u[e]);
      }
    } else if (ACT < 0 && act == 3) {  // BartClassificationHead: tanh
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
    }
    if (drop) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        v[e] = drop_keep(p.drop_seed, (uint32_t)grow, (uint32_t)(gcol + e), p.drop_thr16) ? v[e] * p.drop_scale : 0.f;
    }
    if (res_on) {
      float rr[8];
      if (full8) {
        unpack8(resv[it], rr);
      } else {
        for (int e = 0; e < 8; ++e) rr[e] = e < nvalid ? bf2f(p.residual[(size_t)grow * p.ld_res + gcol + e]) : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rr[e];
    }
    if (cs_on) {
#pragma unroll
      for (int e = 0; e < 8; ++e) csum[e] += v[e];
    }
    if (skip_stores) {
      float keep = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) keep += v[e];
      if (keep == 1.2345e30f) p.out_bf16[0] = 0;
      return true;
    }
    if (p.out_bf16 != nullptr) {
      if (full8) {
        u32x4* const dst = reinterpret_cast<u32x4*>(p.out_bf16 + (size_t)grow * p.ld_out_bf16 + gcol);
        if (WAVE) KMB_NT_STORE(pack8(v), dst);   // streaming: C must not evict the A / B panels from L2
        else *dst = pack8(v);
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
  if (WAVE) {
#pragma unroll
    for (int e = 0; e < 8; ++e) csum_io[e] = csum[e];
  } else if (cs_on) {
    // column sums of this tile's stored values: reduce the row-lanes through LDS, one partial row per 128 rows
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[(tid >> 4) * 128 + c8 + e] = csum[e];
    __syncthreads();
    if (tid < 128 && (FAST || col0 + tid < p.N)) {
      float t = 0.f;
#pragma unroll 8
      for (int k = 0; k < RPP; ++k) t += cs[k * 128 + tid];
      // contract: one partial row per 64 rows of C; this tile fills its first row and zeroes the rest
      const int prow = row0 >> 6;
      p.colsum[(size_t)prow * p.N + col0 + tid] = t;
      for (int k = 1; k < TILE_ROWS / 64; ++k)
        if (FAST || row0 + 64 * k < p.M) p.colsum[(size_t)(prow + k) * p.N + col0 + tid] = 0.f;
    }
  }
}

template <int NT, bool HOIST, int NIT, int LD, bool SWZ, int TILE_ROWS>
__device__ __forceinline__ void gemm_epilogue_phase2(const KmbGemm& p, const float* ef, float* cs, int tid, int row0,
                                                     int col0, int slice) {
  constexpr int RPP = NT / 16;
  auto eoff = [&](int lrow, int c) { return lrow * LD + (SWZ ? (c ^ (((lrow >> 2) & 1) << 4)) : c); };
  if (p.split_k > 1) {  // raw partial sums of this K slice -> slab[slice][M][N]
    const int c8 = (tid & 15) * 8;
    const int gcol = col0 + c8;
    if (gcol >= p.N) return;
    const int nvalid = (p.N - gcol) < 8 ? (p.N - gcol) : 8;
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
  const bool interior = (row0 + TILE_ROWS <= p.M) && (col0 + 128 <= p.N);   // uniform per workgroup
#define KMB_EPI(ACT, RES, CS)                                                                                     \
  do {                                                                                                            \
    if (interior) gemm_epilogue_body<NT, HOIST, NIT, LD, SWZ, TILE_ROWS, ACT, RES, CS, true>(p, ef, cs, tid, row0, col0);  \
    else gemm_epilogue_body<NT, HOIST, NIT, LD, SWZ, TILE_ROWS, ACT, RES, CS, false>(p, ef, cs, tid, row0, col0);          \
  } while (0)
  const bool res = p.residual != nullptr, csf = p.colsum != nullptr;
  if (p.act == 0) {
    if (!res && !csf) KMB_EPI(0, false, false);
    else if (res && !csf) KMB_EPI(0, true, false);
    else if (!res) KMB_EPI(0, false, true);
    else KMB_EPI(0, true, true);
  } else if (p.act == 1 && !res && !csf) {
    KMB_EPI(1, false, false);
  } else if (p.act == 2 && !res) {
    if (csf) KMB_EPI(2, false, true);
    else KMB_EPI(2, false, false);
  } else {
    // everything else (tanh heads, rare combinations): run-time act, non-hoisted compact loop
    gemm_epilogue_body<NT, false, NIT, LD, SWZ, TILE_ROWS, -1, true, true, false>(p, ef, cs, tid, row0, col0);
  }
#undef KMB_EPI
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
  const int ntl = tiles_n * ((p.M + BM - 1) / BM);
  const bool slice_major = (p.tile_order & 4) != 0 && nsl > 1;   // see split_order_note
  const int tile = slice_major ? bid % ntl : bid / nsl, slice = slice_major ? bid / ntl : bid % nsl;
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
// v7: 128x128x64 tile, 256 threads = 4 waves (2x2) of 64x64, two workgroups per CU.  Global -> LDS staging is
// LDS-DMA (global_load_lds, 16 B per lane: no staging VGPRs, no ds_write pass).  A wave instruction writes 1 KiB of
// LDS linearly (base + lane*16), so the XOR swizzle is applied to the per-lane SOURCE address instead: lane i of
// piece p lands on physical chunk (i % 8 or i % 16) of its row and therefore fetches the LOGICAL chunk that the
// swizzle maps there.  Requires K % 64 == 0 (no zero-fill on this path; v1 handles ragged K).
// The K loop is software-pipelined.  (A first LDS-DMA kernel issued its 8 pieces in one burst -- each costs 60-180
// issue cycles -- and then alternated "read fragments / wait / 8 MFMAs": the matrix pipe idled during the burst and
// during every LDS round trip.)  Each K step is two phases around ONE barrier:
//   phase A: MFMAs of k-half 0 (fragments already in registers)  ||  ds_reads of k-half 1
//   wait for the stage t+1 DMA, barrier (everyone is done reading stage t's LDS image)
//   phase B: MFMAs of k-half 1  ||  ds_reads of stage t+1's k-half 0  ||  DMA of stage t+2 into the freed buffer
// with sched_group_barrier interleaving DS / VMEM issue between the MFMAs.  DMA sources are a uniform (SGPR) tile
// base that advances per K step plus a per-lane 32-bit offset computed once.
template <bool KC>
__device__ __forceinline__ void dma_offsets(uint32_t (&off)[4], int ld, int r0, int R, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;
    if (KC) {
      const int row = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      int grow = r0 + row;
      grow = grow < R ? grow : R - 1;
      off[i] = (uint32_t)(((grow - r0) * ld + c * 8) * 2);
    } else {
      const int krow = piece * 4 + (lane >> 4);
      const int ps = lane & 15;
      const int c32 = (ps >> 1) ^ swz_nkc(krow);
      int m = r0 + c32 * 16 + (ps & 1) * 8;
      const int mlast = ((R - 1) >> 3) << 3;
      m = m < R ? m : mlast;
      off[i] = (uint32_t)((krow * ld + (m - r0)) * 2);
    }
  }
}

__device__ __forceinline__ void dma_piece(const char* gbase, uint32_t off, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + off),
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// the same with the non-temporal cache policy (aux bit 1 = nt on gfx950): the line is allocated in L2 as the first to leave.  For
// the STREAMED operand of a persistent launch (the activation panel: read by the column tiles of one round, never again) so
// that it does not push the REUSED one (the weight panels, re-read every round) out of a 4 MB L2.  Experiment: -DKMB_A_NT.
__device__ __forceinline__ void dma_piece_nt(const char* gbase, uint32_t off, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + off),
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 2);
}

// One 128x128 output tile (block `block` of `nblocks` of problem p): the body of gemm_kernel_v7 and of the grouped launch below
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void v7_tile(const KmbGemm& p, char* smem, int block, int nblocks) {
  KMB_STAMP(0);
  KMB_STAMP_ID();
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nsl = p.split_k > 1 ? p.split_k : 1;
  const int bid = (p.tile_order & 1) ? xcd_remap(block, nblocks) : block;
  const int ntl = tiles_n * ((p.M + BM - 1) / BM);
  const bool slice_major = (p.tile_order & 4) != 0 && nsl > 1;   // see split_order_note
  const int tile = slice_major ? bid % ntl : bid / nsl, slice = slice_major ? bid / ntl : bid % nsl;
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

  uint32_t offA[4], offB[4];
  dma_offsets<A_KC>(offA, p.lda, row0, p.M, wave, lane);
  dma_offsets<B_KC>(offB, p.ldb, col0, p.N, wave, lane);
  const size_t stepA = A_KC ? (size_t)BK * 2 : (size_t)BK * p.lda * 2;
  const size_t stepB = B_KC ? (size_t)BK * 2 : (size_t)BK * p.ldb * 2;
  const char* gA = reinterpret_cast<const char*>(p.A) +
                   (A_KC ? ((size_t)row0 * p.lda + (size_t)t_begin * BK) * 2 : (size_t)row0 * 2) +
                   (A_KC ? 0 : (size_t)t_begin * stepA);
  const char* gB = reinterpret_cast<const char*>(p.B) +
                   (B_KC ? ((size_t)col0 * p.ldb + (size_t)t_begin * BK) * 2 : (size_t)col0 * 2) +
                   (B_KC ? 0 : (size_t)t_begin * stepB);
  constexpr int A_TILE = BM * BK * 2;
  char* const dstA = smem + wave * 4096;            // this wave's 4 pieces of the A image (stage 0)
  char* const dstB = smem + A_TILE + wave * 4096;

  // K step `ks` (relative to t_begin) of this tile -> LDS stage buffer; the tile base is forced into SGPRs so the
  // loads take the saddr + 32-bit-voffset form (no per-piece 64-bit VALU address arithmetic)
  auto uniform_ptr = [](const char* ptr) {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  auto dma_stage = [&](int ks, int stage_buf) {
    char* da = dstA + stage_buf * STAGE_BYTES;
    char* db = dstB + stage_buf * STAGE_BYTES;
    const char* ga = uniform_ptr(gA + (size_t)ks * stepA);
    const char* gb = uniform_ptr(gB + (size_t)ks * stepB);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(ga, offA[i], da + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(gb, offB[i], db + i * 1024);
  };

  bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
  dma_stage(0, 0);
  if (nt > 1) {
    dma_stage(1, 1);
    __builtin_amdgcn_s_waitcnt(0x0F78);  // vmcnt(8): stage 0 has landed
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  }
  __syncthreads();
  KMB_STAMP(1);
#pragma unroll
  for (int i = 0; i < 4; ++i) fa0[i] = read_frag<A_KC, true>(smem, wm * 4 + i, 0, r, g);
#pragma unroll
  for (int j = 0; j < 4; ++j) fb0[j] = read_frag<B_KC, true>(smem + A_TILE, wn * 4 + j, 0, r, g);

  constexpr int NDS = (A_KC ? 4 : 8) + (B_KC ? 4 : 8);  // ds_read instructions per fragment set
  auto kstep = [&](int t, auto do_dma, auto do_next) {
    const char* cur = smem + (t & 1) * STAGE_BYTES;
    const char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
    // ---- phase A ----
    if constexpr (!(A_KC && B_KC)) KMB_TR_SYNC();   // the (asm-read) fragments requested in phase B of the previous step have arrived
#pragma unroll
    for (int i = 0; i < 4; ++i) fa1[i] = read_frag<A_KC, true>(cur, wm * 4 + i, 1, r, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb1[j] = read_frag<B_KC, true>(cur + A_TILE, wn * 4 + j, 1, r, g);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0[i], fb0[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA first: they wait on phase B's reads, not on these
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x100, NDS / 4, 0);  // DS read
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);        // MFMA
    }
    __builtin_amdgcn_sched_group_barrier(0x100, NDS / 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): the builtin (not inline asm) so the compiler's own
                                         // wait tracking knows the fragments have arrived
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase B ----  (the DMA overwrites LDS the fragment reads may alias: reads first, DMA pieces behind them)
    if (decltype(do_next)::value) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa0[i] = read_frag<A_KC, true>(nxt, wm * 4 + i, 0, r, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb0[j] = read_frag<B_KC, true>(nxt + A_TILE, wn * 4 + j, 0, r, g);
    }
    if (decltype(do_dma)::value) dma_stage(t + 2, t & 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1[i], fb1[j], acc[i][j], 0, 0, 0);
    if (decltype(do_next)::value && decltype(do_dma)::value) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, NDS / 4, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 1);  // VMEM (LDS-DMA)
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
      }
    } else if (decltype(do_next)::value) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, NDS / 4, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using Yes = std::true_type;
  using No = std::false_type;
  int t = 0;
  for (; t + 2 < nt; ++t) kstep(t, Yes{}, Yes{});
  if (t + 1 < nt) { kstep(t, No{}, Yes{}); ++t; }
  kstep(t, No{}, No{});
  __syncthreads();
  KMB_STAMP(2);
  gemm_epilogue<256>(p, smem, acc, tid, wm, wn, r, g, row0, col0, slice);
  KMB_STAMP(4);
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_kernel_v7(const KmbGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v7_tile<A_KC, B_KC>(p, smem, (int)blockIdx.x, (int)gridDim.x);
}

// v7d ("deep", launch variant 5; round 5): the same tile with FOUR LDS stages instead of two, for launches that put at most one
// workgroup on a CU (<= 256 workgroups: M <= 4096 rows of N = 768, the reference's default batch of 64).  There v7's K step
// is not bound by the MFMAs (0.21 us for its 32 per wave) but by the latency of the ONE stage it has in flight -- about 1 us
// per 64-deep step with nobody else on the CU to hide it (4096 x 768 x 3072: 52-57 us inside a b = 64 step = 48 steps).  Four
// stages keep three in flight behind a counted wait (s_waitcnt vmcnt(16): the in-order return path guarantees the oldest
// stage has landed); 128 KB of LDS, so one workgroup per CU by construction.  Same fragment reads, same MFMA order: bit-identical.
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void v7d_tile(const KmbGemm& p, char* smem, int block, int nblocks) {
  KMB_STAMP(0);
  KMB_STAMP_ID();
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nsl = p.split_k > 1 ? p.split_k : 1;
  const int bid = (p.tile_order & 1) ? xcd_remap(block, nblocks) : block;
  const int ntl = tiles_n * ((p.M + BM - 1) / BM);
  const bool slice_major = (p.tile_order & 4) != 0 && nsl > 1;   // see split_order_note
  const int tile = slice_major ? bid % ntl : bid / nsl, slice = slice_major ? bid / ntl : bid % nsl;
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

  uint32_t offA[4], offB[4];
  dma_offsets<A_KC>(offA, p.lda, row0, p.M, wave, lane);
  dma_offsets<B_KC>(offB, p.ldb, col0, p.N, wave, lane);
  const size_t stepA = A_KC ? (size_t)BK * 2 : (size_t)BK * p.lda * 2;
  const size_t stepB = B_KC ? (size_t)BK * 2 : (size_t)BK * p.ldb * 2;
  const char* gA = reinterpret_cast<const char*>(p.A) +
                   (A_KC ? ((size_t)row0 * p.lda + (size_t)t_begin * BK) * 2 : (size_t)row0 * 2) +
                   (A_KC ? 0 : (size_t)t_begin * stepA);
  const char* gB = reinterpret_cast<const char*>(p.B) +
                   (B_KC ? ((size_t)col0 * p.ldb + (size_t)t_begin * BK) * 2 : (size_t)col0 * 2) +
                   (B_KC ? 0 : (size_t)t_begin * stepB);
  constexpr int A_TILE = BM * BK * 2;
  char* const dstA = smem + wave * 4096;            // this wave's 4 pieces of the A image (stage 0)
  char* const dstB = smem + A_TILE + wave * 4096;

  // K step `ks` (relative to t_begin) of this tile -> LDS stage buffer; the tile base is forced into SGPRs so the
  // loads take the saddr + 32-bit-voffset form (no per-piece 64-bit VALU address arithmetic)
  auto uniform_ptr = [](const char* ptr) {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  auto dma_stage = [&](int ks, int stage_buf) {
    char* da = dstA + stage_buf * STAGE_BYTES;
    char* db = dstB + stage_buf * STAGE_BYTES;
    const char* ga = uniform_ptr(gA + (size_t)ks * stepA);
    const char* gb = uniform_ptr(gB + (size_t)ks * stepB);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(ga, offA[i], da + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(gb, offB[i], db + i * 1024);
  };

  bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
  dma_stage(0, 0);
  if (nt > 1) dma_stage(1, 1);
  if (nt > 2) dma_stage(2, 2);
  if (nt > 3) dma_stage(3, 3);
  // stage 0 has landed once at most the pieces of the younger stages (eight per wave and stage) are outstanding
  if (nt > 3) __builtin_amdgcn_s_waitcnt(0x4F78);        // vmcnt(24)
  else if (nt > 2) __builtin_amdgcn_s_waitcnt(0x4F70);   // vmcnt(16)
  else if (nt > 1) __builtin_amdgcn_s_waitcnt(0x0F78);   // vmcnt(8)
  else __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0)
  __builtin_amdgcn_s_barrier();   // (not __syncthreads(): its fence is a vmcnt(0) -- the three younger stages stay in flight here)
  KMB_STAMP(1);
#pragma unroll
  for (int i = 0; i < 4; ++i) fa0[i] = read_frag<A_KC, KMB_TR_ALL>(smem, wm * 4 + i, 0, r, g);
#pragma unroll
  for (int j = 0; j < 4; ++j) fb0[j] = read_frag<B_KC, KMB_TR_ALL>(smem + A_TILE, wn * 4 + j, 0, r, g);

  constexpr int NDS = (A_KC ? 4 : 8) + (B_KC ? 4 : 8);  // ds_read instructions per fragment set
  // in_flight: stages younger than t + 1 that may still be outstanding at this step's wait (2, 1 or 0)
  auto kstep = [&](int t, auto do_dma, auto do_next, auto in_flight) {
    const char* cur = smem + (t & 3) * STAGE_BYTES;
    const char* nxt = smem + ((t + 1) & 3) * STAGE_BYTES;
    // ---- phase A ----
    if constexpr (KMB_TR_ALL && !(A_KC && B_KC)) KMB_TR_SYNC();
#pragma unroll
    for (int i = 0; i < 4; ++i) fa1[i] = read_frag<A_KC, KMB_TR_ALL>(cur, wm * 4 + i, 1, r, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb1[j] = read_frag<B_KC, KMB_TR_ALL>(cur + A_TILE, wn * 4 + j, 1, r, g);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0[i], fb0[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA first: they wait on phase B's reads, not on these
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x100, NDS / 4, 0);  // DS read
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);        // MFMA
    }
    __builtin_amdgcn_sched_group_barrier(0x100, NDS / 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    // lgkmcnt(0) + vmcnt(8 x in_flight): stage t + 1 has landed; the builtin (not inline asm) so the compiler's own wait
    // tracking knows the fragments have arrived
    if (decltype(in_flight)::value == 2) __builtin_amdgcn_s_waitcnt(0x4070);
    else if (decltype(in_flight)::value == 1) __builtin_amdgcn_s_waitcnt(0x0078);
    else __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase B ----  (the DMA overwrites LDS the fragment reads may alias: reads first, DMA pieces behind them)
    if (decltype(do_next)::value) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa0[i] = read_frag<A_KC, KMB_TR_ALL>(nxt, wm * 4 + i, 0, r, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb0[j] = read_frag<B_KC, KMB_TR_ALL>(nxt + A_TILE, wn * 4 + j, 0, r, g);
    }
    if (decltype(do_dma)::value) dma_stage(t + 4, t & 3);   // into the buffer this step has just read for the last time
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1[i], fb1[j], acc[i][j], 0, 0, 0);
    if (decltype(do_next)::value && decltype(do_dma)::value) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, NDS / 4, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 1);  // VMEM (LDS-DMA)
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
      }
    } else if (decltype(do_next)::value) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, NDS / 4, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using Yes = std::true_type;
  using No = std::false_type;
  using I2 = std::integral_constant<int, 2>;
  using I1 = std::integral_constant<int, 1>;
  using I0 = std::integral_constant<int, 0>;
  int t = 0;
  for (; t + 4 < nt; ++t) kstep(t, Yes{}, Yes{}, I2{});                // stages t + 1 .. t + 3 outstanding, t + 4 requested
  if (nt - t == 4) { kstep(t, No{}, Yes{}, I2{}); ++t; }
  if (nt - t == 3) { kstep(t, No{}, Yes{}, I1{}); ++t; }
  if (nt - t == 2) { kstep(t, No{}, Yes{}, I0{}); ++t; }
  kstep(t, No{}, No{}, I0{});
  __syncthreads();
  KMB_STAMP(2);
  gemm_epilogue<256>(p, smem, acc, tid, wm, wn, r, g, row0, col0, slice);
  KMB_STAMP(4);
}


constexpr int LDS_DEEP = 4 * STAGE_BYTES;   // 128 KB (the epilogue's fp32 image + column sums fit in it)
static_assert(LDS_DEEP >= LDS_BYTES, "the deep variant's epilogue uses the stage buffers");
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_kernel_v7d(const KmbGemm p) {   // (2: the register budget of gemm_kernel_v7 -- 128 KB of LDS keep it at one workgroup per CU; with the 512-register budget of (256, 1) hipcc rotated the accumulators through ~100 v_accvgpr moves per K step)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v7d_tile<A_KC, B_KC>(p, smem, (int)blockIdx.x, (int)gridDim.x);
}

#ifndef KMB_GEMM_DEVICE_ONLY
// Grouped weight gradients (round 5): ONE launch walks the 128x128 tiles of up to KMB_GEMM_GROUP_MAX independent problems of the
// weight-gradient layout (dW = dY^T X: both operands token-major), each tile over the WHOLE token reduction -- no K slices,
// no fp32 slabs, no reduction pass.  For the small batches of the reference's default (64 samples: 2048-4096 tokens) a layer's
// four to six weight gradients have 36-144 tiles each: alone they need 7-fold split-K to cover the chip (63 GEMMs + 91 slab
// reductions per step on the side stream, each behind an event pair).  Together they are 432-504 tiles = one round of two
// workgroups per CU, 16 launches a step.  Used for <= 3072 tokens (kmb_backward's rule; DESIGN.md section 4, round 5: the
// step gets 1.5-5 % faster at b = 8 .. 48; from b = 64 on a launch that holds every workgroup slot of the chip for 85-160 us
// delays the caller's stream more than the split-K launches did, and the rule leaves those batches on them).  Blocks of problem k are
// [first[k], first[k + 1]) with every first[k] a multiple of 8 (the per-XCD tile ranges of xcd_remap stay aligned with the
// hardware's round-robin; surplus blocks exit).  Same tile body as gemm_kernel_v7: bit-identical to the problem launched alone
// without split-K.
struct KmbGemmGroup {
  int n;
  int first[KMB_GEMM_GROUP_MAX + 1];
  int blocks[KMB_GEMM_GROUP_MAX];
  KmbGemm p[KMB_GEMM_GROUP_MAX];
};

__global__ __launch_bounds__(256, 2) void gemm_group_wgrad_kernel(const KmbGemmGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = (int)blockIdx.x;
  int k = 0;
#pragma unroll
  for (int i = 1; i < KMB_GEMM_GROUP_MAX; ++i)
    if (i < grp.n && b >= grp.first[i]) k = i;
  const int local = b - grp.first[k];
  if (local >= grp.blocks[k]) return;   // padding up to the next multiple of 8
  v7_tile<false, false>(grp.p[k], smem, local, grp.blocks[k]);
}
#endif

// ------------------------------------------------------------------------------------------
// fragment reads for tiles that are 256 rows (columns) tall (v8)

template <bool KC, int ROWS, bool ASM = false>
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
#ifndef KMB_TR_BUILTIN
      const s16x4 t = ASM ? kmb_tr_read_asm(lds + off) : __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
#else
      const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(lds + off));
#endif
      out[hh * 4 + 0] = t[0]; out[hh * 4 + 1] = t[1]; out[hh * 4 + 2] = t[2]; out[hh * 4 + 3] = t[3];
    }
    return out;
  }
}

// 256x256x64 tile, 512 threads = 8 waves (2x4), each wave 128x64 (8x4 MFMA tiles); two 64 KB stages filled by
// LDS-DMA; epilogue in two column halves (the fp32 staging image does not fit otherwise).
constexpr int BM4 = 256, BN4 = 256;
constexpr int ST4 = (BM4 + BN4) * BK * 2;   // 64 KB
constexpr int LDS4 = ((2 * ST4 > 2 * EPI_BYTES) ? 2 * ST4 : 2 * EPI_BYTES) + CS512;


// ------------------------------------------------------------------------------------------
// v8: the 256x256 tile with v7's software-pipelined K loop (same per-element accumulation order: bit-identical).  A 128x128 tile needs one 1 KiB LDS-DMA piece per 16 MFMAs, which keeps the CU's
// texture-address path as busy as its matrix pipe (measured: 0.81 us per K step for two co-resident workgroups,
// 0.49 us of it MFMA); the 256x256 tile halves that ratio.  Each K step is four sub-phases of 16 MFMAs per wave:
//   0: A(k0, rows 0-63) x B(k0)      || read A(k0, rows 64-127)
//   1: A(k0, rows 64-127) x B(k0)    || read A(k1, rows 0-63), B(k1)
//   2: A(k1, rows 0-63) x B(k1)      || read A(k1, rows 64-127);  wait DMA(t+1), barrier
//   3: A(k1, rows 64-127) x B(k1)    || read A/B(k0) of stage t+1, then DMA stage t+2 into the freed buffer
template <bool KC>
__device__ __forceinline__ void dma_offsets256(uint32_t (&off)[4], int ld, int r0, int R, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;  // 32 pieces of 1 KiB per 256-row tile
    if (KC) {
      const int row = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      int grow = r0 + row;
      grow = grow < R ? grow : R - 1;
      off[i] = (uint32_t)(((grow - r0) * ld + c * 8) * 2);
    } else {
      const int krow = piece * 2 + (lane >> 5);
      const int ps = lane & 31;
      const int c32 = (ps >> 1) ^ swz_nkc(krow);
      int m = r0 + c32 * 16 + (ps & 1) * 8;
      const int mlast = ((R - 1) >> 3) << 3;
      m = m < R ? m : mlast;
      off[i] = (uint32_t)((krow * ld + (m - r0)) * 2);
    }
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512) void gemm_kernel_v8(const KmbGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  KMB_STAMP(0);
  KMB_STAMP_ID();
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, g = lane >> 4;
  const int tiles_n = (p.N + BN4 - 1) / BN4;
  const int nsl = p.split_k > 1 ? p.split_k : 1;
  const int bid = (p.tile_order & 1) ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int ntl = tiles_n * ((p.M + BM4 - 1) / BM4);
  const bool slice_major = (p.tile_order & 4) != 0 && nsl > 1;   // see split_order_note
  const int tile = slice_major ? bid % ntl : bid / nsl, slice = slice_major ? bid / ntl : bid % nsl;
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

  uint32_t offA[4], offB[4];
  dma_offsets256<A_KC>(offA, p.lda, row0, p.M, wave, lane);
  dma_offsets256<B_KC>(offB, p.ldb, col0, p.N, wave, lane);
  const size_t stepA = A_KC ? (size_t)BK * 2 : (size_t)BK * p.lda * 2;
  const size_t stepB = B_KC ? (size_t)BK * 2 : (size_t)BK * p.ldb * 2;
  const char* gA = reinterpret_cast<const char*>(p.A) + (A_KC ? (size_t)row0 * p.lda * 2 : (size_t)row0 * 2) + (size_t)t_begin * stepA;
  const char* gB = reinterpret_cast<const char*>(p.B) + (B_KC ? (size_t)col0 * p.ldb * 2 : (size_t)col0 * 2) + (size_t)t_begin * stepB;
  char* const dstA = smem + wave * 4096;
  char* const dstB = smem + A_BYTES + wave * 4096;
  auto uniform_ptr = [](const char* ptr) {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  // a stage is fetched in two halves one sub-phase apart: an LDS-DMA piece costs the issuing wave 60-180 cycles, and all
  // of a stage's pieces in sub-phase 3 made that sub-phase as long as two others (measured on v11: K loop -11 %)
  auto dma_a = [&](int ks, int stage_buf) {
    char* da = dstA + stage_buf * ST4;
    const char* ga = uniform_ptr(gA + (size_t)ks * stepA);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(ga, offA[i], da + i * 1024);
  };
  auto dma_b = [&](int ks, int stage_buf) {
    char* db = dstB + stage_buf * ST4;
    const char* gb = uniform_ptr(gB + (size_t)ks * stepB);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(gb, offB[i], db + i * 1024);
  };

  constexpr int NDA = A_KC ? 4 : 8;  // ds_read instructions per 4 A fragments / per 4 B fragments
  constexpr int NDB = B_KC ? 4 : 8;
  bf16x8 fa[2][4], fb[2][4];
  [[maybe_unused]] uint64_t kmb_wait_ticks = 0;  // diagnostic build only
  auto read_a = [&](const char* stage, int kk, int half, bf16x8 (&dst)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = read_frag3<A_KC, BM4, true>(stage, wm * 8 + half * 4 + i, kk, r, g);
  };
  auto read_b = [&](const char* stage, int kk, bf16x8 (&dst)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = read_frag3<B_KC, BN4, true>(stage + A_BYTES, wn * 4 + j, kk, r, g);
  };
  auto mma = [&](int half, const bf16x8 (&a)[4], const bf16x8 (&b)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[half * 4 + i][j], 0, 0, 0);
  };

  dma_a(0, 0);
  dma_b(0, 0);
  if (nt > 1) {
    dma_a(1, 1);                         // its B half goes out in the first K step's sub-phase 0
    __builtin_amdgcn_s_waitcnt(0x0F74);  // vmcnt(4): stage 0 has landed
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  }
  __syncthreads();
  KMB_STAMP(1);
  read_b(smem, 0, fb[0]);
  read_a(smem, 0, 0, fa[0]);

  auto kstep = [&](int t, auto do_dma, auto do_next) {
    const char* cur = smem + (t & 1) * ST4;
    const char* nxt = smem + ((t + 1) & 1) * ST4;
    // ---- sub-phase 0 ----
    if constexpr (!(A_KC && B_KC)) KMB_TR_SYNC();   // the (asm-read) fragments requested in the previous sub-phase have arrived
    read_a(cur, 0, 1, fa[1]);
    if (decltype(do_next)::value) dma_b(t + 1, (t + 1) & 1);   // B half of stage t+1 (its A half: previous sub-phase 3)
    mma(0, fa[0], fb[0]);
    if (decltype(do_next)::value) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, NDA / 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);  // VMEM (LDS-DMA)
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    } else {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, NDA / 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- sub-phase 1 ----
    if constexpr (!(A_KC && B_KC)) KMB_TR_SYNC();
    read_b(cur, 1, fb[1]);
    read_a(cur, 1, 0, fa[0]);
    mma(1, fa[1], fb[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, NDB, 1);  // B(k1) does not overwrite anything in use
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 1);
    __builtin_amdgcn_sched_group_barrier(0x100, NDA, 1);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- sub-phase 2 ----
    if constexpr (!(A_KC && B_KC)) KMB_TR_SYNC();
    read_a(cur, 1, 1, fa[1]);
    mma(0, fa[0], fb[1]);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 2);
    __builtin_amdgcn_sched_group_barrier(0x100, NDA, 2);
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 2);
    __builtin_amdgcn_sched_barrier(0);
    {
      KMB_WAIT_BEGIN();
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): stage t+1 landed, this wave is done reading stage t
      __builtin_amdgcn_s_barrier();
      KMB_WAIT_END(kmb_wait_ticks);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- sub-phase 3 ----
    if (decltype(do_next)::value) {
      read_b(nxt, 0, fb[0]);
      read_a(nxt, 0, 0, fa[0]);
    }
    if (decltype(do_dma)::value) dma_a(t + 2, t & 1);
    mma(1, fa[1], fb[1]);
    if (decltype(do_next)::value && decltype(do_dma)::value) {
      __builtin_amdgcn_sched_group_barrier(0x100, NDB, 3);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 3);
      __builtin_amdgcn_sched_group_barrier(0x100, NDA, 3);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 3);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 3);  // VMEM (LDS-DMA)
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 3);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using Yes = std::true_type;
  using No = std::false_type;
  int t = 0;
  for (; t + 2 < nt; ++t) kstep(t, Yes{}, Yes{});
  if (t + 1 < nt) { kstep(t, No{}, Yes{}); ++t; }
  kstep(t, No{}, No{});
  __syncthreads();
  KMB_STAMP(2);
  KMB_STAMP_VALUE(5, kmb_wait_ticks);
  if (KMB_DIAG_BIT(p.tile_order, 512)) {   // epilogue ablation (diagnostic build only)
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) keep += acc[i][j][0];
    if (keep == 1.2345e30f) p.out_bf16[0] = 0;
    return;
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
    if (h == 0) KMB_STAMP(3);
    // HOIST = false: the other column half's accumulators are still live, hoisting the loads spills 242 VGPRs
    gemm_epilogue_phase2<512, false>(p, ef, reinterpret_cast<float*>(smem + 2 * EPI_BYTES), tid, row0, col0 + h * 128, slice);
    __syncthreads();
  }
  KMB_STAMP(4);
}


#ifndef KMB_GEMM_DEVICE_ONLY   // (gemm_rolesplit.hip includes this file for the device helpers above and below only)
// ------------------------------------------------------------------------------------------
// "All rows" kernel: the vocabulary projection of a generation decode step (R = batch x beams <= 320 rows, N = 50320, K = 768,
// forward layout, fp32 logits).  The 128x128 kernel re-reads the R activation rows for each of 394 column tiles and the tied
// matrix for each of 3 row tiles: 463 MB through the CUs' memory pipes for 141 MB of operands, and that pipe (~50 GB/s per CU),
// not the MFMAs, is what a decode-sized GEMM waits for (51-71 us inside a step).  Here ONE workgroup holds all rows: tile =
// 320 rows x 256 columns, eight waves 2 x 4 of 160 x 64 (10 x 4 MFMA tiles, 160 accumulator registers), LDS stages
// filled by LDS-DMA, 197 workgroups = one round, each weight row read exactly once: 174 MB.  Rounds 4-5: two 72 KB stages of 64 k,
// a plain loop (wait, barrier, 80 MFMAs, barrier, next fetch); round 6: four 36 KB stages of 32 k, see below.
// Same MFMA, same k order, fp32 bias add: bit-identical to every other variant.
// Rows past M are clamped copies of the last row (computed, never stored).
// (Measured and dropped at the end of round 4: 208-column tiles -- 242 workgroups instead of 197, 26 KB of the tied matrix per K step
// and workgroup instead of 32 -- 47.0 -> 47.8 us: what is saved on the weight stream comes back as 45 more copies of the row panel.)
constexpr int VR = 320, VN = 256;
// Round 6: FOUR stages of 32 k (36 KB each, the same 144 KB) instead of two of 64.  The stamps (tools/allrows_stamps.py) put a 64-deep
// step of the two-stage loop at 2.4-2.8 us -- 1.1 us of MFMAs and a wait for ONE 72 KB stage that was issued a step earlier: with a
// single stage in flight behind the barrier a CU drew 27 GB/s through its memory pipe, half of what it takes with a deep queue.  Now
// a step is: wait for the oldest stage, ONE barrier (it also says every wave is done reading the stage before), issue the stage three
// ahead into that freed buffer, 40 MFMAs -- three stages (108 KB) are in flight under the MFMAs.  A stage image is [rows][32 k] =
// 64-byte rows, sixteen rows per 1 KB LDS-DMA piece = one MFMA row tile; the 16-byte k chunk c of row r sits in slot c ^ (-(r >> 2) & 3):
// ds_read_b128 serves the lanes in the groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (guide, LDS section), and with lane =
// 16 g + r reading chunk g of row r each group then covers all sixteen 16-byte slots of the 256-byte bank row ((r >> 2) & 3 instead of
// its negative: two lanes per slot, measured).  Same MFMA, same k order (stage s = the old step s / 2, half s & 1):
// bit-identical sums.  Pieces per stage: 20 of the row panel + 16 of the tied matrix = 5 for waves 0-3, 4 for waves 4-7 (the counted
// waits differ by wave; the branch is scalar).
constexpr int VK = 32;
constexpr int V_A = VR * VK * 2;                  // 20 KB
constexpr int V_STG = (VR + VN) * VK * 2;         // 36 KB
constexpr int V_NSTG = 4;
constexpr int LDS_VOC = V_NSTG * V_STG;           // 144 KB

template <bool STATS>
__global__ __launch_bounds__(512) void gemm_kernel_allrows(const KmbGemm p, float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, g = lane >> 4;
  const int col0 = (int)blockIdx.x * VN;
  f32x4 acc[10][4];
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ns = p.K / VK;
  // lane l of a piece fills LDS bytes [16 l, 16 l + 16): row l >> 2 of the piece's sixteen, slot l & 3 = chunk (l & 3) ^ (-(l >> 4) & 3)
  const int prow = lane >> 2, pchunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  uint32_t offA[3], offB[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {   // row-panel pieces wave, wave + 8 and (waves 0-3) wave + 16
    const int row = (wave + 8 * i) * 16 + prow;
    const int grow = row < p.M ? row : p.M - 1;
    offA[i] = (uint32_t)((grow * p.lda + pchunk * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {   // tied-matrix pieces 2 wave, 2 wave + 1
    const int col = col0 + (2 * wave + i) * 16 + prow;
    const int gcol = col < p.N ? col : p.N - 1;
    offB[i] = (uint32_t)(((gcol - col0) * p.ldb + pchunk * 8) * 2);
  }
  auto uniform_ptr = [](const char* ptr) {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  const char* const gA = reinterpret_cast<const char*>(p.A);
  const char* const gB = reinterpret_cast<const char*>(p.B) + (size_t)col0 * p.ldb * 2;
  auto dma_stage = [&](int s) {
    char* st = smem + (s & (V_NSTG - 1)) * V_STG;
    const char* ga = uniform_ptr(gA + (size_t)s * VK * 2);
    const char* gb = uniform_ptr(gB + (size_t)s * VK * 2);
    dma_piece(ga, offA[0], st + wave * 1024);
    dma_piece(ga, offA[1], st + (wave + 8) * 1024);
    if (wave < 4) dma_piece(ga, offA[2], st + (wave + 16) * 1024);
    dma_piece(gb, offB[0], st + V_A + (2 * wave) * 1024);
    dma_piece(gb, offB[1], st + V_A + (2 * wave + 1) * 1024);
  };
  KMB_STAMP(0);
  KMB_STAMP_ID();
#ifdef KMB_GEMM_STAMP
  const uint64_t kmb_c0 = __builtin_amdgcn_s_memtime(), kmb_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  dma_stage(0);
  if (ns > 1) dma_stage(1);
  if (ns > 2) dma_stage(2);
  const int frag_off = r * 64 + ((g ^ ((0 - (r >> 2)) & 3)) << 4);
  for (int s = 0; s < ns; ++s) {
    // the oldest stage in flight has landed: all but the (up to two) newer ones' pieces -- five per stage for waves 0-3, four for 4-7
    const int newer = ns - 1 - s;
    if (wave < 4) {
      if (newer >= 2) __builtin_amdgcn_s_waitcnt(0x0F7A);        // vmcnt(10)
      else if (newer == 1) __builtin_amdgcn_s_waitcnt(0x0F75);   // vmcnt(5)
      else __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0)
    } else {
      if (newer >= 2) __builtin_amdgcn_s_waitcnt(0x0F78);        // vmcnt(8)
      else if (newer == 1) __builtin_amdgcn_s_waitcnt(0x0F74);   // vmcnt(4)
      else __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    // stage s is complete, and every wave is done reading stage s - 1.  NOT __syncthreads(): its fence is a vmcnt(0) -- the rounds 4-5
    // form of this loop had one, so its counted vmcnt(9) never held and every step drained the stage it had just issued
    __builtin_amdgcn_s_barrier();
    if (s == 0) KMB_STAMP(1);
    if (s == ns / 2) KMB_STAMP(3);
    // The stage three ahead goes into stage s - 1's buffer, one piece per eight MFMAs (a piece holds the issuing wave for 60-180 cycles;
    // as a burst behind the barrier both waves of a SIMD stall at once).  All of the stage's fragment reads first, then its MFMAs.
    // Measured (tools/allrows_stamps.py, profiles/r06_allrows_kernel_stamps.txt): K loop 31.5 us (rounds 4-5) -> 28 us at 1.96 GHz; the
    // same loop without its DMA 24.6 us, without DMA and fragment reads (1920 MFMAs per SIMD, 24 barriers) 24.2 us: it runs at the rate of
    // its MFMAs and barriers, 1.3 PFLOP/s chip-equivalent -- the rate of this library's other GEMM loops.  Two other forms measured the
    // same 27-29 us: reads one row tile ahead of their MFMAs, and the barrier in the middle of a stage's MFMAs with the next stage's
    // first fragments prefetched behind it (247 registers).
    const bool more = s + 3 < ns;
    char* nst = smem + ((s + 3) & (V_NSTG - 1)) * V_STG;
    const char* ga = uniform_ptr(gA + (size_t)(s + 3) * VK * 2);
    const char* gb = uniform_ptr(gB + (size_t)(s + 3) * VK * 2);
    const char* cur = smem + (s & (V_NSTG - 1)) * V_STG + frag_off;
    bf16x8 fb[4], fa[10];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(cur + V_A + (wn * 4 + j) * 1024);
#pragma unroll
    for (int i = 0; i < 10; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(cur + (wm * 10 + i) * 1024);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // C^T tile
      if (more && (i & 1)) {
        if (i == 1) dma_piece(gb, offB[0], nst + V_A + (2 * wave) * 1024);
        if (i == 3) dma_piece(ga, offA[0], nst + wave * 1024);
        if (i == 5) dma_piece(gb, offB[1], nst + V_A + (2 * wave + 1) * 1024);
        if (i == 7) dma_piece(ga, offA[1], nst + (wave + 8) * 1024);
        if (i == 9 && wave < 4) dma_piece(ga, offA[2], nst + (wave + 16) * 1024);
      }
    }
  }
  __syncthreads();   // (the statistics epilogue reuses the stages)
  KMB_STAMP(2);
#ifdef KMB_GEMM_STAMP
  KMB_STAMP_VALUE(5, __builtin_amdgcn_s_memtime() - kmb_c0);        // shader-clock ticks over the K loop ...
  KMB_STAMP_VALUE(6, __builtin_amdgcn_s_memrealtime() - kmb_r0);    // ... and 100 MHz ticks: the clock the loop ran at
#endif
  // transposed accumulators: lane (r, g) holds C[16 i + r][16 j + 4 g .. + 3] -> one 16-byte store per tile and lane
  if constexpr (!STATS) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + wn * 64 + j * 16 + g * 4;
      if (col >= p.N) continue;   // N % 4 == 0 (launcher): a group of four columns is inside or outside
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      if (p.bias != nullptr) b4 = *reinterpret_cast<const f32x4*>(p.bias + col);
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int row = wm * 160 + i * 16 + r;
        if (row < p.M) *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)row * p.ld_out_f32 + col) = acc[i][j] + b4;
      }
    }
  } else {
    // STATS (round 6, the beam step's first stage folded in): besides the logits, every row's (maximum, sum of exp(v - maximum)) over
    // this workgroup's 256 columns -> stats[(row * blocks + block) * 2 + 0 / 1].  The step's selection kernel (loss.hip
    // beam_stats_merge_kernel) gets the row's log-sum-exp from the 197 pairs and reads only the blocks whose maximum can hold one of
    // the k best, instead of streaming the 64 MB of logits a second time.  The exps run between the stores (the epilogue is bound by
    // the store issue rate, the vector ALUs idle).  Columns >= N count as -inf.
    float* smax = reinterpret_cast<float*>(smem);   // [VR][4]: per (row, wave column) -- the stages are free behind the K loop's last barrier
    float* ssum = smax + VR * 4;                    // [VR][4]
    f32x4 b4[4];
    bool inside[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + wn * 64 + j * 16 + g * 4;
      inside[j] = col < p.N;
      b4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias != nullptr && inside[j]) b4[j] = *reinterpret_cast<const f32x4*>(p.bias + col);
    }
    const bool ragged = col0 + VN > p.N;   // the last block only
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      float lm = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = acc[i][j] + b4[j];
        if (ragged && !inside[j]) acc[i][j] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        lm = fmaxf(fmaxf(lm, fmaxf(acc[i][j][0], acc[i][j][1])), fmaxf(acc[i][j][2], acc[i][j][3]));
      }
      lm = fmaxf(lm, __shfl_xor(lm, 16, 64));
      lm = fmaxf(lm, __shfl_xor(lm, 32, 64));
      if (g == 0) smax[(wm * 160 + i * 16 + r) * 4 + wn] = lm;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int row = wm * 160 + i * 16 + r;
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(smax + row * 4);
      const float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));   // finite: every block has a column < N
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (row < p.M && inside[j])
          *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)row * p.ld_out_f32 + col0 + wn * 64 + j * 16 + g * 4) = acc[i][j];
#pragma unroll
        for (int q = 0; q < 4; ++q) sum += __expf(acc[i][j][q] - m);
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      if (g == 0) ssum[row * 4 + wn] = sum;
    }
    __syncthreads();
    if (tid < p.M) {   // M <= VR <= 512 threads
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(smax + tid * 4);
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(ssum + tid * 4);
      // [row][block] pairs: the selection kernel reads a row's pairs with consecutive lanes (this side: 320 eight-byte stores)
      float2 ms;
      ms.x = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      ms.y = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      reinterpret_cast<float2*>(stats)[(size_t)tid * gridDim.x + blockIdx.x] = ms;
    }
  }
  KMB_STAMP(4);
}

#endif  // KMB_GEMM_DEVICE_ONLY

// ------------------------------------------------------------------------------------------
// v11: persistent 256x256 tile.  One workgroup per CU (grid = 256), four waves (2x2), each wave a 128x128 block of C
// (8x8 MFMA tiles, 256 accumulator registers -- the whole AGPR file; one wave per SIMD).  Why:
//   * LDS bandwidth.  v8's 128x64 wave block reads 24 fragments per 64 MFMAs: 192 KB of ds_reads + 64 KB of LDS-DMA
//     writes per K step against 128 B/clk is as long as the step's MFMA time.  A 128x128 block reads 16 fragments
//     per 64 MFMAs (128 KB per step).
//   * the per-tile fixed cost.  A workgroup walks its tiles as ONE linear sequence of K steps: the DMA cursor runs two
//     steps ahead of the MFMAs straight across tile boundaries, so only the first tile pays a cold prologue, and the
//     epilogue goes through a wave-private 8 KB staging image (no workgroup barrier, the two pipeline stages stay
//     free for the next tile's operands that are landing meanwhile).
// Tiles are dealt per XCD in contiguous ranges (neighbours share A/B panels in that XCD's L2).  Same LDS images,
// swizzles and per-element accumulation order as v7/v8: bit-identical sums.  No split-K (the weight gradients stay
// on v7/v8).
// Measured (tools/gemm_stamps.py, 16384 x 3072 x 768, three tiles per workgroup): prologue 2.7 us once; K loop
// 1.62 us per 64-deep step with all 16 LDS-DMA pieces of a stage issued behind the barrier (the same 1.33 PFLOP/s as
// v8: the LDS-bandwidth argument did not move it) and 1.51 us = 1.43 PFLOP/s with the stage issued in two halves a
// sub-phase apart -- an LDS-DMA piece costs the issuing wave 60-180 cycles, which one wave per SIMD cannot hide behind
// 16-cycle MFMAs; epilogue 3.75 us per tile = 128 KB at ~14 B/clk/CU -- the CU's store-issue rate for 16-byte stores,
// unchanged by hiding the LDS round trip (software-pipelined staging), by non-temporal stores or by starting the
// workgroups of an XCD up to 12 us apart (no HBM burst limit at this size).  Whole kernel 76-77 us against 90 us for
// v8; the vendor library's stream-K kernel (256 workgroups x 256 threads, 256x256x64, tools/gemm_yardstick.py) takes
// 70.5 us per call in a back-to-back loop on this shape.
template <bool KC, int NP = 8>
__device__ __forceinline__ void dma_offsets256w4(uint32_t (&off)[NP], int ld, int r0, int R, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int piece = wave * NP + i;  // 4 * NP pieces of 1 KiB per tile (32 for 256 rows, 24 for a 192-row KC image)
    if (KC) {
      const int row = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      int grow = r0 + row;
      grow = grow < R ? grow : R - 1;
      off[i] = (uint32_t)(((grow - r0) * ld + c * 8) * 2);
    } else {
      const int krow = piece * 2 + (lane >> 5);
      const int ps = lane & 31;
      const int c32 = (ps >> 1) ^ swz_nkc(krow);
      int m = r0 + c32 * 16 + (ps & 1) * 8;
      const int mlast = ((R - 1) >> 3) << 3;
      m = m < R ? m : mlast;
      off[i] = (uint32_t)((krow * ld + (m - r0)) * 2);
    }
  }
}

constexpr int EPW_BYTES = 16 * 128 * 4;            // wave-private fp32 staging: 16 rows x 128 columns, XOR-swizzled
constexpr int LDS11 = 2 * ST4 + 4 * EPW_BYTES;     // 160 KB: the whole CU
constexpr int LDS12 = 2 * (BM4 + 128) * BK * 2 + 4 * EPW_BYTES;   // 256x128 tiles: 128 KB

template <int ACT, bool RES, bool CS, bool FAST, int WROWS, int NJ = 8>
__device__ __forceinline__ void v11_epilogue(const KmbGemm& p, f32x4 (&acc)[8][NJ], float* ef, int lane, int r, int g,
                                             int row0w, int col0w) {
  constexpr int WCOLS = NJ * 16;   // columns of this wave's block (128, 96 for the 256x192 tile, 64 for the 8-wave kernel)
  constexpr int LDE = WCOLS > 64 ? 128 : 64;   // row stride of the staging image (floats)
  float csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
  // transposed accumulators (see the MFMA operand order in the K loop): lane (r, g) holds C[16 i + r][16 j + 4 g .. +3],
  // one ds_write_b128 per MFMA tile; 16-byte groups XOR-swizzled by the row so that the 16 row-lanes spread over banks
  float* const wbase = ef + r * LDE;
  const int sw = (r & 7) << 3;
  auto stage = [&](const f32x4 (&a)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(wbase + ((j * 16 + g * 4) ^ sw)) = a[j];
    asm volatile("" ::: "memory");   // keep the stores inside their switch arm (no select tree over the accumulators)
  };
#pragma unroll 1
  for (int i = 0; i < WROWS / 16; ++i) {
    switch (i) {   // static accumulator indices in every arm (a run-time index would put acc in scratch)
      case 0: stage(acc[0]); break;
      case 1: stage(acc[1]); break;
      case 2: stage(acc[2]); break;
      case 3: stage(acc[3]); break;
      case 4: stage(acc[4]); break;
      case 5: stage(acc[5]); break;
      case 6: stage(acc[6]); break;
      default: stage(acc[7]); break;
    }
    gemm_epilogue_body<64, (WCOLS > 64), 4, LDE, true, 128, ACT, RES, CS, FAST, true, WCOLS>(p, ef, nullptr, lane, row0w + i * 16,
                                                                                    col0w, csum);
  }
  if (CS && (ACT >= 0 || p.colsum != nullptr)) {
    // column sums over this wave's 128 rows: fold the four row-lanes, one partial row per 64 rows of C (first filled,
    // second zeroed)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      csum[e] += __shfl_xor(csum[e], 16);
      csum[e] += __shfl_xor(csum[e], 32);
    }
    if (lane < WCOLS / 8) {
      const int prow = row0w >> 6;
      const int c8 = lane * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (FAST || col0w + c8 + e < p.N) {
          p.colsum[(size_t)prow * p.N + col0w + c8 + e] = csum[e];
          if (WROWS == 128 && (FAST || row0w + 64 < p.M)) p.colsum[(size_t)(prow + 1) * p.N + col0w + c8 + e] = 0.f;
        }
      }
    }
  }
}

// The hot epilogue classes of v11, everything decided at compile time.  With ONE wave per SIMD nothing hides an
// instruction: the general body (run-time option tests, edge handling, spilled-SGPR reloads) costs ~450 instructions
// per 16-row chunk = 7 us per 256x256 tile (in-kernel stamps), as long as the tile's MFMAs at K = 256.  Interior wave
// blocks with a bf16 output only; same operation order as gemm_epilogue_body (bit-identical results).
//   BIAS: + bias[col];  SCALE: * col_scale (the whole wave block lies in the scaled columns);  ACT 1: GeLU (+ optional
//   pre-activation store), 2: * GeLU'(aux);  DROP: dropout mask;  RES: + residual;  CS: column sums.
template <bool BIAS, bool SCALE, int ACT, bool RES, bool DROP, bool CS, int WROWS, bool F32 = false, int NJ = 8>
__device__ __forceinline__ void v11_epilogue_lean(const KmbGemm& p, f32x4 (&acc)[8][NJ], float* ef, int lane, int r, int g,
                                                  int row0w, int col0w) {
  constexpr int WCOLS = NJ * 16;
  // lane map of the row-major pass: CL column-lanes of 8 columns x RPI rows per iteration, NIT iterations per 16-row chunk
  // (128- and 96-column blocks: 16 x 4, four iterations; the 8-wave kernel's 64-column blocks: 8 x 8, two iterations)
  constexpr int CL = WCOLS > 64 ? 16 : 8, RPI = 64 / CL, NIT = 16 / RPI, LDE = WCOLS > 64 ? 128 : 64;
  const int lr = lane / CL;
  // a 96-column wave block (256x192 tile) keeps the 128-column lane map: the last four column-lanes of every row group
  // redo column-lane 11's work on the same addresses (same values: harmless duplicate stores) instead of branching
  // a 96- / 48-column wave block keeps the 128- / 64-column lane map: the column-lanes past the block redo the last
  // column-lane's work on the same addresses (same values: harmless duplicate stores) instead of branching
  const int c8 = ((lane % CL) * 8 < WCOLS) ? (lane % CL) * 8 : WCOLS - 8;
  const int gcol = col0w + c8;
  kmb_f32x2 bias2[4], csum2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bias2[e] = BIAS ? kmb_f32x2{p.bias[gcol + 2 * e], p.bias[gcol + 2 * e + 1]} : kmb_f32x2{0.f, 0.f};
    csum2[e] = kmb_f32x2{0.f, 0.f};
  }
  const kmb_f32x2 scale2 = {p.col_scale, p.col_scale};
  const kmb_f32x2 dscale2 = {p.drop_scale, p.drop_scale};
  // staging write (transposed accumulators: lane (r, g) holds C[16 i + r][16 j + 4 g .. +3]) and read addresses
  // Swizzle of a staged row: its 16-byte groups XORed with the row's low three bits.  Writes: the 8 lanes of a ds_write_b128
  // group (8 rows, one column group) spread over all 32 banks.  Reads: a lane of the row-major pass reads its eight floats as
  // two ds_read_b128, and the 16 lanes of a read group (two rows of different parity) then cover all 64 banks.  (Rounds 1-3
  // XORed at 32-byte granularity: every read asked for the even 16-byte groups only and the writes for every other one --
  // two-way conflicts on both, 15-25 % of the LDS cycles of the forward kernels: tools/pmc_stalls.sh.)
  float* const wbase = ef + r * LDE;
  const int sw = (r & 7) << 2;
  auto stage = [&](const f32x4 (&a)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(wbase + ((j * 16 + g * 4) ^ sw)) = a[j];
    asm volatile("" ::: "memory");
  };
  const float* rd[NIT];   // this lane's floats 0-3 of row-iteration it; floats 4-7 are rd_hi floats further (RPI is even: the row parity is lr's)
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    rd[it] = ef + (lr + RPI * it) * LDE + (c8 ^ (((lr + RPI * it) & 7) << 2));
  const int rd_hi = (lr & 1) ? -4 : 4;
  // row pointers of this lane's first row; a row-iteration is 4 rows further, a chunk 16
  bf16_t* out = F32 ? nullptr : p.out_bf16 + (size_t)(row0w + lr) * p.ld_out_bf16 + gcol;
  float* out32 = F32 ? p.out_f32 + (size_t)(row0w + lr) * p.ld_out_f32 + gcol : nullptr;   // fp32 logits (ld % 4 == 0)
  bf16_t* pre = (ACT == 1 && p.preact != nullptr) ? p.preact + (size_t)(row0w + lr) * p.ld_preact + gcol : nullptr;
  const bf16_t* side = nullptr;   // residual (RES) or GeLU' argument (ACT 2): one 16-byte load per row
  size_t ld_side = 0;
  if (RES) { side = p.residual + (size_t)(row0w + lr) * p.ld_res + gcol; ld_side = (size_t)p.ld_res; }
  if (ACT == 2) { side = p.aux + (size_t)(row0w + lr) * p.ld_aux + gcol; ld_side = (size_t)p.ld_aux; }
  constexpr bool SIDE = RES || ACT == 2;
  static_assert(!(RES && ACT == 2), "one side stream");
  // Side loads run two chunks ahead of their use (a chunk is ~0.3 us, an HBM miss longer).  The chunks are walked in pairs:
  // even chunks keep their side values in sE / hE, odd ones in sO / hO; a chunk first consumes its registers (unpacks them)
  // and then requests chunk i + 2 into the SAME registers, so the loop-carried value is defined by the load itself.  (Rounds
  // 1-3 rotated three register sets, s0 <- s1 <- s2, at the bottom of a one-chunk loop: the copy s1 <- s2 is a USE of the load
  // issued in that same iteration, so hipcc put `s_waitcnt vmcnt(0)` into every chunk -- the prefetch distance was zero and
  // each chunk also waited for its own stores: 10 us of the fc2 data-gradient tile's epilogue, tools/epilogue_burst.py.)
  // The eight-wave kernels (NIT = 2: 128 registers in all; with the pair loop they spilled, and a scratch access in the K
  // loop breaks its counted vmcnt waits) use ONE register set and a one-chunk distance: consume, request chunk i + 1, compute
  // chunk i -- their second wave per SIMD covers the rest.
  constexpr bool PAIRS = NIT == 4;
  constexpr int AHEAD = PAIRS ? 2 : 1;
  u32x4 sE[NIT];
  [[maybe_unused]] u32x4 sO[NIT];
  if (SIDE) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      sE[it] = *reinterpret_cast<const u32x4*>(side + (size_t)(RPI * it) * ld_side);
      if constexpr (PAIRS) sO[it] = *reinterpret_cast<const u32x4*>(side + (size_t)(16 + RPI * it) * ld_side);
    }
  }
  // act 5: the rows' shifts, loaded like the side operand two chunks ahead of their use (a load at the point of use exposed
  // its latency in every row-iteration: the head's forward GEMM 2.24 -> 3.19 ms)
  float hE[NIT];
  [[maybe_unused]] float hO[NIT];
  const float* shift_base = ACT == 5 ? p.row_shift + row0w + lr : nullptr;
  if constexpr (ACT == 5) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      hE[it] = shift_base[RPI * it];
      if constexpr (PAIRS) hO[it] = shift_base[16 + RPI * it];
    }
  }
  auto stage_chunk = [&](int i) {
    switch (i) {   // static accumulator indices in every arm (a run-time index would put acc in scratch)
      case 0: stage(acc[0]); break;
      case 1: stage(acc[1]); break;
      case 2: stage(acc[2]); break;
      case 3: stage(acc[3]); break;
      case 4: stage(acc[4]); break;
      case 5: stage(acc[5]); break;
      case 6: stage(acc[6]); break;
      default: stage(acc[7]); break;
    }
  };
  static_assert((WROWS / 16) % 2 == 0, "chunks are walked in pairs");
  auto chunk = [&](const int i, u32x4 (&sv)[NIT], float (&hv)[NIT]) {
    // this chunk's side values out of their registers, then chunk i + 2's requested into them (the last two chunks re-read
    // rows that are in cache; never used)
    [[maybe_unused]] float su[SIDE ? NIT : 1][8];
    [[maybe_unused]] float hc[ACT == 5 ? NIT : 1];
    const int ahead = i + AHEAD < WROWS / 16 ? i + AHEAD : WROWS / 16 - 1;
    if (SIDE) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) unpack8(sv[it], su[it]);
      __builtin_amdgcn_sched_barrier(0);   // the unpacks stay in front of the reload of their source registers
#pragma unroll
      for (int it = 0; it < NIT; ++it) sv[it] = *reinterpret_cast<const u32x4*>(side + (size_t)(16 * ahead + RPI * it) * ld_side);
    }
    if constexpr (ACT == 5) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) hc[it] = hv[it];
#pragma unroll
      for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(hc[it]));   // a value of its own, not an alias of the register being reloaded
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int it = 0; it < NIT; ++it) hv[it] = shift_base[16 * ahead + RPI * it];
    }
    // this chunk's rows out of LDS first, then the next chunk's accumulators into the same image: the LDS executes
    // a wave's accesses in order, so the writes queue behind the reads and their latency hides under this chunk's math
    f32x4 lo4[NIT], hi4[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      lo4[it] = *reinterpret_cast<const f32x4*>(rd[it]);
      hi4[it] = *reinterpret_cast<const f32x4*>(rd[it] + rd_hi);
    }
    asm volatile("" ::: "memory");
    if (i + 1 < WROWS / 16) stage_chunk(i + 1);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const f32x4 lo = lo4[it];
      const f32x4 hi = hi4[it];
      kmb_f32x2 v[4] = {{lo[0], lo[1]}, {lo[2], lo[3]}, {hi[0], hi[1]}, {hi[2], hi[3]}};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (BIAS) v[e] = v[e] + bias2[e];
        if (SCALE) v[e] = v[e] * scale2;
      }
      const size_t roff = (size_t)(16 * i + RPI * it);
      if constexpr (ACT == 5) {
        // tied-head cross-entropy: exp(v - shift[row]) is what is stored; the row's fp32 sum over this wave block and the
        // shifted value at the label's column go to the side buffers (see KmbGemm)
        static_assert(ACT != 5 || WCOLS == 64 || WCOLS == 128, "act 5: 64- or 128-column wave blocks");
        const int grow = row0w + lr + 16 * i + RPI * it;
        const float c = hc[it];
        const kmb_f32x2 c2 = {c, c};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] - c2;
        if (p.pick_col != nullptr) {   // uniform; optional (the engine does without: the shift IS the label's logit)
          const int rel = (int)((long long)p.pick_col[grow] - (long long)gcol);   // the label's column relative to this lane's eight
          if (rel >= 0 && rel < 8) {
            float picked = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (rel == 2 * e) picked = v[e][0];
              if (rel == 2 * e + 1) picked = v[e][1];
            }
            p.pick_out[grow] = picked;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // 2^115 bounds a stored value and a 50k-column row sum inside fp32 / bf16: a logit more than 80 above the label's (a
          // row whose loss exceeds 80 nats) saturates instead of turning the row's sum, loss and gradients into inf / NaN
          const kmb_f32x2 t = v[e] * 1.4426950408889634f;
          v[e] = kmb_f32x2{__builtin_amdgcn_exp2f(fminf(t[0], 115.f)), __builtin_amdgcn_exp2f(fminf(t[1], 115.f))};
        }
        float sum = (v[0][0] + v[0][1]) + (v[1][0] + v[1][1]) + ((v[2][0] + v[2][1]) + (v[3][0] + v[3][1]));
#pragma unroll
        for (int o = 1; o < CL; o <<= 1) sum += __shfl_xor(sum, o);   // the CL column-lanes of a row are consecutive lanes
        if ((lane % CL) == 0) {
          float* slot = p.row_sums + (size_t)grow * p.row_sums_ld + (col0w >> 6);
          slot[0] = sum;
          if (WCOLS == 128) slot[1] = 0.f;
        }
      } else if (ACT == 1) {
        if (pre != nullptr) {   // GeLU and GeLU' from one evaluation; the derivative is stored for backward (ACT 2)
          kmb_f32x2 dv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            kmb_f32x2 y;
            gelu_both2(v[e], y, dv[e]);
            v[e] = y;
          }
          const u32x4 pk = {pack2bf(dv[0][0], dv[0][1]), pack2bf(dv[1][0], dv[1][1]), pack2bf(dv[2][0], dv[2][1]), pack2bf(dv[3][0], dv[3][1])};
          KMB_NT_STORE(pk, reinterpret_cast<u32x4*>(pre + roff * p.ld_preact));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu2(v[e]);
        }
      } else if (ACT == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * kmb_f32x2{su[it][2 * e], su[it][2 * e + 1]};
      }
      if (DROP) {
        const uint32_t grow = (uint32_t)(row0w + lr + 16 * i + RPI * it);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const kmb_f32x2 kept = v[e] * dscale2;
          v[e][0] = drop_keep(p.drop_seed, grow, (uint32_t)(gcol + 2 * e), p.drop_thr16) ? kept[0] : 0.f;
          v[e][1] = drop_keep(p.drop_seed, grow, (uint32_t)(gcol + 2 * e + 1), p.drop_thr16) ? kept[1] : 0.f;
        }
      }
      if (RES) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] + kmb_f32x2{su[it][2 * e], su[it][2 * e + 1]};
      }
      if (CS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) csum2[e] = csum2[e] + v[e];
      }
      if (F32) {
        float* o = out32 + roff * p.ld_out_f32;
        KMB_NT_STORE((f32x4{v[0][0], v[0][1], v[1][0], v[1][1]}), reinterpret_cast<f32x4*>(o));
        KMB_NT_STORE((f32x4{v[2][0], v[2][1], v[3][0], v[3][1]}), reinterpret_cast<f32x4*>(o + 4));
      } else {
        const u32x4 pk = {pack2bf(v[0][0], v[0][1]), pack2bf(v[1][0], v[1][1]), pack2bf(v[2][0], v[2][1]), pack2bf(v[3][0], v[3][1])};
#ifdef KMB_PLAIN_FFN_OUT   // experiment build: the FFN's wide activations (GeLU output, its gradient) with default-policy stores
        if (ACT == 1 || ACT == 2) *reinterpret_cast<u32x4*>(out + roff * p.ld_out_bf16) = pk;
        else KMB_NT_STORE(pk, reinterpret_cast<u32x4*>(out + roff * p.ld_out_bf16));
#else
        KMB_NT_STORE(pk, reinterpret_cast<u32x4*>(out + roff * p.ld_out_bf16));
#endif
      }
    }
  };
  stage_chunk(0);
  if constexpr (PAIRS) {
#pragma unroll 1
    for (int i = 0; i < WROWS / 16; i += 2) {
      chunk(i, sE, hE);
      chunk(i + 1, sO, hO);
    }
  } else {
#pragma unroll 1
    for (int i = 0; i < WROWS / 16; ++i) chunk(i, sE, hE);
  }
  if (CS) {
    float csum[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { csum[2 * e] = csum2[e][0]; csum[2 * e + 1] = csum2[e][1]; }
#pragma unroll
    for (int e = 0; e < 8; ++e) {   // fold the row-lanes
      if (CL == 8) csum[e] += __shfl_xor(csum[e], 8);
      csum[e] += __shfl_xor(csum[e], 16);
      csum[e] += __shfl_xor(csum[e], 32);
    }
    if (lane < CL) {
      const int prow = row0w >> 6;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        p.colsum[(size_t)prow * p.N + gcol + e] = csum[e];
        if (WROWS == 128) p.colsum[(size_t)(prow + 1) * p.N + gcol + e] = 0.f;
      }
    }
  }
}

// BNT = 256: waves 2x2, each 128x128.  BNT = 128 (variant 12): waves 4x1, each 64x128 -- twice as many tiles, for shapes
// whose 256x256 tile count is not a multiple of the 256 workgroups (N = 768: 1.5 tiles per workgroup -> 3).
// NW = 8 (variant 14, BNT = 256 only): the same persistent structure with EIGHT waves, 2 x 4 of 128 x 64 (128 accumulator
// registers each, two waves per SIMD).  One wave alone issues a vector instruction every 4 cycles, two waves sharing a
// SIMD one every 2 (MI355X_MICROARCH.md, cycle constants): with one wave per SIMD a tile's epilogue is pure single-wave
// VALU / store issue time with the matrix pipe idle -- the GeLU + GeLU' epilogue of a K = 768 tile costs 15.3 us against
// an 18.8 us K loop -- and the second wave halves exactly that part.  The K loop is v8's (same fragments, same
// accumulation order: bit-identical), MFMA-paced either way.
template <bool A_KC, bool B_KC, int BNT, int NW = 4>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))   // (round 6: no amdgpu_num_vgpr(255) any more -- it capped the accumulator file at 252 registers, so two of the four-wave 256 x 256 kernels' 64 accumulator tiles lived in VGPRs and were swapped through a[0:3] / a[232:235] behind s_nop 5 stalls: 32-56 v_accvgpr moves per K step)
void gemm_kernel_v11(const KmbGemm p, uint32_t* sched, int dyn_first) {
  static_assert(NW == 4 || (NW == 8 && (BNT == 256 || BNT == 192)), "eight waves: 256 x 256 and 256 x 192 tiles only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  KMB_STAMP(0);
  KMB_STAMP_ID();
  [[maybe_unused]] uint64_t kmb_loop_ticks = 0, kmb_epi_ticks = 0, kmb_wait_ticks = 0, kmb_drain_ticks = 0;  // diagnostic build
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // BNT = 256: waves 2x2 of 128x128;  128: waves 4x1 of 64x128;  192: waves 2x2 of 128x96 (N = 768 = 4 x 192: 256 / 512
  // tiles at M = 16384 / 32768, one / two per CU, where 256-wide tiles leave a quarter / half round idle)
  constexpr int WN = NW == 8 ? 4 : BNT == 128 ? 1 : 2, WM = NW / WN, WROWS = BM4 / WM, MH = WROWS / 64;   // 2,2,128,2  or  1,4,64,1  or (NW 8) 4,2,128,2
  constexpr int WCOLS = BNT / WN, NJ = WCOLS / 16;     // columns / MFMA tile columns of a wave block: 128 / 8 or 96 / 6
  // B image: a non-KC 192-column tile keeps the 256-column row stride (the XOR swizzle of the transposing reads
  // permutes 32-byte chunks within groups of eight: 12 chunks do not close under it); its pieces cover the full stride
  // and the lanes past column 191 fetch bytes nobody reads
  constexpr int BIMG = (BNT == 192 && !B_KC) ? 256 : BNT;
  constexpr int NPA = 32 / NW;                         // 1 KiB LDS-DMA pieces of A per wave and stage
  constexpr int NPB = BIMG / 32 * 4 / NW;              // ... of B
  constexpr int STG = (BM4 + BIMG) * BK * 2;           // one pipeline stage
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 15, g = lane >> 4;
  const int tiles_n = (p.N + BNT - 1) / BNT;
  const int tiles_m = (p.M + BM4 - 1) / BM4;
  const int ntiles = tiles_m * tiles_n;
  // Tile enumeration.  Row-major (index = tm * tiles_n + tn) by default.  tile_order bit 3: COLUMN-BLOCK-major for wide
  // outputs (the tied LM head: 197 column tiles of a 77 MB matrix) -- blocks of CB column tiles, all row panels of a
  // block before the next block, so that the 32 workgroups of an XCD (contiguous index range) work on 4 row panels x 8
  // column tiles whose 8 weight panels (3 MB) stay in that XCD's L2 while the row panels stream past once per block;
  // row-major, every row panel pulled the whole weight matrix through the L2 again (rocprofv3 FETCH_SIZE of the LM
  // head: 10.5 GB for 128 MB of operands).  The XCDs' contiguous ranges then partition the COLUMNS of the weight.
  constexpr int CB = 8;
  const bool col_blocks = (p.tile_order & 8) != 0 && tiles_n > CB;
  const int cb_full = tiles_n / CB;                          // full blocks; a last block holds the remaining columns
  auto decode_tile = [&](int t, int& tm, int& tn) {
    if (!col_blocks) { tm = t / tiles_n; tn = t - tm * tiles_n; return; }
    const int blk = t / (CB * tiles_m);
    if (blk < cb_full) {
      const int rem = t - blk * (CB * tiles_m);
      tm = rem / CB; tn = blk * CB + (rem - tm * CB);
    } else {
      const int wl = tiles_n - cb_full * CB;               // > 0 here
      const int rem = t - cb_full * (CB * tiles_m);
      tm = rem / wl; tn = cb_full * CB + (rem - tm * wl);
    }
  };
  const int pf_sharers = col_blocks ? CB : tiles_n;         // workgroups that walk a row panel together (L2 prefetch shares)
  // this workgroup's tiles: XCD x owns a contiguous range, its workgroups take every (grid/8)-th tile of it
  // (tile_order bit 0; without it workgroup b simply takes tiles b, b + grid, ...: better when C dominates the traffic)
  const bool xcd_ranges = (p.tile_order & 1) != 0;
  const int per = xcd_ranges ? (int)gridDim.x >> 3 : (int)gridDim.x;
  const int xcd = (int)blockIdx.x & 7, loc = (int)blockIdx.x >> 3;
  const int tq = ntiles >> 3, trem = ntiles & 7;
  const int range0 = !xcd_ranges ? 0 : xcd < trem ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq;
  const int range1 = !xcd_ranges ? ntiles : range0 + tq + (xcd < trem ? 1 : 0);
  const int first = xcd_ranges ? range0 + loc : (int)blockIdx.x;
  // Tiles after the first are handed out by an atomic counter (per XCD range, or one for the whole grid): a workgroup
  // that gets its CU late -- another kernel's workgroups, e.g. an RCCL all-reduce on the communication stream, hold
  // some CUs -- simply takes fewer tiles instead of leaving its fixed share to a second round.  sched[0..7] tile
  // counters, sched[8] finished workgroups; the last one to finish zeroes them for the next launch using this slot.
  const bool dyn = sched != nullptr;
  uint32_t* const my_ctr = sched + (xcd_ranges ? xcd : 0);
  auto retire = [&]() {
    if (dyn && tid == 0) {
      if (atomicAdd(sched + 8, 1u) == gridDim.x - 1u) {
#pragma unroll
        for (int i = 0; i < 9; ++i) sched[i] = 0u;
      }
    }
  };
  int* const next_slot = reinterpret_cast<int*>(smem + 2 * STG);   // word 0 of wave 0's epilogue staging (idle in the K loop)
  // the wave that issues the L2 touches: the LAST one (wave 0 also hands out tiles; the touch costs its wave five scalar instructions per K
  // step since round 5); its landing words: 256 bytes of its own staging image (KMB_L2_TOUCH)
  constexpr int PFW = NW - 1;
  const unsigned touch_lds = kmb_lds_addr(smem + 2 * STG + PFW * (WCOLS > 64 ? EPW_BYTES : EPW_BYTES / 2) + 2048);
  uint32_t fetched = 0u;
  // dyn_first (the GPU is shared with a communication kernel: kmb_gemm_shared_device): the first tile comes from the
  // counter too, so a workgroup that is placed late finds nothing left and exits instead of holding the launch open
  // for its fixed first tile; costs one atomic round trip (~2 us) before the prologue.
  const int dyn_base = (dyn && dyn_first) ? range0 : range0 + per;
  int first_tile = first;
  if (dyn && dyn_first) {
    if (tid == 0) *next_slot = range0 + (int)atomicAdd(my_ctr, 1u);
    __syncthreads();
    first_tile = __builtin_amdgcn_readfirstlane(*next_slot);
    __syncthreads();
  }
  if (first_tile >= range1) { retire(); return; }
#if defined(KMB_GEMM_STAMP) && KMB_STAMP_SLOTS >= 12
  // in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6): shader-clock ticks over 100 MHz real-time ticks
  const uint64_t kmb_c0 = __builtin_amdgcn_s_memtime(), kmb_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef KMB_DIAG
  // (tools/epilogue_burst.py) KMB_GEMM_STAGGER = s: the workgroups start in four groups, s half-microseconds apart, so that
  // their epilogues -- a burst of 64 MB of stores when all 256 workgroups reach them together -- are spread over a tile time
  if (const int stg = (p.tile_order >> 16) & 255) {
    const int q = ((int)blockIdx.x >> 3) & 3;
    for (int i = 0; i < q * stg; ++i) __builtin_amdgcn_s_sleep(18);
  }
#endif
  const int nt = p.K / BK;   // >= 2 (launcher)
  constexpr int A_BYTES = BM4 * BK * 2;
  const size_t stepA = A_KC ? (size_t)BK * 2 : (size_t)BK * p.lda * 2;
  const size_t stepB = B_KC ? (size_t)BK * 2 : (size_t)BK * p.ldb * 2;
  auto uniform_ptr = [](const char* ptr) {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };

  // ---- DMA cursor: (tile, K step) of the next stage to fetch; runs two steps ahead of the MFMAs ----
  uint32_t offA[NPA], offB[NPB];
  const char *gA_d, *gB_d;
  int tile_d = first_tile, td = 0;
  // ---- L2 prefetch of the activation operand (K-contiguous A only) ----
  // Inside a training step A was just streamed out by the previous kernel: every LDS-DMA request of a stage is an HBM
  // miss, and the tiles_n workgroups that share a row panel (same XCD, running in step) all miss on the same lines, each
  // holding a request slot per line for the whole HBM latency (in-kernel stamps: 1.76 instead of 1.12 us per K step).
  // Each of them therefore touches ITS share of the panel's rows (256 / tiles_n rows = one 128-byte line each per K
  // step, ONE load instruction of wave 0) KMB_PFD steps ahead of the DMA cursor: the lines are in the XCD's L2 when the
  // DMAs of all sharers ask for them.  The load's result is never used; it stays outstanding across the stage wait
  // (vmcnt(1) instead of 0 for the touching wave) and must only be complete one step later (see the K loop for why its destination
  // is one register web for the whole kernel).
#ifndef KMB_V11_PREFETCH
#define KMB_V11_PREFETCH 1
#endif
#ifndef KMB_V11_PFD
#define KMB_V11_PFD 2
#endif
  constexpr int KMB_PFD = KMB_V11_PFD;
  constexpr bool PF_ON = KMB_V11_PREFETCH != 0 && A_KC;
  const bool pf_rt = (p.tile_order & 2) != 0;   // set per launch (kmb_gemm_launch): only where A is expected to come from HBM
  // The first KMB_PFD steps of a tile have no earlier step of the same tile to be prefetched from: they are touched
  // from the tile `per` places earlier in the range -- the tile whose workgroup is one round ahead of the one that will
  // take this tile (tiles are handed out in range order, so the workgroups of one round run the sharers of a panel).
  uint32_t pf_off = 0u, pfn_off = 0u;
  const char* pfn_base = reinterpret_cast<const char*>(p.A);
  bool pf_pending = false, pfn_ok = false;
  auto pf_row_offset = [&](int tn, int row0) {
    int share = (BM4 + pf_sharers - 1) / pf_sharers;
    share = share > 64 ? 64 : share;
    int prow = tn * share + (lane < share ? lane : 0);
    prow = prow < BM4 ? prow : BM4 - 1;
    prow = row0 + prow < p.M ? prow : 0;
    return (uint32_t)prow * (uint32_t)p.lda * 2u;
  };
  auto set_dma_tile = [&](int tile) {
    int tm, tn;
    decode_tile(tile, tm, tn);
    const int row0 = tm * BM4, col0 = tn * BNT;
    if (PF_ON) {
      pf_off = pf_row_offset(col_blocks ? tn % CB : tn, row0);
      const int tx = tile + per;
      pfn_ok = tx < range1;
      if (pfn_ok) {
        int tmx, tnx;
        decode_tile(tx, tmx, tnx);
        pfn_off = pf_row_offset(col_blocks ? tnx % CB : tnx, tmx * BM4);
        pfn_base = uniform_ptr(reinterpret_cast<const char*>(p.A) + (size_t)tmx * BM4 * p.lda * 2);
      }
    }
    dma_offsets256w4<A_KC, NPA>(offA, p.lda, row0, p.M, wave, lane);
    if constexpr (BNT == 128) dma_offsets<B_KC>(offB, p.ldb, col0, p.N, wave, lane);
    else dma_offsets256w4<B_KC, NPB>(offB, p.ldb, col0, p.N, wave, lane);
    gA_d = uniform_ptr(reinterpret_cast<const char*>(p.A) + (A_KC ? (size_t)row0 * p.lda * 2 : (size_t)row0 * 2));
    gB_d = uniform_ptr(reinterpret_cast<const char*>(p.B) + (B_KC ? (size_t)col0 * p.ldb * 2 : (size_t)col0 * 2));
  };
  int tile_next = first_tile;   // the tile after the one being multiplied; known once the DMA cursor reaches it
  auto advance_cursor = [&]() {   // before a fetch: step to the next tile when this one's K steps are all issued
    if (td == nt) {
      td = 0;
      tile_next = dyn ? __builtin_amdgcn_readfirstlane(*next_slot) : tile_d + per;
      if (tile_next < range1) tile_d = tile_next;   // past the last tile: fetch its first steps again (never read)
      set_dma_tile(tile_d);
    }
  };
  char* const dstA = smem + wave * (NPA * 1024);
  char* const dstB = smem + A_BYTES + wave * (NPB * 1024);
  // A stage's fetch is issued in two halves, one sub-phase apart (an LDS-DMA piece costs the issuing wave 60-180
  // cycles; sixteen of them in one 32-MFMA sub-phase made that sub-phase as long as the other three together)
  // (128-row wave blocks) where the 8 A and 8 B pieces of a stage go: A pieces [0, NA3) in sub-phase 3, the rest in the
  // next step's sub-phase 0; B pieces [0, NB3) in sub-phase 3, [NB3, NB3 + NB0) in sub-phase 0, the rest in sub-phase 1
#ifndef KMB_V11_SPLIT
#define KMB_V11_SPLIT 1
#endif
  constexpr int NA3 = KMB_V11_SPLIT == 3 ? NPA / 2 : NPA;
  constexpr int NB3 = KMB_V11_SPLIT == 0 ? NPB : 0;
  constexpr int NB0 = KMB_V11_SPLIT == 0 ? 0 : KMB_V11_SPLIT == 2 ? NPB / 2 : NPB;
  constexpr int NB1 = NPB - NB3 - NB0;
  constexpr int P3 = (NA3 + NB3) / 2, P0 = (NPA - NA3 + NB0) / 2, P1 = NB1 / 2;   // pairs of pieces per sub-phase
  auto dma_a = [&](int buf, int lo, int hi) {
    char* da = dstA + buf * STG;
#ifndef KMB_V11_NODMA   // (timing experiment only: the K loop without its LDS-DMA -- results are garbage)
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
#ifdef KMB_A_NT
      if (i >= lo && i < hi) { if constexpr (A_KC) dma_piece_nt(gA_d, offA[i], da + i * 1024); else dma_piece(gA_d, offA[i], da + i * 1024); }
#else
      if (i >= lo && i < hi) dma_piece(gA_d, offA[i], da + i * 1024);
#endif
    }
#endif
    if (hi == NPA && lo < hi) gA_d = uniform_ptr(gA_d + stepA);
  };
  auto dma_b = [&](int buf, int lo, int hi) {
    char* db = dstB + buf * STG;
#ifndef KMB_V11_NODMA
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      if (i >= lo && i < hi) dma_piece(gB_d, offB[i], db + i * 1024);
#endif
    if (hi == NPB && lo < hi) {
      gB_d = uniform_ptr(gB_d + stepB);
      ++td;
    }
  };
  auto dma_stage_a = [&](int buf) { dma_a(buf, 0, NPA); };
  auto dma_stage_b = [&](int buf, int) { dma_b(buf, 0, NPB); };

  constexpr int NDA = A_KC ? 4 : 8;          // ds_read instructions per 4 A fragments
  constexpr int NDB = B_KC ? NJ : 2 * NJ;    // ... per NJ B fragments
  constexpr int NM = 4 * NJ;                 // MFMAs of one sub-phase (4 A fragments x NJ B fragments)
  bf16x8 fa[2][4], fb[2][NJ];
  f32x4 acc[8][NJ];
  // Eight waves: 128 registers hold everything beside the accumulators, and hipcc keeps one address register per
  // distinct fragment address it can hoist out of the K loop (the XOR swizzles are not additive) -- two of them spilled, and
  // a scratch reload inside the loop is a VMEM load whose wait (vmcnt) also drains the LDS-DMA pieces in flight.  So the
  // fragment addresses are rebuilt per use from ONE lane constant per operand (same addresses as read_frag3, checked
  // exhaustively on the host when this was written): with swz the lane's swizzle term,
  //   K-contiguous image:   row * 128 + (((kk * 4 + g) ^ ((r >> 1) & 7)) << 4)        = (cA ^ (kk << 6)) + rowtile * 2048
  //   token-major image:    krow * 512 + ((rowtile ^ swz) << 5) + ((r & 3) << 3)        = (cB ^ (rowtile << 5)) + (kk * 32 + hh * 4) * 512
  // and the constant is laundered through an empty asm in every call so that the XORs are not hoisted again.
  constexpr bool REMAT = NW == 8;
  const int rm_ka = r * 128 + ((g ^ ((r >> 1) & 7)) << 4);
  const int rm_sw = ((r >> 2) & 3) | ((g & 1) << 2);
  const int rm_l = (g * 8 + (r >> 2)) * (BIMG * 2) + ((r & 3) << 3);
  const int rm_nb = rm_l | (rm_sw << 5);                                                   // bits 5-8 of rm_l are zero
  const int rm_na = ((g * 8 + (r >> 2)) * (BM4 * 2) + ((r & 3) << 3)) | (rm_sw << 5);
  // inline-asm transposing reads (kmb_tr_read_asm) in every tile width since round 6 (KMB_TR_ALL)
#ifndef KMB_TR_BUILTIN
  constexpr bool TRASM = (BNT == 256 || KMB_TR_ALL) && !(A_KC && B_KC);
#else
  constexpr bool TRASM = false;
#endif
  auto tr_read = [&](const char* ptr) {
#ifndef KMB_TR_BUILTIN
    if constexpr (TRASM) return kmb_tr_read_asm(ptr);
    else
#endif
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
  };
  auto read_a = [&](const char* stage, int kk, int half, bf16x8 (&dst)[4]) {
    if constexpr (REMAT && A_KC) {
      int c = rm_ka;
      asm volatile("" : "+v"(c));
      const char* base = stage + (c ^ (kk << 6)) + (wm * (MH * 4) + half * 4) * 2048;
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(base + i * 2048);
    } else if constexpr (REMAT) {   // token-major A: row tiles wm * 8 + half * 4 + i
      int c = rm_na;
      asm volatile("" : "+v"(c));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* pj = stage + kk * (32 * BM4 * 2) + (c ^ ((wm * (MH * 4) + half * 4 + i) << 5));
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const s16x4 t = tr_read(pj + hh * (4 * BM4 * 2));
          dst[i][hh * 4 + 0] = t[0]; dst[i][hh * 4 + 1] = t[1]; dst[i][hh * 4 + 2] = t[2]; dst[i][hh * 4 + 3] = t[3];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i] = read_frag3<A_KC, BM4, TRASM>(stage, wm * (MH * 4) + half * 4 + i, kk, r, g);
    }
  };
  auto read_b = [&](const char* stage, int kk, bf16x8 (&dst)[NJ]) {
    if constexpr (REMAT && B_KC) {
      int c = rm_ka;
      asm volatile("" : "+v"(c));
      const char* base = stage + A_BYTES + (c ^ (kk << 6)) + (wn * NJ) * 2048;
#pragma unroll
      for (int j = 0; j < NJ; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(base + j * 2048);
    } else if constexpr (REMAT) {
      int c = rm_nb;
      asm volatile("" : "+v"(c));
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const char* pj = stage + A_BYTES + kk * (32 * BIMG * 2) + (c ^ ((wn * NJ + j) << 5));
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const s16x4 t = tr_read(pj + hh * (4 * BIMG * 2));
          dst[j][hh * 4 + 0] = t[0]; dst[j][hh * 4 + 1] = t[1]; dst[j][hh * 4 + 2] = t[2]; dst[j][hh * 4 + 3] = t[3];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) dst[j] = read_frag3<B_KC, BIMG, TRASM>(stage + A_BYTES, wn * NJ + j, kk, r, g);
    }
  };
#ifdef KMB_V11_MFMA32_TIMING
  // TIMING EXPERIMENT ONLY (diagnostic builds): the same fragments fed to 32x32x16 MFMAs -- half as many instructions,
  // 32 cycles each, 8 of them holding the issue port.  Results are meaningless; the K loop's duration is the point.
  f32x16 acc32[4][4];
  static_assert(NJ == 8, "timing experiment: 128-column wave blocks only");
  auto mma = [&](int half, const bf16x8 (&a)[4], const bf16x8 (&b)[8]) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          acc32[half * 2 + ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[jj * 2 + ks], a[ii * 2 + ks], acc32[half * 2 + ii][jj], 0, 0, 0);
  };
#define KMB_MF(n) ((n) / 2)
#else
  auto mma = [&](int half, const bf16x8 (&a)[4], const bf16x8 (&b)[NJ]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[half * 4 + i][j], 0, 0, 0);  // C^T tile
  };
#define KMB_MF(n) (n)
#endif

  set_dma_tile(tile_d);
  if (PF_ON && pf_rt && wave == PFW) {   // steps 2 .. PFD + 1 of the first tile (issued before, so complete before, the stage pieces)
#pragma unroll
    for (int i = 2; i < 2 + KMB_PFD; ++i) {
      if (i < nt) {
        const char* pbase = uniform_ptr(gA_d + (size_t)i * stepA);
        KMB_L2_TOUCH(pf_off, pbase, touch_lds);   // one wave only: a word of its own staging image
      }
    }
  }
  dma_stage_a(0);
  dma_stage_b(0, 2);
  advance_cursor();
  // (128-row wave blocks) stage 1's sub-phase-3 share now, the rest in the first K step's sub-phases 0 and 1
  if constexpr (MH == 1) {
    dma_stage_a(1);
    dma_stage_b(1, 2);
    __builtin_amdgcn_s_waitcnt(0x0F7C);   // vmcnt(12) = the pieces of stage 1: stage 0 has landed
  } else {
    dma_a(1, 0, NA3);
    dma_b(1, 0, NB3);
    __builtin_amdgcn_s_waitcnt(NA3 + NB3 == 16 ? 0x4F70 : NA3 + NB3 == 8 ? 0x0F78 : 0x0F74);   // vmcnt(16 | 8 | 4)
  }
  __syncthreads();
  read_b(smem, 0, fb[0]);
  read_a(smem, 0, 0, fa[0]);
  KMB_STAMP(1);

  float* const ef = reinterpret_cast<float*>(smem + 2 * STG + wave * (WCOLS > 64 ? EPW_BYTES : EPW_BYTES / 2));
  int it = 0;   // linear K-step counter: stage buffer = it & 1
  for (int tile = first_tile; tile < range1; tile = tile_next) {
    if (dyn && tid == 0) fetched = atomicAdd(my_ctr, 1u);   // lands by the first K step's vmcnt(0)
#pragma unroll
    for (int i = 0; i < MH * 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef KMB_V11_MFMA32_TIMING
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
#endif
    [[maybe_unused]] const uint64_t kmb_t_loop = KMB_NOW();
    for (int t = 0; t < nt; ++t, ++it) {
      // publish the next tile to the other waves: written in step 1, behind step 1's barrier when the cursor reads it
      // in step nt - 2 >= 2 (the launcher hands out a counter only when nt >= 4)
      if (dyn && t == 1 && tid == 0) *next_slot = dyn_base + (int)fetched;
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): k0 fragments of this stage (needed now anyway; a known-empty
                                           // LDS queue here lets the compiler count the waits below exactly)
      if constexpr (TRASM) KMB_TR_SYNC();
      const char* cur = smem + (it & 1) * STG;
      const char* nxt = smem + ((it + 1) & 1) * STG;
      if constexpr (MH == 2) {
      // ---- sub-phase 0: A(k0, rows 0-63) x B(k0)  ||  read A(k0, rows 64-127) ----
      read_a(cur, 0, 1, fa[1]);
      dma_a((it + 1) & 1, NA3, NPA);          // the rest of the stage whose first pieces went out in the previous
      dma_b((it + 1) & 1, NB3, NB3 + NB0);    // sub-phase 3
      mma(0, fa[0], fb[0]);
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(4), 0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, NDA / 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(4), 0);
      }
#pragma unroll
      for (int q = 0; q < P0; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);  // VMEM (LDS-DMA)
        __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF((NM - 12) / (P0 > 0 ? P0 : 1)), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- sub-phase 1: A(k0, rows 64-127) x B(k0)  ||  read B(k1), A(k1, rows 0-63) ----
      if constexpr (TRASM) KMB_TR_SYNC();
      read_b(cur, 1, fb[1]);
      read_a(cur, 1, 0, fa[0]);
      dma_b((it + 1) & 1, NB3 + NB0, NPB);
      mma(1, fa[1], fb[0]);
      if constexpr (NM >= 24) {
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(6), 1);   // MFMAs first: their operands were read a sub-phase ago
      __builtin_amdgcn_sched_group_barrier(0x100, NDB / 2, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(5), 1);
      __builtin_amdgcn_sched_group_barrier(0x100, NDB / 2, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(5), 1);
      __builtin_amdgcn_sched_group_barrier(0x100, NDA, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(4), 1);
      } else {   // 16 / 12 MFMAs (64- / 48-column wave blocks of the eight-wave kernels)
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(NM / 4), 1);
      __builtin_amdgcn_sched_group_barrier(0x100, (NDB + 1) / 2, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(NM / 4), 1);
      __builtin_amdgcn_sched_group_barrier(0x100, NDB / 2, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(NM / 4), 1);
      __builtin_amdgcn_sched_group_barrier(0x100, NDA, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(NM / 4), 1);
      }
#pragma unroll
      for (int q = 0; q < P1; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 1);  // VMEM (LDS-DMA)
        __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF((NM - 20) / (P1 > 0 ? P1 : 1)), 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      advance_cursor();
      if (PF_ON) {
        // The L2 touch is the YOUNGEST memory operation at this step's stage wait (all pieces of the stage are out), so
        // that wait is vmcnt(1) for the touching wave (the last one) and the touch -- an HBM miss by design -- has until the NEXT step's wait to
        // land.  (Issued as the oldest operation of the window behind a plain vmcnt(0) it sat in front of the stage's
        // pieces in the in-order return queue and the whole gain was gone: 42.2 vs 39.5 ms of in-step GEMM time.)
        // It has no register destination (KMB_L2_TOUCH: a 4-byte LDS-DMA into a word of wave 0's staging image, idle in the K loop).
        // No touch in a tile's last K step: that step's wait is vmcnt(0), nothing is in flight across the epilogue, which
        // writes that staging image.
        const int ps = td + KMB_PFD;   // td: the step the next fetch of this workgroup asks for
        const bool in_tile = ps < nt;
        pf_pending = pf_rt && wave == PFW && t + 1 < nt && (in_tile || (pfn_ok && ps - nt < nt));
        if (pf_pending) {
          const char* pbase = uniform_ptr(in_tile ? gA_d + (size_t)KMB_PFD * stepA : pfn_base + (size_t)(ps - nt) * stepA);
          const uint32_t poff = in_tile ? pf_off : pfn_off;
          KMB_L2_TOUCH(poff, pbase, touch_lds);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- sub-phase 2: A(k1, rows 0-63) x B(k1)  ||  read A(k1, rows 64-127); stage it+1 landed, barrier ----
      if constexpr (TRASM) KMB_TR_SYNC();
      read_a(cur, 1, 1, fa[1]);
      mma(0, fa[0], fb[1]);
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(8), 2);
      __builtin_amdgcn_sched_group_barrier(0x100, NDA, 2);
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(NM - 8), 2);
      __builtin_amdgcn_sched_barrier(0);
      {
        KMB_WAIT_BEGIN();
        if (PF_ON && pf_pending) __builtin_amdgcn_s_waitcnt(0x0071);  // vmcnt(1): this step's L2 touch stays in flight
        else __builtin_amdgcn_s_waitcnt(0x0070);                       // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        // lgkmcnt(0) once more, in straight-line code (round 6).  It already holds on both paths into the barrier, but hipcc's wait tracking
        // loses it at the join of the branch above and put its own lgkmcnt(0) in front of sub-phase 3's first MFMA -- i.e. BEHIND the eight
        // to twenty fragment reads of the next stage that sub-phase 3 issues first: their whole LDS latency was exposed in every K step of
        // every persistent kernel (ISA: `s_barrier, ds_read x 8, s_waitcnt lgkmcnt(0), v_mfma`; tools/gemm_kloop_audit.py --sig).
        __builtin_amdgcn_s_waitcnt(0xC07F);
        KMB_WAIT_END(kmb_wait_ticks);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- sub-phase 3: A(k1, rows 64-127) x B(k1)  ||  read k0 of stage it+1, fetch stage it+2 into this buffer ----
      read_b(nxt, 0, fb[0]);
      read_a(nxt, 0, 0, fa[0]);
      dma_a(it & 1, 0, NA3);
      dma_b(it & 1, 0, NB3);
      mma(1, fa[1], fb[1]);
      __builtin_amdgcn_sched_group_barrier(0x100, NDB, 3);
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(4), 3);
      __builtin_amdgcn_sched_group_barrier(0x100, NDA, 3);
      __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF(4), 3);
#pragma unroll
      for (int q = 0; q < P3; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 3);  // VMEM (LDS-DMA)
        __builtin_amdgcn_sched_group_barrier(0x008, KMB_MF((NM - 8) / P3), 3);
      }
      __builtin_amdgcn_sched_barrier(0);
      } else {
      // 64-row wave block: two sub-phases of 32 MFMAs (fa[0] / fa[1] are the two K halves)
      advance_cursor();
      // ---- sub-phase a: A(k0) x B(k0)  ||  read B(k1), A(k1) ----
      read_b(cur, 1, fb[1]);
      read_a(cur, 1, 0, fa[1]);
      mma(0, fa[0], fb[0]);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NDB / 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NDB / 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NDA, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
      {
        KMB_WAIT_BEGIN();
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        KMB_WAIT_END(kmb_wait_ticks);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- sub-phase b: A(k1) x B(k1)  ||  read k0 of stage it+1, fetch stage it+2 into this buffer ----
      read_b(nxt, 0, fb[0]);
      read_a(nxt, 0, 0, fa[0]);
      dma_stage_a(it & 1);            // (the wait sits between this block's two sub-phases: one issue point only)
      dma_stage_b(it & 1, 2);
      mma(0, fa[1], fb[1]);
      __builtin_amdgcn_sched_group_barrier(0x100, NDB, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
      __builtin_amdgcn_sched_group_barrier(0x100, NDA, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 1);  // VMEM (LDS-DMA): 4 + 8 pieces
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef KMB_V11_MFMA32_TIMING
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = acc32[i >> 1][j >> 1][((i & 1) * 2 + (j & 1)) * 4 + e];
#endif
    // ---- epilogue of this tile (the next tile's first two stages are in flight / resident meanwhile) ----
    if constexpr (TRASM) KMB_TR_SYNC();   // the next tile's first fragments (asm reads of the last sub-phase) live across the epilogue: arrived before anything moves them
    [[maybe_unused]] const uint64_t kmb_t_epi = KMB_NOW();
    int tm, tn;
    decode_tile(tile, tm, tn);
    const int row0w = tm * BM4 + wm * WROWS, col0w = tn * BNT + wn * WCOLS;
    if (row0w < p.M && col0w < p.N && !KMB_DIAG_BIT(p.tile_order, 512)) {   // bit 9: epilogue ablation (diagnostic build only, tools/gemm_epilogue_bound.py)
      const bool interior = (row0w + WROWS <= p.M) && (col0w + WCOLS <= p.N);
      const bool hb = p.bias != nullptr, hr = p.residual != nullptr, hd = p.drop_thr16 != 0u, hc = p.colsum != nullptr;
      const bool hs = col0w < p.col_scale_n;   // wave-uniform when col_scale_n is a multiple of 128 (checked below)
      const bool lean_ok = interior && p.out_bf16 != nullptr && p.out_f32 == nullptr && !KMB_DIAG_BIT(p.tile_order, 256) &&
                           (p.col_scale_n <= 0 || (p.col_scale_n % WCOLS) == 0);
#define KMB_LEAN(B, S, A, R, D, C) v11_epilogue_lean<B, S, A, R, D, C, WROWS, false, NJ>(p, acc, ef, lane, r, g, row0w, col0w)
      if (interior && p.out_f32 != nullptr && p.out_bf16 == nullptr && p.beta == 0.f && (p.ld_out_f32 & 3) == 0 &&
          !KMB_DIAG_BIT(p.tile_order, 256) && p.act == 0 && hb && !hr && !hd && !hc && p.col_scale_n <= 0) {
        v11_epilogue_lean<true, false, 0, false, false, false, WROWS, true, NJ>(p, acc, ef, lane, r, g, row0w, col0w);   // logits
      } else if (lean_ok && p.act == 0 && hb && !hr && !hd && !hc) {
        if (hs) KMB_LEAN(true, true, 0, false, false, false);
        else KMB_LEAN(true, false, 0, false, false, false);
      } else if (lean_ok && p.act == 0 && hb && hr && !hc && !hs) {
        if (hd) KMB_LEAN(true, false, 0, true, true, false);
        else KMB_LEAN(true, false, 0, true, false, false);
      } else if (lean_ok && p.act == 0 && !hb && !hd && !hc && !hs) {
        if (hr) KMB_LEAN(false, false, 0, true, false, false);
        else KMB_LEAN(false, false, 0, false, false, false);
      } else if (lean_ok && p.act == 1 && hb && !hr && !hd && !hc && !hs) {
        KMB_LEAN(true, false, 1, false, false, false);
      } else if (lean_ok && p.act == 2 && !hb && !hr && !hd && hc && !hs) {
        KMB_LEAN(false, false, 2, false, false, true);
      } else if (lean_ok && p.act == 5 && hb && !hr && !hd && !hc && !hs && (WCOLS == 64 || WCOLS == 128)) {
        if constexpr (WCOLS == 64 || WCOLS == 128) KMB_LEAN(true, false, 5, false, false, false);
      } else {
        v11_epilogue<-1, true, true, false, WROWS, NJ>(p, acc, ef, lane, r, g, row0w, col0w);   // edges and rare classes
      }
#undef KMB_LEAN
    }
    [[maybe_unused]] const uint64_t kmb_t_drain = KMB_NOW();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): keeps the epilogue's pending loads out of the K loop's wait state
#ifdef KMB_GEMM_STAMP
    {
      const uint64_t now = __builtin_amdgcn_s_memrealtime();
      kmb_loop_ticks += kmb_t_epi - kmb_t_loop; kmb_epi_ticks += kmb_t_drain - kmb_t_epi; kmb_drain_ticks += now - kmb_t_drain;
    }
#endif
  }
  KMB_STAMP_VALUE(2, kmb_loop_ticks);
  KMB_STAMP_VALUE(3, kmb_epi_ticks);
  KMB_STAMP_VALUE(5, kmb_wait_ticks);
  KMB_STAMP_VALUE(6, kmb_drain_ticks);
  KMB_STAMP(4);
#if defined(KMB_GEMM_STAMP) && KMB_STAMP_SLOTS >= 12
  KMB_STAMP_VALUE(8, __builtin_amdgcn_s_memtime() - kmb_c0);
  KMB_STAMP_VALUE(9, __builtin_amdgcn_s_memrealtime() - kmb_r0);
#endif
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the two look-ahead fetches past the last tile target this LDS
  retire();
}

#ifndef KMB_GEMM_DEVICE_ONLY
// ------------------------------------------------------------------------------------------
// Narrow tile for the generation path.  A decode step multiplies M = batch x beams rows (320 at the benchmark
// setting) by every weight matrix: with 128x128 tiles that is 18-72 workgroups on 256 CUs and each launch costs a
// full tile latency (23 us measured, 56 GFLOP/step at 67 TFLOP/s).  Here a workgroup takes a 128 x 32 slice
// (72-288 workgroups for the same layers), four waves of 32 x 32, a four-deep LDS-DMA ring, same LDS images / swizzles /
// accumulation order as v7 (bit-identical results).  Forward layout only (X . W^T), epilogue = bias, q-scale, GeLU,
// residual, bf16 / fp32 store; everything else stays on v7.
constexpr int BNS = 32;
constexpr int STAGE_S = (BM + BNS) * BK * 2;            // 20 KiB
constexpr int EPI_LD_S = 36;                            // fp32 staging row stride (32 + 4: conflict-free b128 reads)
constexpr int NST_S = 4;                                // ring depth: three stages in flight (the K loop is pure latency)
constexpr int LDS_S = NST_S * STAGE_S;                  // 80 KiB: two workgroups per CU
static_assert(LDS_S >= BM * EPI_LD_S * 4, "epilogue staging must fit");

__global__ __launch_bounds__(256, 2) void gemm_kernel_narrow(const KmbGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int tiles_n = (p.N + BNS - 1) / BNS;
  const int nsl = p.split_k > 1 ? p.split_k : 1;   // split-K: slice s of a tile writes raw sums to slab[s][M][N]
  const int tile = (int)blockIdx.x / nsl, slice = (int)blockIdx.x % nsl;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int row0 = tm * BM, col0 = tn * BNS;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt_all = p.K / BK;
  const int t_begin = (int)((long)nt_all * slice / nsl), t_end = (int)((long)nt_all * (slice + 1) / nsl);
  const int nt = t_end - t_begin;
  uint32_t offA[4], offB;
  dma_offsets<true>(offA, p.lda, row0, p.M, wave, lane);
  {  // B: 32 rows x 128 B = 4 pieces, one per wave (rows 8 * wave ...), same swizzle as the 128-row image
    const int row = wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int grow = col0 + row;
    grow = grow < p.N ? grow : p.N - 1;
    offB = (uint32_t)(((grow - col0) * p.ldb + c * 8) * 2);
  }
  const char* gA = reinterpret_cast<const char*>(p.A) + (size_t)row0 * p.lda * 2 + (size_t)t_begin * (BK * 2);
  const char* gB = reinterpret_cast<const char*>(p.B) + (size_t)col0 * p.ldb * 2 + (size_t)t_begin * (BK * 2);
  constexpr int A_TILE = BM * BK * 2;
  auto dma_stage = [&](int ks, int buf) {
    char* da = smem + buf * STAGE_S + wave * 4096;
    char* db = smem + buf * STAGE_S + A_TILE + wave * 1024;
    const char* ga = gA + (size_t)ks * (BK * 2);
    const char* gb = gB + (size_t)ks * (BK * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(ga, offA[i], da + i * 1024);
    dma_piece(gb, offB, db);
  };
  for (int ks = 0; ks < NST_S - 1 && ks < nt; ++ks) dma_stage(ks, ks);
  for (int t = 0; t < nt; ++t) {
    // stage t has landed when at most the (<= 2) younger stages' pieces (5 per stage and wave) are still in flight
    const int younger = (nt - 1 - t) < (NST_S - 2) ? (nt - 1 - t) : (NST_S - 2);
    if (younger == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // also: every wave is done with the buffer the next DMA overwrites (read in step t - 1)
    if (t + NST_S - 1 < nt) dma_stage(t + NST_S - 1, (t + NST_S - 1) & (NST_S - 1));
    const char* cur = smem + (t & (NST_S - 1)) * STAGE_S;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = read_frag<true>(cur, wave * 2 + i, kk, r, g);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = read_frag<true>(cur + A_TILE, j, kk, r, g);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();
  // ---- epilogue: accumulators -> fp32 LDS [128][36] -> row-major math, 16-byte stores (4 lanes per row) ----
  float* ef = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) ef[(wave * 32 + i * 16 + g * 4 + q) * EPI_LD_S + j * 16 + r] = acc[i][j][q];
  __syncthreads();
  const int c8 = (tid & 3) * 8;
  const int gcol = col0 + c8;
  if (gcol >= p.N) return;
  const int nvalid = (p.N - gcol) < 8 ? (p.N - gcol) : 8;
  if (p.split_k > 1) {   // raw partial sums (the launcher requires N % 8 == 0)
    float* slab = p.slab + (size_t)slice * p.M * p.N;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int lrow = (tid >> 2) + 64 * it;
      const int grow = row0 + lrow;
      if (grow >= p.M) break;
      float* o = slab + (size_t)grow * p.N + gcol;
      *reinterpret_cast<f32x4*>(o) = *reinterpret_cast<const f32x4*>(ef + lrow * EPI_LD_S + c8);
      *reinterpret_cast<f32x4*>(o + 4) = *reinterpret_cast<const f32x4*>(ef + lrow * EPI_LD_S + c8 + 4);
    }
    return;
  }
  float bias8[8], scale8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bias8[e] = (p.bias != nullptr && e < nvalid) ? p.bias[gcol + e] : 0.f;
    scale8[e] = (gcol + e < p.col_scale_n) ? p.col_scale : 1.0f;
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int lrow = (tid >> 2) + 64 * it;
    const int grow = row0 + lrow;
    if (grow >= p.M) break;
    float v[8];
    const f32x4 lo = *reinterpret_cast<const f32x4*>(ef + lrow * EPI_LD_S + c8);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(ef + lrow * EPI_LD_S + c8 + 4);
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (v[e] + bias8[e]) * scale8[e];
    if (p.act == 1) {
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const kmb_f32x2 y = gelu2(kmb_f32x2{v[e], v[e + 1]});
        v[e] = y[0]; v[e + 1] = y[1];
      }
    }
    if (p.residual != nullptr) {
      for (int e = 0; e < nvalid; ++e) v[e] += bf2f(p.residual[(size_t)grow * p.ld_res + gcol + e]);
    }
    if (p.out_bf16 != nullptr) {
      if (nvalid == 8) {
        *reinterpret_cast<u32x4*>(p.out_bf16 + (size_t)grow * p.ld_out_bf16 + gcol) = pack8(v);
      } else {
        for (int e = 0; e < nvalid; ++e) p.out_bf16[(size_t)grow * p.ld_out_bf16 + gcol + e] = f2bf(v[e]);
      }
    }
    if (p.out_f32 != nullptr) {
      float* o = p.out_f32 + (size_t)grow * p.ld_out_f32 + gcol;
      for (int e = 0; e < nvalid; ++e) o[e] = v[e] + (p.beta != 0.f ? p.beta * o[e] : 0.f);
    }
  }
}


// Variants that were built, verified bit-identical and then REMOVED because they measured slower on every training
// shape (MI355X, b=256): a 256x128 three-stage ring with counted vmcnt (one workgroup per CU: -10..25 %), a
// transposed-block MFMA with a register epilogue and 8-byte stores (-20 %), several tiles per workgroup with the
// next tile's first K step prefetched under the epilogue (-5..30 %), and the un-pipelined forms of v7 / v8 (their K
// loops were 3-25 % slower).  A start-time phase stagger between co-resident workgroups / between CUs was also
// measured (stamps: the K loop gets shorter, the epilogue longer, the tile time does not move) and dropped, and so was
// a four-deep ring of 32-wide K stages for the 256x256 tile (DMA issued three steps ahead): bit-identical, same speed
// -- so DMA latency is not what parks v8's waves at the per-step barrier.  4096^3 runs at 1.29 PFLOP/s, within 3 % of
// the CDNA4 guide's 8-phase 256^2 template (1.32-1.34 on random operands).
// A register epilogue without LDS staging was built too: MFMA operands swapped so that a lane holds four consecutive
// columns (C^T blocks), one v_permlane16_swap_b32 per dword between neighbouring column blocks to make that eight
// (tools/permlane_probe.hip; note hipcc folds four __builtin_amdgcn_permlane16_swap of a vector's elements into one
// -- inline asm is required), math and 16-byte stores straight from registers.  Bit-identical, v7 +-0 %, v8 -13 %:
// the staging round trip is not what the epilogue waits for.  With that epilogue the stage buffers are free during it,
// so a PERSISTENT v7 was built as well (a workgroup walks tiles b, b + 512, ...; the next tile's first two stages are
// requested before the epilogue starts, hiding the 2.4 us prologue wait): bit-identical, 0..-20 % -- slower, not
// faster.  Time per tile round stays ~15 us however the phases are arranged; tools/gemm_ksweep.py puts it as
// T(K) = 25 us + K / (1.13 PFLOP/s-equivalent) for (16384, 3072, K).  Both removed again.
// What did pay: LDS-DMA staging, the software-pipelined K loop, one uniform branch into a class-specialised epilogue
// (instruction fetch, not the stores, bounded the generic one), hoisted epilogue loads, hardware bf16 conversion,
// split-K for the weight gradients, per-shape choice between the 128x128 and 256x256 tiles and the XCD tile order.

#endif  // KMB_GEMM_DEVICE_ONLY
}  // namespace

#ifndef KMB_GEMM_DEVICE_ONLY   // host side: checks, launch rules, tuner
void kmb_gemm_set_shared_device(int on) { g_shared_device = on ? 1 : 0; }

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
  if (p.act == 5 && (!p.row_shift || !p.row_sums || !p.bias || !p.out_bf16 || p.out_f32 || p.split_k > 1 || (p.M & 255) || (p.N & 255) ||
                     !p.a_kc || !p.b_kc || (p.K % BK) || p.K < 2 * BK || p.residual || p.colsum || p.drop_thr16 || p.col_scale_n > 0 ||
                     p.row_sums_ld < p.N / 64 || (long)(p.M / 256) * (p.N / 256) < 128 || (p.pick_col && !p.pick_out)))
    return "gemm: act 5 (exp with row sums) needs the persistent forward layout: M, N multiples of 256 with >= 128 tiles, bias, bf16 output, row_shift, row_sums";
  if (!p.a_kc && p.b_kc) return "gemm: (M-contiguous A, K-contiguous B) is not instantiated";
  if (p.colsum && p.split_k > 1) return "gemm: column sums are not available with split-K";
  if (p.split_k > 1) {
    if (!p.slab || ((uintptr_t)p.slab & 15)) return "gemm: split-K needs a 16-byte aligned slab";
    if (p.split_k > (p.K + BK - 1) / BK) return "gemm: more K slices than K steps";
  }
  return nullptr;
}

namespace {

// one workgroup per CU; fewer when there are fewer tiles (a multiple of 8: the per-XCD tile ranges)
unsigned v11_grid(const KmbGemm& p, int bn) {
  const long tiles = (long)((p.M + BM4 - 1) / BM4) * ((p.N + bn - 1) / bn);
  unsigned g = tiles >= 256 ? 256u : (unsigned)(tiles & ~7L);
  if (const char* e = KMB_DIAG_ENV("KMB_GEMM_GRID")) {   // diagnostic build (tools/epilogue_burst.py): fewer persistent workgroups, read per launch
    const unsigned v = (unsigned)atoi(e) & ~7u;
    if (v >= 8u && v < g) g = v;
  }
  return g;
}

// Tile counters of the persistent variants: 16 words per launch, zeroed once; the last workgroup of a launch leaves its
// slot zeroed again.  Slots are handed out from a ring PER STREAM (and device): kernels of one stream run in order, so a
// slot is free again long before its stream's ring comes back to it, and launches of different streams (the side
// stream's weight gradients, the optimizer's or RCCL's stream beside them) can never share a counter however far one
// stream runs ahead of another.
uint32_t* v11_sched_slot(hipStream_t stream) {
  constexpr int NSLOT = 8;
  struct Ring { uint32_t* base; unsigned seq; };
  static std::map<std::pair<int, hipStream_t>, Ring> rings;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  auto it = rings.find({dev, stream});
  if (it == rings.end()) {
    uint32_t* base = nullptr;
    if (hipMalloc(&base, NSLOT * 16 * sizeof(uint32_t)) != hipSuccess) return nullptr;
    if (hipMemset(base, 0, NSLOT * 16 * sizeof(uint32_t)) != hipSuccess) return nullptr;   // synchronous: done before any launch
    it = rings.emplace(std::make_pair(dev, stream), Ring{base, 0u}).first;
  }
  return it->second.base + (size_t)(it->second.seq++ % NSLOT) * 16;
}

// variant 1: register-staged 128x128 (any K); 7: LDS-DMA + pipelined 128x128; 8: LDS-DMA + pipelined 256x256;
// 10: role-split persistent 256x128 (eight waves: one group multiplies while the other fetches and runs the previous tile's epilogue);
// 11 / 12 / 13: persistent 256x256 / 256x128 / 256x192 (four waves); 14 / 15: persistent 256x256 / 256x192, eight waves
hipError_t launch_variant(int variant, const KmbGemm& p, hipStream_t stream) {
  const int nsl = p.split_k > 1 ? p.split_k : 1;
  if (variant == 6)   // eight-wave persistent kernel around the bare K loop (gemm_lean.hip); tiles from a counter while the device is shared
    return kmb_gemm_lean_launch(p, stream, g_shared_device ? v11_sched_slot(stream) : nullptr, g_shared_device);
  if (variant == 9) {   // two workgroups per CU (gemm_pair.hip)
    // its tiles are dealt statically: while another kernel (RCCL, kmb_gemm_shared_device) holds CUs, the persistent 256 x 128
    // kernel with dynamic hand-out takes the launch instead (same tile shape: every launch variant 9 admits, it admits)
    if (!g_shared_device) return kmb_gemm_pair_launch(p, stream);
    variant = 12;
  }
#ifdef KMB_WITH_ROLESPLIT   // experiment build only (tools/experiments/gemm_rolesplit.hip, build.py --variant rolesplit): not in the product library
  if (variant == 10) return kmb_gemm_rs_launch(p, stream);
#endif
  if (variant == 11) {
    dim3 grid(v11_grid(p, BN4)), block(256);
    uint32_t* sched = p.K / BK >= 4 ? v11_sched_slot(stream) : nullptr;
    if (p.a_kc && p.b_kc) hipLaunchKernelGGL((gemm_kernel_v11<true, true, 256>), grid, block, LDS11, stream, p, sched, g_shared_device);
    else if (p.a_kc) hipLaunchKernelGGL((gemm_kernel_v11<true, false, 256>), grid, block, LDS11, stream, p, sched, g_shared_device);
    else hipLaunchKernelGGL((gemm_kernel_v11<false, false, 256>), grid, block, LDS11, stream, p, sched, g_shared_device);
  } else if (variant == 12) {
    dim3 grid(v11_grid(p, 128)), block(256);
    uint32_t* sched = p.K / BK >= 4 ? v11_sched_slot(stream) : nullptr;
    if (p.a_kc && p.b_kc) hipLaunchKernelGGL((gemm_kernel_v11<true, true, 128>), grid, block, LDS12, stream, p, sched, g_shared_device);
    else if (p.a_kc) hipLaunchKernelGGL((gemm_kernel_v11<true, false, 128>), grid, block, LDS12, stream, p, sched, g_shared_device);
    else hipLaunchKernelGGL((gemm_kernel_v11<false, false, 128>), grid, block, LDS12, stream, p, sched, g_shared_device);
  } else if (variant == 13) {
    dim3 grid(v11_grid(p, 192)), block(256);
    uint32_t* sched = p.K / BK >= 4 ? v11_sched_slot(stream) : nullptr;
    if (p.a_kc && p.b_kc) hipLaunchKernelGGL((gemm_kernel_v11<true, true, 192>), grid, block, LDS11, stream, p, sched, g_shared_device);
    else if (p.a_kc) hipLaunchKernelGGL((gemm_kernel_v11<true, false, 192>), grid, block, LDS11, stream, p, sched, g_shared_device);
    else hipLaunchKernelGGL((gemm_kernel_v11<false, false, 192>), grid, block, LDS11, stream, p, sched, g_shared_device);
  } else if (variant == 14) {
    dim3 grid(v11_grid(p, BN4)), block(512);
    uint32_t* sched = p.K / BK >= 4 ? v11_sched_slot(stream) : nullptr;
    if (p.a_kc && p.b_kc) hipLaunchKernelGGL((gemm_kernel_v11<true, true, 256, 8>), grid, block, LDS11, stream, p, sched, g_shared_device);
    else if (p.a_kc) hipLaunchKernelGGL((gemm_kernel_v11<true, false, 256, 8>), grid, block, LDS11, stream, p, sched, g_shared_device);
    else hipLaunchKernelGGL((gemm_kernel_v11<false, false, 256, 8>), grid, block, LDS11, stream, p, sched, g_shared_device);
  } else if (variant == 15) {
    dim3 grid(v11_grid(p, 192)), block(512);
    uint32_t* sched = p.K / BK >= 4 ? v11_sched_slot(stream) : nullptr;
    if (p.a_kc && p.b_kc) hipLaunchKernelGGL((gemm_kernel_v11<true, true, 192, 8>), grid, block, LDS11, stream, p, sched, g_shared_device);
    else if (p.a_kc) hipLaunchKernelGGL((gemm_kernel_v11<true, false, 192, 8>), grid, block, LDS11, stream, p, sched, g_shared_device);
    else hipLaunchKernelGGL((gemm_kernel_v11<false, false, 192, 8>), grid, block, LDS11, stream, p, sched, g_shared_device);
  } else if (variant == 8) {
    const int tiles = ((p.M + BM4 - 1) / BM4) * ((p.N + BN4 - 1) / BN4);
    dim3 grid(tiles * nsl), block(512);
    if (p.a_kc && p.b_kc) hipLaunchKernelGGL((gemm_kernel_v8<true, true>), grid, block, LDS4, stream, p);
    else if (p.a_kc) hipLaunchKernelGGL((gemm_kernel_v8<true, false>), grid, block, LDS4, stream, p);
    else hipLaunchKernelGGL((gemm_kernel_v8<false, false>), grid, block, LDS4, stream, p);
  } else {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    dim3 grid(tiles * nsl), block(256);
    if (variant == 5) {   // four LDS stages, one workgroup per CU (v7d_ok)
      if (p.a_kc && p.b_kc) hipLaunchKernelGGL((gemm_kernel_v7d<true, true>), grid, block, LDS_DEEP, stream, p);
      else if (p.a_kc) hipLaunchKernelGGL((gemm_kernel_v7d<true, false>), grid, block, LDS_DEEP, stream, p);
      else hipLaunchKernelGGL((gemm_kernel_v7d<false, false>), grid, block, LDS_DEEP, stream, p);
    } else if (variant == 7) {
      if (p.a_kc && p.b_kc) hipLaunchKernelGGL((gemm_kernel_v7<true, true>), grid, block, LDS_BYTES, stream, p);
      else if (p.a_kc) hipLaunchKernelGGL((gemm_kernel_v7<true, false>), grid, block, LDS_BYTES, stream, p);
      else hipLaunchKernelGGL((gemm_kernel_v7<false, false>), grid, block, LDS_BYTES, stream, p);
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

// v7d (four LDS stages): launches of at most one workgroup per CU with a K loop worth pipelining
bool v7d_ok(const KmbGemm& p) {
  const long wgs = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * (p.split_k > 1 ? p.split_k : 1);
  const int nt = p.K / BK / (p.split_k > 1 ? p.split_k : 1);
  return wgs <= 256 && (p.K % BK) == 0 && nt >= 4 && p.act != 5;
}

// v11 (persistent, one workgroup per CU): enough tiles for half the CUs, two K steps, no split-K
bool v11_ok(const KmbGemm& p, int bn = BN4) {
  const long tiles = (long)((p.M + BM4 - 1) / BM4) * ((p.N + bn - 1) / bn);
  return p.split_k <= 1 && (p.K % BK) == 0 && p.K >= 2 * BK && tiles >= 128;   // at least half the CUs get a tile
}

// L2 prefetch of the activation operand by the persistent kernels (tile_order bit 1).  The tuning above times
// back-to-back launches, whose operands sit in the Infinity Cache, so it cannot see what the prefetch is for.
// Round 2 enabled it by a per-shape rule (K >= 2048 forward launches, >= 160 MB backward operands) taken from in-step
// timing at a time when the tuner often chose the round-robin tile order, under which the sharers of a row panel sit on
// eight different XCDs and the touch helps nobody.  With per-XCD tile ranges on every persistent launch (round 3) the
// sharers of a panel are on one XCD, and the prefetch pays on every shape: in-step GEMM time of a b = 1024 step, same box,
// alternating processes (tools/gemm_ab_seq.sh): always 39.5 / 39.6 / 41.2 ms, rule 43.3 / 42.7 / 43.7, never 43.5 / 43.4.
bool prefetch_a(const KmbGemm& p) {
  static int mode = -1;   // KMB_GEMM_PREFETCH = 0 (never) | 1 (always) | 2 (round 2's rule) | unset (always)
  if (mode < 0) {
    const char* e = KMB_DIAG_ENV("KMB_GEMM_PREFETCH");
    mode = e ? atoi(e) : 1;
  }
  if (mode != 2) return mode == 1 && p.a_kc;
  if (!p.a_kc || p.K < 2048) return false;
  return p.b_kc || (double)p.M * p.K * 2.0 >= 160e6;
}

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

namespace {

// A launch configuration: bits 0-3 kernel variant, bits 4-6 tile order, bit 8 / 9: L2 prefetch explicitly off / on (neither:
// prefetch_a()'s rule).  Applies the launch rules that do not depend on timing and launches.
hipError_t launch_config(const KmbGemm& p, int cfg, hipStream_t stream) {
  KmbGemm q = p;
  const int variant = cfg & 15;
  bool pf = prefetch_a(p);
  if (cfg & 0x100) pf = false;
  if (cfg & 0x200) pf = p.a_kc != 0;
  q.tile_order = ((cfg >> 4) & 7) | (pf ? 2 : 0);
  if ((variant == 6 || variant == 9 || variant == 10) && p.a_kc && p.b_kc && p.N >= 32 * 256) q.tile_order |= 8;   // role-split: per-XCD ranges always; column blocks for wide outputs
  if (variant >= 11) {
    // Persistent variants: per-XCD contiguous tile ranges ALWAYS (bit 0), column blocks for wide outputs (bit 3).  The
    // tuner's back-to-back timing cannot see the difference (operands sit in the Infinity Cache there); inside a step
    // the round-robin order pulls every activation row panel into all eight L2s -- rocprofv3 FETCH_SIZE per launch,
    // tools/r3_traffic.sh: fc1 forward 822 MB for 105 MB of operands, the N = 768 data gradients 1458 for 407 -- and
    // measures 0.5-2 % slower (tools/gemm_ab_env.sh).  KMB_GEMM_FORCE_ORDER = 0 | 1 overrides bit 0, KMB_GEMM_COLBLOCKS=0
    // switches the column blocks off (A/B measurements).
    static int fo = -2, cbk = -1;
    if (fo == -2) { const char* e = KMB_DIAG_ENV("KMB_GEMM_FORCE_ORDER"); fo = e ? atoi(e) : -1; }
    if (cbk < 0) { const char* e = KMB_DIAG_ENV("KMB_GEMM_COLBLOCKS"); cbk = e ? atoi(e) : 1; }
    q.tile_order = (q.tile_order & ~1) | (fo >= 0 ? (fo & 1) : 1);
    if (cbk && p.a_kc && p.b_kc && p.N >= 32 * 256) q.tile_order |= 8;
  }
  {
    // diagnostic (tools/gemm_epilogue_bound.py): KMB_GEMM_ABLATE_DYNAMIC=1 at process start makes the launcher re-read
    // KMB_GEMM_ABLATE at every launch; "1" skips every epilogue (outputs are NOT written: timing only)
    static const bool dyn_ablate = KMB_DIAG_ENV("KMB_GEMM_ABLATE_DYNAMIC") != nullptr;
    if (dyn_ablate) {
      const char* ab = KMB_DIAG_ENV("KMB_GEMM_ABLATE");
      if (ab && ab[0] == '1') q.tile_order |= 512;
    }
  }
  if (p.split_k > 1) {
    // Split-K launches: per-XCD contiguous ranges + slice-major enumeration ALWAYS (bits 0 and 2), like the persistent rule
    // above and for the same reason -- the tuner's back-to-back timing cannot tell the orders apart, the L2s can: with
    // the round-robin order the 12 column tiles that share a 768-row operand slice sit on eight XCDs (fc2's weight
    // gradient 768 x 3072 x 65536: 1544 MB fetched for 503 MB of operands, its transpose 3072 x 768 with the other order
    // 628; profiles/r03_gemm_traffic_by_shape_b1024.txt).  In-step: b = 1024 neutral, b = 256 -0.4...-1.3 %
    // (profiles/r03_ab_split_order_instep.txt).  KMB_GEMM_SPLIT_ORDER = 0 | 2: force slice-minor / leave the tuner's pick.
    static int so = -2;
    if (so == -2) { const char* e = KMB_DIAG_ENV("KMB_GEMM_SPLIT_ORDER"); so = e ? atoi(e) : 1; }
    if (so == 0) q.tile_order &= ~4;
    else if (so == 1) q.tile_order |= 5;
  }
  if (const char* e = KMB_DIAG_ENV("KMB_GEMM_STAGGER"))   // diagnostic build: start delay of a persistent workgroup, bits 16-23 (see gemm_kernel_v11)
    q.tile_order |= (atoi(e) & 255) << 16;
  if (p.act == 5) q.tile_order &= ~256;   // (diagnostic build) the store ablation has no exp / row-sum form: act 5 always takes its lean epilogue
  return launch_variant(variant, q, stream);
}

// In-step refinement of the tuner's choice.  The first launch of a shape ranks the variants by back-to-back launches on
// operands that sit in the Infinity Cache; inside a training step (operands just streamed out by the previous kernel,
// another stream's GEMM sharing the chip) the ranking is a different one: round 3 measured per-shape differences of up to
// +-12 % between the back-to-back winner and the runner-up inside a step (profiles/r03_ab_eightwave_variants_instep_b1024.txt),
// and +-10 % from the L2 prefetch depending on the shape.  Every variant returns the same bits, so exploring while the
// job runs is safe: the next launches of the shape cycle through the back-to-back front-runners (within 8 % of the best,
// at most three) x {L2 prefetch on, off}, each launch timed where it runs (two events on its own stream, read back
// lazily when a later launch of the shape finds them complete), and the configuration with the lowest mean of its two
// fastest of three samples is kept.  Entries preloaded from KMB_GEMM_TUNE_FILE are final.
// MEASURED, and therefore OFF unless KMB_GEMM_REFINE=1: the GEMM launches of a step timed one at a time get 2 % faster with
// the refined choices (41.0 / 41.7 / 41.1 ms against 42.2 / 42.0 / 42.2, same box), the step itself -- weight gradients
// overlapping on the second stream -- does not (53.41 / 53.57 / 53.39 ms against 53.37 / 53.62 / 53.32 after 30 warm-up
// steps), and while it explores it costs 1 % (52.6-52.8 against 52.1-52.2 with bench.py's 6 warm-up steps): a launch's
// time inside an overlapped step depends more on which kernel of the other stream it shares the chip with than on the
// variant, so three samples rank noise.
struct Refine {
  std::vector<int> cfg;
  std::vector<std::vector<float>> ms;
  struct Pend { hipEvent_t e0, e1; int idx; };
  std::vector<Pend> pend;
  std::vector<int> issued;
  bool done = true;
  int final_cfg = 7;
};
std::map<TuneKey, Refine> g_refine;
std::vector<hipEvent_t> g_refine_events;
constexpr int REFINE_SAMPLES = 3;

hipEvent_t refine_event() {
  if (!g_refine_events.empty()) { hipEvent_t e = g_refine_events.back(); g_refine_events.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

}  // namespace

// The all-rows kernel (gemm_kernel_allrows): nullptr if `p` can run on it.  Plain forward GEMM with fp32 output only.
const char* kmb_gemm_allrows_check(const KmbGemm& p) {
  if (!p.a_kc || !p.b_kc) return "all-rows GEMM: forward layout only";
  if (p.M <= 0 || p.M > VR) return "all-rows GEMM: at most 320 rows";
  if ((p.K % BK) || p.K < BK || (p.N & 3) || p.N <= 0) return "all-rows GEMM: K % 64 == 0, N % 4 == 0";
  if (p.act != 0 || p.residual || p.drop_thr16 || p.colsum || p.split_k > 1 || p.preact || p.aux || p.col_scale_n > 0 || p.beta != 0.f)
    return "all-rows GEMM: plain epilogue (bias) only";
  if (!p.out_f32 || p.out_bf16 || (p.ld_out_f32 & 3) || ((uintptr_t)p.out_f32 & 15) || (p.bias && ((uintptr_t)p.bias & 15)))
    return "all-rows GEMM: fp32 output, 16-byte aligned";
  if ((p.lda & 7) || (p.ldb & 7) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) return "all-rows GEMM: operand alignment";
  return nullptr;
}

// stats != nullptr: also the rows' per-block (maximum, sum-exp) pairs, kmb_gemm_allrows_stats_floats(p.N) floats:
// stats[(row * blocks + block) * 2] = max, [... + 1] = sum of exp(v - max) over the block's 256 columns (blocks = ceil(N / 256))
int kmb_gemm_allrows_blocks(int N) { return (N + VN - 1) / VN; }
size_t kmb_gemm_allrows_stats_floats(int N) { return (size_t)kmb_gemm_allrows_blocks(N) * 2 * VR; }
hipError_t kmb_gemm_allrows_launch(const KmbGemm& p, float* stats, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel_allrows<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_VOC);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_kernel_allrows<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_VOC);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (stats != nullptr)
    hipLaunchKernelGGL(gemm_kernel_allrows<true>, dim3((p.N + VN - 1) / VN), dim3(512), LDS_VOC, stream, p, stats);
  else
    hipLaunchKernelGGL(gemm_kernel_allrows<false>, dim3((p.N + VN - 1) / VN), dim3(512), LDS_VOC, stream, p, (float*)nullptr);
  return hipGetLastError();
}

// Grouped weight gradients (gemm_group_wgrad_kernel): nullptr if the n problems can go out as one launch
const char* kmb_gemm_group_check(const KmbGemm* probs, int n) {
  if (n < 1 || n > KMB_GEMM_GROUP_MAX) return "gemm group: 1 .. KMB_GEMM_GROUP_MAX problems";
  for (int i = 0; i < n; ++i) {
    const KmbGemm& p = probs[i];
    if (const char* why = kmb_gemm_check(p)) return why;
    if (p.a_kc || p.b_kc) return "gemm group: weight-gradient layout only (both operands token-major)";
    if ((p.K % BK) || p.K < BK) return "gemm group: the reduction length must be a multiple of 64";
    if (p.split_k > 1 || p.slab) return "gemm group: no split-K";
    if (!p.out_f32 || p.out_bf16 || p.act != 0 || p.residual || p.aux || p.preact || p.colsum || p.drop_thr16 || p.bias || p.col_scale_n > 0)
      return "gemm group: plain fp32 output only";
  }
  return nullptr;
}

hipError_t kmb_gemm_group_launch(const KmbGemm* probs, int n, hipStream_t stream) {
  // (Measured and dropped: an LDS request of 96 KB, which keeps the launch at ONE workgroup per CU so that the other 64 KB
  // slot of every CU stays free for the caller's stream -- the launch then takes two rounds and the step got 5 % slower at
  // b = 64, 10 % at b = 128; profiles/r05_grouped_weight_gradients.md.)
  constexpr int lds_req = LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_group_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_req);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  KmbGemmGroup grp;
  memset(&grp, 0, sizeof(grp));
  grp.n = n;
  int at = 0;
  for (int i = 0; i < n; ++i) {
    grp.p[i] = probs[i];
    grp.p[i].tile_order = 1;   // per-XCD contiguous tile ranges: the column tiles that share a dY panel sit on one XCD
    grp.p[i].split_k = 0;
    grp.blocks[i] = ((probs[i].M + BM - 1) / BM) * ((probs[i].N + BN - 1) / BN);
    grp.first[i] = at;
    at += (grp.blocks[i] + 7) & ~7;
  }
  for (int i = n; i <= KMB_GEMM_GROUP_MAX; ++i) grp.first[i] = at;
  hipLaunchKernelGGL(gemm_group_wgrad_kernel, dim3(at), dim3(256), lds_req, stream, grp);
  return hipGetLastError();
}

// Every variant computes bit-identical results (same per-element accumulation order), so the choice is pure
// speed: the first launch of a new shape times the eligible variants on the real operands (measure, don't guess).
hipError_t kmb_gemm_launch(const KmbGemm& p, hipStream_t stream) {
  static int forced = -1, autotune = 1, verbose = 0;
  static const char* tune_file = nullptr;   // KMB_GEMM_TUNE_FILE: choices are appended here and preloaded from here, so a
                                            // profiled run (rocprofv3 --pmc) contains no tuning launches
  if (forced < 0) {
    tune_file = getenv("KMB_GEMM_TUNE_FILE");
    if (tune_file) {
      if (FILE* f = fopen(tune_file, "r")) {
        TuneKey k;
        int best;
        while (fscanf(f, "%d %d %d %d %d %d %d %d", &k.akc, &k.bkc, &k.M, &k.N, &k.K, &k.split, &k.act, &best) == 8)
          g_best[k] = best;
        fclose(f);
      }
    }
    const char* ev = getenv("KMB_GEMM_VARIANT");
    forced = ev ? atoi(ev) : 0;
    const char* ea = getenv("KMB_GEMM_AUTOTUNE");
    if (ea && ea[0] == '0') autotune = 0;
    verbose = KMB_DIAG_ENV("KMB_GEMM_VERBOSE") != nullptr;
    (void)hipFuncSetAttribute((const void*)gemm_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v7<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v7<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v7<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v7d<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DEEP);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v7d<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DEEP);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v7d<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DEEP);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v8<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v8<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v8<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<true, true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS11);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<true, false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS11);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<false, false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS11);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<true, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS12);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<true, false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS12);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<false, false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS12);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<true, true, 192>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS11);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<true, false, 192>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS11);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<false, false, 192>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS11);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<true, true, 256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS11);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<true, false, 256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS11);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<false, false, 256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS11);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<true, true, 192, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS11);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<true, false, 192, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS11);
    (void)hipFuncSetAttribute((const void*)gemm_kernel_v11<false, false, 192, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS11);
  }
  const bool dma_ok = (p.K % BK) == 0;           // LDS-DMA variants have no K-edge zero fill
  const bool big = dma_ok && p.M > 128;
  if (!dma_ok) return launch_variant(1, p, stream);
  // generation path: few rows, forward layout, plain epilogue -> narrow tiles (more, shorter workgroups)
  {
    static int narrow_ok = -1;
    if (narrow_ok < 0) {
      const char* e = KMB_DIAG_ENV("KMB_GEMM_NARROW");
      narrow_ok = !(e && e[0] == '0');
      (void)hipFuncSetAttribute((const void*)gemm_kernel_narrow, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_S);
    }
    const int tiles128 = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    if (narrow_ok && !forced && p.a_kc && p.b_kc && p.M <= 512 && tiles128 < 128 && (p.split_k <= 1 || (p.N & 7) == 0) &&
        p.act <= 1 && p.preact == nullptr && p.colsum == nullptr && p.drop_thr16 == 0u && (p.act == 0 || p.aux == nullptr)) {
      dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BNS - 1) / BNS) * (p.split_k > 1 ? p.split_k : 1)), block(256);
      hipLaunchKernelGGL(gemm_kernel_narrow, grid, block, LDS_S, stream, p);
      return hipGetLastError();
    }
  }
  if (forced) {
    int v = forced;
    if (v == 5 && !v7d_ok(p)) v = 7;
    if (v == 6 && !kmb_gemm_lean_ok(p)) v = 11;
    if (v == 9 && !kmb_gemm_pair_ok(p)) v = 11;
#ifdef KMB_WITH_ROLESPLIT
    if (v == 10 && !kmb_gemm_rs_ok(p)) v = 11;
#else
    if (v == 10) v = 11;   // the role-split experiment is not part of this library
#endif
    if (v == 11 && !v11_ok(p)) v = 8;
    if (v == 12 && !v11_ok(p, 128)) v = 8;
    if (v == 13 && !v11_ok(p, 192)) v = 8;
    if (v == 14 && !v11_ok(p)) v = 8;
    if (v == 15 && !v11_ok(p, 192)) v = 8;
    if (v == 8 && !(big && p.N > 128)) v = 7;
    if (v != 1 && v != 5 && v != 6 && v != 7 && v != 8 && v != 9 && v != 10 && (v < 11 || v > 15)) v = 7;
    if (p.act == 5 && v != 6 && v != 9 && v != 10 && (v < 11 || v == 13 || v == 15)) v = 11;
    KmbGemm q = p;
    q.tile_order = p.tile_order | (prefetch_a(p) ? 2 : 0);
    return launch_variant(v, q, stream);
  }
  if (!big || p.N <= 128) return launch_variant(7, p, stream);
  const TuneKey key{p.a_kc, p.b_kc, p.M, p.N, p.K, p.split_k, p.act};
  auto it = g_best.find(key);
  if (it == g_best.end()) {
    if (!autotune || writes_an_input(p)) return p.act == 5 ? launch_config(p, 11, stream) : launch_variant(7, p, stream);
    // (variant 10, the role-split kernel, lives in tools/experiments/ since round 5: bit-identical, slower than the persistent
    //  variants on every benchmark-batch shape but two -- DESIGN.md section 4 "Round 4"; `build.py --variant rolesplit` links it)
    const int cands[21] = {5, 5 + 16, 5 + 16 * 5,                                          // four LDS stages (<= 256 workgroups)
                           7, 7 + 16, 8, 8 + 16, 11, 11 + 16, 12, 12 + 16, 13, 13 + 16,   // variant | (tile_order << 4)
                           14, 14 + 16, 15, 15 + 16,
                           7 + 16 * 5, 8 + 16 * 5,                                        // split-K only: slice-major
                           9,                                                             // two workgroups per CU (gemm_pair.hip)
                           6};                                                            // eight waves around the bare K loop (gemm_lean.hip)
    float best_ms = 1e30f;
    int best = p.act == 5 ? 11 : 7;
    std::vector<std::pair<float, int>> timed;   // (ms, candidate) of every eligible candidate
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return launch_variant(7, p, stream);
    static unsigned exclude = ~0u;   // KMB_GEMM_EXCLUDE=14,15: variants the tuner may not pick (same-box A/B measurements)
    if (exclude == ~0u) {
      exclude = 0u;
      if (const char* ex = KMB_DIAG_ENV("KMB_GEMM_EXCLUDE"))
        for (const char* q = ex; *q;) {
          const int v = atoi(q);
          if (v > 0 && v < 32) exclude |= 1u << v;
          while (*q && *q != ',') ++q;
          if (*q == ',') ++q;
        }
    }
    for (int c : cands) {
      if (exclude & (1u << (c & 15))) continue;
      if ((c & 15) == 5 && !v7d_ok(p)) continue;
      if ((c & 15) == 6 && !kmb_gemm_lean_ok(p)) continue;
      if ((c & 15) == 9 && !kmb_gemm_pair_ok(p)) continue;
      if ((c & 15) == 11 && !v11_ok(p)) continue;
      if ((c & 15) == 12 && !v11_ok(p, 128)) continue;
      if ((c & 15) == 13 && !v11_ok(p, 192)) continue;
      if ((c & 15) == 14 && !v11_ok(p)) continue;
      if ((c & 15) == 15 && !v11_ok(p, 192)) continue;
      if (p.act == 5 && (c & 15) != 6 && (c & 15) != 9 && (c & 15) != 10 && ((c & 15) < 11 || (c & 15) == 13 || (c & 15) == 15)) continue;   // lean epilogue of the 256- / 128-column persistent variants (and the role-split one) only
      if (((c >> 4) & 4) && p.split_k <= 1) continue;
      KmbGemm q = p;
      q.tile_order = c >> 4;
      hipError_t e = launch_variant(c & 15, q, stream);  // warm
      if (e != hipSuccess) return e;
      float ms = 1e30f;
      for (int round = 0; round < 2; ++round) {   // best of two timed rounds of three launches (one noisy round used
        (void)hipEventRecord(e0, stream);         // to flip close choices from run to run)
        for (int rep = 0; rep < 3; ++rep) (void)launch_variant(c & 15, q, stream);
        (void)hipEventRecord(e1, stream);
        if (hipEventSynchronize(e1) != hipSuccess) return hipGetLastError();
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0, e1);
        ms = t < ms ? t : ms;
      }
      if (verbose) fprintf(stderr, "[kmb gemm tune] akc=%d bkc=%d M=%d N=%d K=%d split=%d act=%d v%d order%d %.1f us\n",
                           p.a_kc, p.b_kc, p.M, p.N, p.K, p.split_k, p.act, c & 15, c >> 4, ms / 3 * 1e3);
      // diagnostic build: KMB_GEMM_BIAS6=<percent> ranks variant 6 as if it were that much faster than timed back-to-back (its
      // cross-tile L2 touch costs it ~4 us per tile in this loop and pays inside a step: DESIGN.md section 4 "Round 4")
      static const float bias6 = KMB_DIAG_ENV("KMB_GEMM_BIAS6") ? 1.f - 0.01f * (float)atof(KMB_DIAG_ENV("KMB_GEMM_BIAS6")) : 1.f;
      const float rank_ms = (c & 15) == 6 ? ms * bias6 : ms;
      if (rank_ms < best_ms) { best_ms = rank_ms; best = c; }
      timed.emplace_back(ms, c);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    it = g_best.emplace(key, best).first;
    static const bool refine_on = KMB_DIAG_ENV("KMB_GEMM_REFINE") && KMB_DIAG_ENV("KMB_GEMM_REFINE")[0] == '1';   // opt-in: see Refine
    Refine& R = g_refine[key];
    R.final_cfg = best;
    R.done = true;
    if (refine_on) {
      std::sort(timed.begin(), timed.end());
      std::vector<int> front;   // front-runners: distinct variants (one tile order each: the faster one) within 8 % of the best
      for (const auto& tc : timed) {
        if (tc.first > best_ms * 1.08f || front.size() >= 3) break;
        bool dup = false;
        for (int f : front) dup = dup || (f & 15) == (tc.second & 15);
        if (!dup) front.push_back(tc.second);
      }
      for (int f : front) {
        if ((f & 15) >= 11 && p.a_kc) { R.cfg.push_back(f | 0x200); R.cfg.push_back(f | 0x100); }
        else R.cfg.push_back(f);
      }
      if (R.cfg.size() > 1) {
        R.done = false;
        R.ms.assign(R.cfg.size(), {});
        R.issued.assign(R.cfg.size(), 0);
      }
    }
    if (tune_file && R.done) {
      if (FILE* f = fopen(tune_file, "a")) {
        fprintf(f, "%d %d %d %d %d %d %d %d\n", key.akc, key.bkc, key.M, key.N, key.K, key.split, key.act, best);
        fclose(f);
      }
    }
  }
  {
    auto rit = g_refine.find(key);
    if (rit != g_refine.end() && !rit->second.done) {
      Refine& R = rit->second;
      // harvest the launches of this shape that have completed since
      for (size_t i = 0; i < R.pend.size();) {
        if (hipEventQuery(R.pend[i].e1) == hipSuccess) {
          float t = 0.f;
          if (hipEventElapsedTime(&t, R.pend[i].e0, R.pend[i].e1) == hipSuccess) R.ms[R.pend[i].idx].push_back(t);
          g_refine_events.push_back(R.pend[i].e0);
          g_refine_events.push_back(R.pend[i].e1);
          R.pend[i] = R.pend.back();
          R.pend.pop_back();
        } else {
          ++i;
        }
      }
      (void)hipGetLastError();   // hipErrorNotReady of a query is not an error of this launch
      int pick = -1;
      bool all = true;
      for (size_t i = 0; i < R.cfg.size(); ++i) {
        if ((int)R.ms[i].size() < REFINE_SAMPLES) all = false;
        if (R.issued[i] < REFINE_SAMPLES && (pick < 0 || R.issued[i] < R.issued[pick])) pick = (int)i;
      }
      if (all) {
        float best_s = 1e30f;
        for (size_t i = 0; i < R.cfg.size(); ++i) {
          std::sort(R.ms[i].begin(), R.ms[i].end());
          const float sc = R.ms[i][0] + R.ms[i][1];   // the two fastest of three: one sample beside a long kernel of another stream does not decide
          if (sc < best_s) { best_s = sc; R.final_cfg = R.cfg[i]; }
        }
        if (verbose) {
          fprintf(stderr, "[kmb gemm refine] akc=%d bkc=%d M=%d N=%d K=%d split=%d act=%d:", p.a_kc, p.b_kc, p.M, p.N, p.K, p.split_k, p.act);
          for (size_t i = 0; i < R.cfg.size(); ++i)
            fprintf(stderr, " v%d|o%d|%s %.1f", R.cfg[i] & 15, (R.cfg[i] >> 4) & 7, (R.cfg[i] & 0x200) ? "pf" : (R.cfg[i] & 0x100) ? "nopf" : "rule",
                    (R.ms[i][0] + R.ms[i][1]) * 500.f);
          fprintf(stderr, " -> v%d\n", R.final_cfg & 15);
        }
        R.done = true;
        it->second = R.final_cfg;
        for (auto& pe : R.pend) { g_refine_events.push_back(pe.e0); g_refine_events.push_back(pe.e1); }   // none left: all sampled
        R.pend.clear();
        if (tune_file) {
          if (FILE* f = fopen(tune_file, "a")) {
            fprintf(f, "%d %d %d %d %d %d %d %d\n", key.akc, key.bkc, key.M, key.N, key.K, key.split, key.act, R.final_cfg);
            fclose(f);
          }
        }
      } else if (pick >= 0) {
        hipEvent_t a = refine_event(), b = refine_event();
        if (a && b) {
          (void)hipEventRecord(a, stream);
          const hipError_t e = launch_config(p, R.cfg[pick], stream);
          (void)hipEventRecord(b, stream);
          R.pend.push_back({a, b, pick});
          R.issued[pick] += 1;
          return e;
        }
      } else {
        // every configuration has its launches in flight: run the back-to-back choice until they complete
      }
    }
  }
  return launch_config(p, it->second, stream);
}
#endif  // KMB_GEMM_DEVICE_ONLY
