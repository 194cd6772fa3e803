// Variant 9: TWO persistent workgroups per CU (forward layout X W^T: both operands K-contiguous).
//
// Why.  The persistent kernels of gemm.hip run one workgroup per CU -- 160 KB of LDS, all 512 registers of a lane -- so
// a tile's epilogue (20 % of the in-step GEMM time: CU-side work, tools/epilogue_burst.py) runs with the matrix pipe
// idle, and every 1 KiB LDS-DMA piece a wave issues holds that wave's MFMAs back while the vector-memory path accepts it.
// The role-split kernel (variant 10) put both kinds of work into a partner wave of the SAME workgroup and lost: one
// barrier per K step couples the partners, and the partner is one in-order instruction stream.  Here the partner is
// another WORKGROUP: two independent 4-wave workgroups per CU (256 registers per lane each, one wave of each per SIMD),
// each walking its own 256 x 128 tiles with its own barriers, so one's epilogue, tile switch and issue stalls sit beside
// the other's MFMAs without any coupling in the source.
//
// What makes it fit.  Two workgroups per CU get 80 KB of LDS each, and a 256 x 128 tile's 64-deep stage is 48 KB: two
// do not fit.  So the stages are 32 deep (24 KB; one MFMA k-depth) and there are three of them (72 KB): while step q
// multiplies stage q, stage q + 1 has landed and stages q + 2 / q + 3 are being fetched (q + 3 into the buffer stage q's
// last fragments left at the step's barrier).  A 32-deep slice of a K-contiguous operand is HALF a 128-byte line per
// row; measured (tools/dma_rate.hip, profiles/r04_lds_dma_piece_shape_rate.txt) the CU takes half-line pieces in at the
// rate of whole lines from memory and at 60-90 % of it from L2 -- affordable.  A row of the LDS image is 64 bytes, a piece
// (1 KiB) is one 16-row MFMA tile; the 16-byte chunks of a row are XOR-swizzled (pr_swz) so that the four 16-lane groups
// of a ds_read_b128 each cover all 64 banks; the swizzle is applied to the per-lane SOURCE address of the piece.
// The epilogue's fp32 staging image (4 KB per wave) lives in the stage buffer the tile's last K step has just left.
//
// Same MFMA, same k order per accumulator, same epilogue code (v11_epilogue_lean) as every other variant: bit-identical
// results (tools/gemm_v11_check.py).
//
// Limits (kmb_gemm_pair_ok): both operands K-contiguous, M % 256 == 0, N % 128 == 0, K % 192 == 0 (the K loop is unrolled
// over the three buffers x the two fragment sets), no split-K, bf16 output, one of the lean epilogue classes.
#define KMB_GEMM_DEVICE_ONLY
#include "gemm.hip"

namespace {

constexpr int PR_BM = 256, PR_BN = 128, PR_BK = 32;
constexpr int PR_A_BYTES = PR_BM * PR_BK * 2;           // 16 KB
constexpr int PR_STG = (PR_BM + PR_BN) * PR_BK * 2;     // 24 KB
constexpr int PR_NSTG = 3;
constexpr int PR_LDS = PR_NSTG * PR_STG;                // 72 KB: two workgroups per CU
constexpr int PR_EPW = 16 * 64 * 4;                     // fp32 staging image of a wave: 16 rows x 64 columns

enum { PR_BIAS = 0, PR_BIAS_RES = 1, PR_PLAIN = 2, PR_GELU = 3, PR_DGELU_CS = 4, PR_CE = 5 };

// chunk swizzle of a 64-byte image row (r = row within its 16-row tile): with chunk' = chunk ^ pr_swz(r) the lanes of every
// ds_read_b128 group ({0-3, 12-15, 20-27}, ... -- MI355X_MICROARCH.md, LDS) hit 16 distinct 16-byte bank groups
__device__ __forceinline__ int pr_swz(int r) { return (0x78 >> (((r >> 2) & 3) * 2)) & 3; }

__device__ __forceinline__ const char* pr_uniform(const char* ptr) {
  const uint64_t a = reinterpret_cast<uint64_t>(ptr);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

template <int EC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) __attribute__((amdgpu_num_vgpr(255))) void gemm_kernel_pair(const KmbGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;          // 2 x 2 waves of 128 x 64
  const int r = lane & 15, g = lane >> 4;
  constexpr int NJ = 4;
  const int tiles_n = p.N / PR_BN, tiles_m = p.M / PR_BM;
  const int ntiles = tiles_m * tiles_n;
  // column blocks for wide outputs (the tied head), as in gemm_kernel_v11
  constexpr int CB = 8;
  const bool col_blocks = (p.tile_order & 8) != 0 && tiles_n > CB;
  const int cb_full = tiles_n / CB;
  auto decode_tile = [&](int t, int& tm, int& tn) {
    if (!col_blocks) { tm = t / tiles_n; tn = t - tm * tiles_n; return; }
    const int blk = t / (CB * tiles_m);
    if (blk < cb_full) {
      const int rem = t - blk * (CB * tiles_m);
      tm = rem / CB; tn = blk * CB + (rem - tm * CB);
    } else {
      const int wl = tiles_n - cb_full * CB;
      const int rem = t - cb_full * (CB * tiles_m);
      tm = rem / wl; tn = cb_full * CB + (rem - tm * wl);
    }
  };
  // XCD x (= blockIdx % 8) owns a contiguous tile range; its workgroups take every (grid / 8)-th tile of it
  const int per = (int)gridDim.x >> 3;
  const int xcd = (int)blockIdx.x & 7, loc = (int)blockIdx.x >> 3;
  const int tq = ntiles >> 3, trem = ntiles & 7;
  const int range0 = xcd < trem ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq;
  const int range1 = range0 + tq + (xcd < trem ? 1 : 0);
  const int first_tile = range0 + loc;
  if (first_tile >= range1) return;
  const int nt = p.K / PR_BK;   // a multiple of 6 (launcher)

  // ---- LDS-DMA: per-lane source offsets are kernel constants (interior tiles only), the tile base is scalar ----
  uint32_t offA[4], offB[2];
  {
    const int rr = lane >> 2, ch = ((lane & 3) ^ pr_swz(rr)) << 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) offA[i] = (uint32_t)(((wave * 4 + i) * 16 + rr) * p.lda * 2 + ch);
#pragma unroll
    for (int i = 0; i < 2; ++i) offB[i] = (uint32_t)(((wave * 2 + i) * 16 + rr) * p.ldb * 2 + ch);
  }
  const char *gA_d, *gB_d;
  int tile_d = first_tile, td = 0;
  // ---- L2 touch of the activation panel (as in gemm_kernel_v11, "L2 prefetch of the activation operand") ----
  // Two 24 KB stages in flight per workgroup cover an L2 round trip, not a memory one (2.2 us: profiles/r04_lds_dma_piece_shape_
  // rate.txt), and inside a step A was just streamed out by the previous kernel.  So the workgroups that share a row panel (the
  // tiles of one tm: consecutive tiles, running side by side on one XCD) each touch THEIR share of the panel's rows PR_PFD steps
  // ahead of the DMA cursor -- one load instruction per wave behind every stage's pieces, a 128-byte line serves two steps, the
  // step's parity picks which half of the wave's rows it touches -- and the pieces of all sharers then hit L2.  The load's result
  // is never used; every wait leaves it outstanding (vmcnt 8 = the previous touch, a stage's six pieces, this touch) so that it
  // has three steps to land; it has no register destination: it lands in a dummy LDS word behind the stages (KMB_L2_TOUCH, gemm.hip).
  constexpr int PR_PFD = 4;
  const int sharers = col_blocks ? CB : tiles_n;
  int pf_share = (PR_BM + sharers - 1) / sharers;
  pf_share = pf_share > 256 ? 256 : pf_share;
  const int pf_gs = (pf_share + 7) >> 3;                  // rows per (wave, parity) group; <= 32
  const char *gA_tile, *gA_nx;                            // panel bases of the cursor's tile and of the workgroup's tile after it
  int pf_rows = 0, pf_rows_nx = 0;                        // first row of this workgroup's share in those panels
  auto set_dma_tile = [&](int tile) {
    int tm, tn;
    decode_tile(tile, tm, tn);
    gA_d = pr_uniform(reinterpret_cast<const char*>(p.A) + (size_t)tm * PR_BM * p.lda * 2);
    gB_d = pr_uniform(reinterpret_cast<const char*>(p.B) + (size_t)tn * PR_BN * p.ldb * 2);
    gA_tile = gA_d;
    pf_rows = (col_blocks ? tn % CB : tn) * pf_share;
    const int tx = tile + per < range1 ? tile + per : tile;
    decode_tile(tx, tm, tn);
    gA_nx = pr_uniform(reinterpret_cast<const char*>(p.A) + (size_t)tm * PR_BM * p.lda * 2);
    pf_rows_nx = (col_blocks ? tn % CB : tn) * pf_share;
  };
  const unsigned touch_lds = kmb_lds_addr(smem + PR_LDS + wave * 256);
  auto touch = [&]() {   // behind a stage's pieces (td = the step the NEXT fetch asks for)
    const int ps = td - 1 + PR_PFD;                        // the step whose line is touched (with its odd neighbour's)
    const bool nx = ps >= nt;
    const int s2 = (nx ? ps - nt : ps) & ~1;
    const char* sbase = pr_uniform((nx ? gA_nx : gA_tile) + s2 * (PR_BK * 2));
    const int grp = (wave * 2 + (ps & 1)) * pf_gs;
    int row = grp + (lane < pf_gs ? lane : pf_gs - 1);
    row = row < pf_share ? row : pf_share - 1;
    row += nx ? pf_rows_nx : pf_rows;
    row = row < PR_BM ? row : PR_BM - 1;
    const uint32_t voff = (uint32_t)row * (uint32_t)p.lda * 2u;
#if !defined(KMB_PR_NODMA) && !defined(KMB_PR_NOTOUCH)
    KMB_L2_TOUCH(voff, sbase, touch_lds);   // landing words behind the three stages (never read)
#else
    asm volatile("" ::"v"(voff), "s"(sbase));
#endif
  };
  // the next stage of the DMA cursor into buffer `buf`; past the workgroup's last tile the last tile's steps are fetched
  // again (never read) so that every wait sees the same number of younger pieces
  auto advance_cursor = [&]() {   // before a fetch, outside the scheduled regions (it branches)
    if (td == nt) {
      td = 0;
      if (tile_d + per < range1) tile_d += per;
      set_dma_tile(tile_d);
    }
  };
  auto dma_stage = [&](int buf) {
    char* da = smem + buf * PR_STG + wave * 4096;
    char* db = smem + buf * PR_STG + PR_A_BYTES + wave * 2048;
#ifndef KMB_PR_NODMA   // (timing experiments only: -DKMB_PR_NODMA / -DKMB_PR_NOEPI / -DKMB_PR_NOTOUCH builds compute garbage)
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(gA_d, offA[i], da + i * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma_piece(gB_d, offB[i], db + i * 1024);
#endif
    gA_d = pr_uniform(gA_d + PR_BK * 2);
    gB_d = pr_uniform(gB_d + PR_BK * 2);
    ++td;
  };

  // ---- fragments ----
  const int frag_c = r * 64 + ((g ^ pr_swz(r)) << 4);   // lane constant of every fragment address
  bf16x8 fa0[4], fa1[4], fbx[4], fby[4];
  f32x4 acc[8][NJ];
  auto read_a = [&](const char* stage, int half, bf16x8 (&dst)[4]) {
    const char* base = stage + (wm * 8 + half * 4) * 1024 + frag_c;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(base + i * 1024);
  };
  auto read_b = [&](const char* stage, bf16x8 (&dst)[4]) {
    const char* base = stage + PR_A_BYTES + (wn * 4) * 1024 + frag_c;
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(base + j * 1024);
  };
  auto mma = [&](int half, const bf16x8 (&a)[4], const bf16x8 (&b)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[half * 4 + i][j], 0, 0, 0);   // C^T tile
  };

  // One K step.  BUF: this step's stage buffer (compile time); the next stage is in BUF + 1, the fetch goes into BUF.
  //   phase A: A(rows 0-63) x B  ||  read A(rows 64-127)
  //   the 6 youngest operations (stage q + 2's pieces) may stay in flight: stage q + 1 has landed; barrier
  //   phase B: A(rows 64-127) x B  ||  read B, A(rows 0-63) of stage q + 1  ||  fetch stage q + 3 into this step's buffer
  // LAST (a tile's last step): phase B is the MFMAs only -- the buffer becomes the epilogue's staging image, its fetch and the
  // next tile's first fragments follow the epilogue.
  auto kstep = [&](auto buf_c, auto last_c, bf16x8 (&fb)[4], bf16x8 (&fbn)[4]) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    constexpr int NXT = (BUF + 1) % PR_NSTG;
    const char* cur = smem + BUF * PR_STG;
    const char* nxt = smem + NXT * PR_STG;
    if constexpr (!LAST) advance_cursor();
    __builtin_amdgcn_sched_barrier(0);
    read_a(cur, 1, fa1);
    mma(0, fa0, fb);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0): touch, six pieces, touch may stay in flight
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!LAST) {
      read_b(nxt, fbn);
      read_a(nxt, 0, fa0);
      dma_stage(BUF);
    }
    mma(1, fa1, fb);
    if constexpr (!LAST) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 1);   // VMEM (LDS-DMA)
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!LAST) touch();
    __builtin_amdgcn_sched_barrier(0);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using B2 = std::integral_constant<int, 2>;
  using Yes = std::true_type;
  using No = std::false_type;

#ifdef KMB_PR_STAGGER   // experiment: the second workgroup of a CU (dispatched in the second half of the grid) starts late, so that
  if ((int)blockIdx.x >= ((int)gridDim.x >> 1)) {   // its epilogues fall into the first one's K loops
    for (int i = 0; i < KMB_PR_STAGGER * nt / 24; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  // ---- prologue: stages 0, 1, 2 of the first tile ----
  set_dma_tile(tile_d);
  dma_stage(0);
  touch();
  dma_stage(1);
  touch();
  dma_stage(2);   // (nt >= 6: no tile change inside the prologue)
  touch();
  __builtin_amdgcn_s_waitcnt(0x0F7F);   // vmcnt(15) = touch, stage 1, touch, stage 2, touch: stage 0 has landed
  __builtin_amdgcn_s_barrier();
  read_b(smem, fbx);
  read_a(smem, 0, fa0);

  for (int tile = first_tile; tile < range1; tile += per) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // every tile starts in buffer 0 with its B fragments in fbx (nt is a multiple of 6)
    for (int t = 0; t + 6 < nt; t += 6) {
      kstep(B0{}, No{}, fbx, fby);
      kstep(B1{}, No{}, fby, fbx);
      kstep(B2{}, No{}, fbx, fby);
      kstep(B0{}, No{}, fby, fbx);
      kstep(B1{}, No{}, fbx, fby);
      kstep(B2{}, No{}, fby, fbx);
    }
    kstep(B0{}, No{}, fbx, fby);
    kstep(B1{}, No{}, fby, fbx);
    kstep(B2{}, No{}, fbx, fby);
    kstep(B0{}, No{}, fby, fbx);
    kstep(B1{}, No{}, fbx, fby);
    kstep(B2{}, Yes{}, fby, fbx);
    // ---- epilogue: staging in buffer 2 (every wave is past the last step's barrier: nobody reads it any more); stages 0 and
    // 1 of the next tile are in flight / resident in buffers 0 and 1 ----
#ifdef KMB_PR_NOEPI
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(acc[i][j]));   // keep the MFMAs alive
    if (false)
#endif
    {
      int tm, tn;
      decode_tile(tile, tm, tn);
      const int row0w = tm * PR_BM + wm * 128, col0w = tn * PR_BN + wn * 64;
      float* const ef = reinterpret_cast<float*>(smem + 2 * PR_STG + wave * PR_EPW);
      const bool hs = col0w < p.col_scale_n, hd = p.drop_thr16 != 0u, hr = p.residual != nullptr;
#define KMB_PR_LEAN(B, S, A, R, D, C) v11_epilogue_lean<B, S, A, R, D, C, 128, false, NJ>(p, acc, ef, lane, r, g, row0w, col0w)
      if constexpr (EC == PR_BIAS) { if (hs) KMB_PR_LEAN(true, true, 0, false, false, false); else KMB_PR_LEAN(true, false, 0, false, false, false); }
      if constexpr (EC == PR_BIAS_RES) { if (hd) KMB_PR_LEAN(true, false, 0, true, true, false); else KMB_PR_LEAN(true, false, 0, true, false, false); }
      if constexpr (EC == PR_PLAIN) { if (hr) KMB_PR_LEAN(false, false, 0, true, false, false); else KMB_PR_LEAN(false, false, 0, false, false, false); }
      if constexpr (EC == PR_GELU) KMB_PR_LEAN(true, false, 1, false, false, false);
      if constexpr (EC == PR_DGELU_CS) KMB_PR_LEAN(false, false, 2, false, false, true);
      if constexpr (EC == PR_CE) KMB_PR_LEAN(true, false, 5, false, false, false);
#undef KMB_PR_LEAN
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();             // every wave is done with its staging image
    __builtin_amdgcn_sched_barrier(0);
    advance_cursor();
    dma_stage(2);                             // the fetch the last step left out (the next tile's step 2)
    touch();
    __builtin_amdgcn_s_waitcnt(0x0F77);       // vmcnt(7): everything older than these -- the next tile's stages 0 and 1, the epilogue -- is done
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_b(smem, fbx);
    read_a(smem, 0, fa0);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): nothing of this workgroup is in flight when it ends
}

int pr_class(const KmbGemm& p) {
  const bool hb = p.bias != nullptr, hr = p.residual != nullptr, hd = p.drop_thr16 != 0u, hc = p.colsum != nullptr;
  const bool hs = p.col_scale_n > 0;
  if (p.act == 0 && hb && !hr && !hd && !hc) return PR_BIAS;
  if (p.act == 0 && hb && hr && !hc && !hs) return PR_BIAS_RES;
  if (p.act == 0 && !hb && !hd && !hc && !hs) return PR_PLAIN;
  if (p.act == 1 && hb && !hr && !hd && !hc && !hs) return PR_GELU;
  if (p.act == 2 && !hb && !hr && !hd && hc && !hs) return PR_DGELU_CS;
  if (p.act == 5 && hb && !hr && !hd && !hc && !hs) return PR_CE;
  return -1;
}

}  // namespace

bool kmb_gemm_pair_ok(const KmbGemm& p) {
  if (!p.a_kc || !p.b_kc || p.split_k > 1 || (p.K % (6 * PR_BK)) != 0) return false;
  if ((p.M % PR_BM) != 0 || (p.N % PR_BN) != 0) return false;
  if ((long)(p.M / PR_BM) * (p.N / PR_BN) < 128) return false;
  if (p.out_bf16 == nullptr || p.out_f32 != nullptr || p.beta != 0.f) return false;
  if (p.col_scale_n > 0 && (p.col_scale_n % 64) != 0) return false;
  if (p.act == 5 && (p.row_shift == nullptr || p.row_sums == nullptr)) return false;
  if ((long)p.lda * 2 * PR_BM >= (1L << 31) || (long)p.ldb * 2 * PR_BN >= (1L << 31)) return false;   // 32-bit piece offsets
  return pr_class(p) >= 0;
}

hipError_t kmb_gemm_pair_launch(const KmbGemm& p, hipStream_t stream) {
  if (!kmb_gemm_pair_ok(p)) return hipErrorInvalidValue;
  const long tiles = (long)(p.M / PR_BM) * (p.N / PR_BN);
  const dim3 grid(tiles >= 512 ? 512u : (unsigned)(tiles & ~7L));
#define KMB_PR_CASE(E)                                                                                                        \
  case E: {                                                                                                                   \
    static bool attr = false;                                                                                                 \
    if (!attr) {                                                                                                              \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel_pair<E>, hipFuncAttributeMaxDynamicSharedMemorySize, PR_LDS + 1024); \
      if (e != hipSuccess) return e;                                                                                          \
      attr = true;                                                                                                            \
    }                                                                                                                         \
    hipLaunchKernelGGL((gemm_kernel_pair<E>), grid, dim3(256), PR_LDS + 1024, stream, p);                                             \
    break;                                                                                                                    \
  }
  switch (pr_class(p)) {
    KMB_PR_CASE(PR_BIAS)
    KMB_PR_CASE(PR_BIAS_RES)
    KMB_PR_CASE(PR_PLAIN)
    KMB_PR_CASE(PR_GELU)
    KMB_PR_CASE(PR_DGELU_CS)
    KMB_PR_CASE(PR_CE)
    default: return hipErrorInvalidValue;
  }
#undef KMB_PR_CASE
  return hipGetLastError();
}
