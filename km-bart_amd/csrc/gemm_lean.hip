// Variant 6: the persistent 256 x 256 eight-wave GEMM REBUILT AROUND ITS K LOOP (forward X W^T and data-gradient dY W layouts).
//
// Why.  tools/mfma_loop.hip times the K step of the 256 x 256 tile with nothing around it: eight waves (128 x 64 blocks, two
// per SIMD), 24 fragment reads and 64 MFMAs per wave, one barrier, the next-but-one stage fetched by LDS-DMA in one burst
// behind the barrier -- 1.42 us per step, 1.5 PFLOP/s, 77 % of the peak at the clock the chip holds (1.2 us / 1.8 PFLOP/s
// without the fetch).  The same step inside gemm_kernel_v11<.., 8> (variant 14) takes 1.64 us WITHOUT its LDS-DMA and
// ~2 us with it (profiles/r04_gemm_in_kernel_clock_and_bare_loop.txt): what it carries besides the step -- per-use address
// rebuilding, the tile-decode / cursor / L2-touch arithmetic and its branches inside the scheduled region (13 branches and 32
// spilled-scalar reloads per step in the ISA), the dynamic tile hand-out -- costs more than any of it buys.  This kernel is the
// microbenchmark's loop with the minimum around it:
//   * per-XCD contiguous tile ranges, dealt statically -- or from an atomic counter while another kernel shares the device --,
//     every tile interior (M % 256 == N % 256 == 0, K % 64 == 0);
//   * per-lane LDS-DMA offsets are kernel constants, the tile bases scalar; the cursor's tile switch is the only branch in a step;
//   * a tile's steps run in one branch-free loop; the last step leaves out the next fragments' reads, the lean epilogue
//     (v11_epilogue_lean, one class per kernel instance) follows, then the reads;
//   * the L2 touch of the activation panel as one register-free load per wave behind a stage's pieces; no role split, no second accumulator set.
// Same LDS images, swizzles, MFMA operand order and k order per accumulator as every other variant: bit-identical results
// (tools/gemm_v11_check.py).
#define KMB_GEMM_DEVICE_ONLY
#include "gemm.hip"

namespace {

constexpr int LN_STG = (256 + 256) * BK * 2;          // 64 KB
constexpr int LN_A = 256 * BK * 2;                    // 32 KB
constexpr int LN_EPW = 16 * 64 * 4;                   // fp32 staging image of a wave (16 rows x 64 columns)
constexpr int LN_LDS = 2 * LN_STG + 8 * LN_EPW;       // 160 KB

enum { LN_BIAS = 0, LN_BIAS_RES = 1, LN_PLAIN = 2, LN_GELU = 3, LN_DGELU_CS = 4, LN_CE = 5 };

__device__ __forceinline__ const char* ln_uniform(const char* ptr) {
  const uint64_t a = reinterpret_cast<uint64_t>(ptr);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

template <bool B_KC, int EC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_kernel_lean(const KmbGemm p, uint32_t* sched, int dyn_first) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves of 128 x 64
  const int r = lane & 15, g = lane >> 4;
  constexpr int NJ = 4;
  const int tiles_n = p.N / 256, tiles_m = p.M / 256;
  const int ntiles = tiles_m * tiles_n;
  constexpr int CB = 8;   // column blocks for wide outputs, as in gemm_kernel_v11
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
  const int per = (int)gridDim.x >> 3;
  const int xcd = (int)blockIdx.x & 7, loc = (int)blockIdx.x >> 3;
  const int tq = ntiles >> 3, trem = ntiles & 7;
  const int range0 = xcd < trem ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq;
  const int range1 = range0 + tq + (xcd < trem ? 1 : 0);
  const int nt = p.K / BK;   // >= 5 (launcher; >= 8 when tiles are handed out dynamically)
  // Tiles after the first (with dyn_first: all of them) come from an atomic counter per XCD range when another kernel shares the
  // device (sched != nullptr: kmb_gemm_shared_device, an RCCL exchange on the communication stream): a workgroup that gets its CU
  // late takes fewer tiles instead of leaving a fixed share to a straggling round.  As in gemm_kernel_v11: sched[0..7] tile counters,
  // sched[8] finished workgroups, the last one zeroes them; the next tile is published through one LDS word.
  const bool dyn = sched != nullptr;
  uint32_t* const my_ctr = sched + xcd;
  auto retire = [&]() {
    if (dyn && tid == 0) {
      if (atomicAdd(sched + 8, 1u) == gridDim.x - 1u) {
#pragma unroll
        for (int i = 0; i < 9; ++i) sched[i] = 0u;
      }
    }
  };
  int* const next_slot = reinterpret_cast<int*>(smem + 2 * LN_STG);   // word 0 of wave 0's staging image (idle in the K loop)
  const int dyn_base = (dyn && dyn_first) ? range0 : range0 + per;
  int first_tile = range0 + loc;
  if (dyn && dyn_first) {
    if (tid == 0) *next_slot = range0 + (int)atomicAdd(my_ctr, 1u);
    __syncthreads();
    first_tile = __builtin_amdgcn_readfirstlane(*next_slot);
    __syncthreads();
  }
  if (first_tile >= range1) { retire(); return; }

  // ---- LDS-DMA: kernel-constant lane offsets (interior tiles), scalar tile bases; the cursor runs two steps ahead ----
  constexpr int NPW = 4;
  // The token-major B's transposing reads as inline asm (gemm.hip, kmb_tr_read_asm) wherever the library uses them (KMB_TR_ALL).
  constexpr bool TRASM = KMB_TR_ALL && !B_KC;
  uint32_t offA[NPW], offB[NPW];
  dma_offsets256w4<true, 4>(offA, p.lda, 0, 1 << 30, wave, lane);
  dma_offsets256w4<B_KC, 4>(offB, p.ldb, 0, 1 << 30, wave, lane);
  const size_t stepB = B_KC ? (size_t)BK * 2 : (size_t)BK * p.ldb * 2;
  const char *gA_d, *gB_d;
  int tile_d = first_tile, td = 0;
  int tile_next = first_tile;   // the tile after the one being multiplied: known once the cursor has left that one
  // ---- L2 touch of the activation panel (gemm_kernel_v11, "L2 prefetch of the activation operand"): one stage in flight covers
  // an L2 round trip, not a memory one, and inside a step A was just streamed out by the previous kernel.  The workgroups that
  // share a row panel (the tiles of one tm: consecutive tiles, side by side on one XCD) each touch THEIR share of its rows
  // LN_PFD steps ahead of the DMA cursor, one load instruction per wave right behind a stage's pieces; the step's counted wait leaves
  // it outstanding, so it has two steps to land.  Its result is never used and it has NO register destination (round 6; KMB_L2_TOUCH,
  // gemm.hip): a 4-byte LDS-DMA into 256 bytes of the issuing wave's own epilogue staging image, which nothing reads or writes during
  // a K loop.  Rounds 4-5 sent it to v255 on the premise that the allocator never hands v255 out -- a property of one compilation,
  // guarded by an ISA test, and false as soon as the transposing reads became inline asm -- because the cross-tile touches (the last
  // steps of a tile touch the next tile's first steps) were in flight DURING the epilogue, which owns the staging images.  Now a
  // tile's last two steps issue no touch: the two steps before them issue two each (the lines the last two would have asked for,
  // two steps earlier), every touch has landed at the last step's wait, and the waits are counted per step (tile schedule below).
  // Not for the data-gradient layout's B (the weights: L2 / Infinity Cache residents).
  constexpr int LN_PFD = 2;
#ifdef KMB_LN_NOTOUCH   // (timing builds only: the kernel without its L2 touches)
  constexpr int TCH = 0;
#else
  constexpr int TCH = 1;
#endif
  const int sharers = col_blocks ? CB : tiles_n;
  const int pf_share = (256 + sharers - 1) / sharers;
  const int pf_gs = (pf_share + 7) >> 3;                 // rows per wave
  const char *gA_tile, *gA_nx;
  int pf_rows = 0, pf_rows_nx = 0;
  auto set_dma_tile = [&](int tile) {
    int tm, tn;
    decode_tile(tile, tm, tn);
    gA_d = ln_uniform(reinterpret_cast<const char*>(p.A) + (size_t)tm * 256 * p.lda * 2);
    gB_d = ln_uniform(reinterpret_cast<const char*>(p.B) + (B_KC ? (size_t)tn * 256 * p.ldb * 2 : (size_t)tn * 256 * 2));
    gA_tile = gA_d;
    pf_rows = (col_blocks ? tn % CB : tn) * pf_share;
    const int tx = (!dyn && tile + per < range1) ? tile + per : tile;   // (dynamic hand-out: the tile after this one is not known yet)
    decode_tile(tx, tm, tn);
    gA_nx = ln_uniform(reinterpret_cast<const char*>(p.A) + (size_t)tm * 256 * p.lda * 2);
    pf_rows_nx = (col_blocks ? tn % CB : tn) * pf_share;
  };
  const unsigned touch_lds = kmb_lds_addr(smem + 2 * LN_STG + wave * LN_EPW + 2048);   // (word 0 of wave 0's image is next_slot)
  auto touch = [&](int ahead) {   // this wave's rows of the panel lines that the cursor will ask for `ahead` steps from now
    const int ps = td + ahead;
    const bool nx = ps >= nt;
    const char* sbase = ln_uniform((nx ? gA_nx : gA_tile) + (size_t)(nx ? ps - nt : ps) * (BK * 2));
    int row = wave * pf_gs + (lane < pf_gs ? lane : pf_gs - 1);
    row = row < pf_share ? row : pf_share - 1;
    row += nx ? pf_rows_nx : pf_rows;
    row = row < 256 ? row : 255;
    const uint32_t voff = (uint32_t)row * (uint32_t)p.lda * 2u;
    KMB_L2_TOUCH(voff, sbase, touch_lds);
  };
  // the cursor's stage -> `stage` (past the last tile the last tile is fetched again, never read).  The step's touches are issued BEHIND it by
  // the caller (before advance_cursor()), so that they are the youngest operations at the next step's wait.
  auto dma_stage = [&](char* stage) {
    char* da = stage + wave * 4096;
    char* db = stage + LN_A + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(gA_d, offA[i], da + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(gB_d, offB[i], db + i * 1024);
    gA_d = ln_uniform(gA_d + BK * 2);
    gB_d = ln_uniform(gB_d + stepB);
  };
  auto advance_cursor = [&]() {
    if (++td == nt) {
      td = 0;
      tile_next = dyn ? __builtin_amdgcn_readfirstlane(*next_slot) : tile_d + per;
      if (tile_next < range1) tile_d = tile_next;
      set_dma_tile(tile_d);
    }
  };

  bf16x8 fa[2][4], fb[2][NJ];
  f32x4 acc[8][NJ];
  auto read_a = [&](const char* st, int kk, int half, bf16x8 (&d)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = read_frag3<true, 256>(st, wm * 8 + half * 4 + i, kk, r, g);
  };
  // token-major B as inline-asm transposing reads: ONE address register per column tile (lane constant + the stage's LDS address), the K half
  // (kk) and the row group (hh) in the instruction's offset field -- same addresses as read_frag3<false, 256>: its swizzle term depends on
  // the lane only ((krow & 3) = (r >> 2) & 3, bit 3 of krow = g & 1), so off = (kk * 32 + hh * 4) * 512 + [lane part].  (With a separate
  // address per read hipcc hoisted 16 of them per stage out of the loop and spilled into it.)
  [[maybe_unused]] uint32_t bbase[NJ];
  if constexpr (TRASM) {
    const int krow0 = g * 8 + (r >> 2);
#pragma unroll
    for (int j = 0; j < NJ; ++j) bbase[j] = (uint32_t)(LN_A + krow0 * 512 + (((wn * NJ + j) ^ swz_nkc(krow0)) << 5) + ((r & 3) << 3));
  }
  auto read_b = [&](const char* st, auto kk_c, bf16x8 (&d)[NJ]) {
    constexpr int kk = decltype(kk_c)::value;
    if constexpr (TRASM) {
#ifndef KMB_TR_BUILTIN
      const uint32_t sb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)st;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const uint32_t a = sb + bbase[j];
        const s16x4 t0 = kmb_tr_read_asm_off<kk * 32 * 512>(a);
        const s16x4 t1 = kmb_tr_read_asm_off<(kk * 32 + 4) * 512>(a);
        d[j][0] = t0[0]; d[j][1] = t0[1]; d[j][2] = t0[2]; d[j][3] = t0[3];
        d[j][4] = t1[0]; d[j][5] = t1[1]; d[j][6] = t1[2]; d[j][7] = t1[3];
      }
#endif
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) d[j] = read_frag3<B_KC, 256, false>(st + LN_A, wn * NJ + j, kk, r, g);
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  auto mma = [&](int half, const bf16x8 (&a)[4], const bf16x8 (&b)[NJ]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[half * 4 + i][j], 0, 0, 0);   // C^T tile
  };
  constexpr int NDA = 4, NDB = B_KC ? NJ : 2 * NJ;   // ds_read instructions per 4 A / NJ B fragments
  // One 64-deep K step on stage `cur` (tools/mfma_loop.hip, variant 7):
  //   0: A(k0, rows 0-63) x B(k0)    || read A(k0, rows 64-127)
  //   1: A(k0, rows 64-127) x B(k0)  || read B(k1), A(k1, rows 0-63)
  //   2: A(k1, rows 0-63) x B(k1)    || read A(k1, rows 64-127); every piece of this wave has landed (vmcnt W: the touches behind them may be out); barrier
  //   3: A(k1, rows 64-127) x B(k1)  || read k0 of stage `nxt` (not in a tile's LAST step), fetch the cursor's stage into `cur` + NT touches
  // w: vector-memory operations the step's wait leaves outstanding (the touches issued behind the previous step's pieces); ntouch: touches this
  // step issues behind ITS pieces (0, 1 or 2) -- both wave-uniform run-time values (tile schedule below): two one-instruction branches at the
  // step's barrier and at its end, outside the scheduled sub-phases.
  auto kstep = [&](char* cur, const char* nxt, auto last_c, int w, int ntouch) {
    constexpr bool LAST = decltype(last_c)::value;
    if constexpr (TRASM) KMB_TR_SYNC();
    read_a(cur, 0, 1, fa[1]);
    mma(0, fa[0], fb[0]);
#pragma unroll
    for (int q = 0; q < NDA; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TRASM) KMB_TR_SYNC();
    read_b(cur, K1{}, fb[1]);
    read_a(cur, 1, 0, fa[0]);
    mma(1, fa[1], fb[0]);
#pragma unroll
    for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 1); __builtin_amdgcn_sched_group_barrier(0x100, (NDB + NDA + 7) / 8, 1); }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TRASM) KMB_TR_SYNC();
    read_a(cur, 1, 1, fa[1]);
    mma(0, fa[0], fb[1]);
#pragma unroll
    for (int q = 0; q < NDA; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 2); __builtin_amdgcn_sched_group_barrier(0x100, 1, 2); __builtin_amdgcn_sched_group_barrier(0x008, 2, 2); }
    __builtin_amdgcn_sched_barrier(0);
    // vmcnt(w) lgkmcnt(0): this wave's pieces of the previous step have landed, the w touches behind them may still be out
    if (LAST || w == 0) __builtin_amdgcn_s_waitcnt(0x0070);
    else if (w == 1) __builtin_amdgcn_s_waitcnt(0x0071);
    else __builtin_amdgcn_s_waitcnt(0x0072);
    __builtin_amdgcn_s_barrier();
    // (already true on every path into the barrier; restated in straight-line code so that hipcc's wait tracking, which loses it at the join of
    // the branches above, does not put its own lgkmcnt(0) in front of the first MFMA below -- behind the asm reads of the next stage)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!LAST) {
      read_b(nxt, K0{}, fb[0]);
      read_a(nxt, 0, 0, fa[0]);
    }
    dma_stage(cur);
    mma(1, fa[1], fb[1]);
    if constexpr (!LAST) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 3); __builtin_amdgcn_sched_group_barrier(0x100, (NDB + NDA + 7) / 8, 3); }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!LAST && TCH != 0) {
      if (ntouch >= 1) touch(LN_PFD);
      if (ntouch >= 2) touch(LN_PFD + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    advance_cursor();
  };
  using Yes = std::true_type;
  using No = std::false_type;
  // ---- prologue: stages 0 and 1 of the first tile (their touches: steps 2 and 3, which no in-loop touch covers for the first tile) ----
  set_dma_tile(tile_d);
  dma_stage(smem);
  if constexpr (TCH != 0) touch(LN_PFD);
  advance_cursor();
  dma_stage(smem + LN_STG);
  if constexpr (TCH != 0) touch(LN_PFD);
  advance_cursor();
  __builtin_amdgcn_s_waitcnt(TCH ? 0x0F7A : 0x0F78);   // vmcnt(10) = touch, stage 1, touch: stage 0 has landed
  __builtin_amdgcn_s_barrier();
  read_b(smem, K0{}, fb[0]);
  read_a(smem, 0, 0, fa[0]);

  float* const ef = reinterpret_cast<float*>(smem + 2 * LN_STG + wave * LN_EPW);
  int it = 0;   // linear K-step counter: stage buffer = it & 1
  uint32_t fetched = 0u;
  for (int tile = first_tile; tile < range1; tile = tile_next) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A tile's K steps and what is outstanding at each step's wait (p = this wave's 8 pieces of a stage, T = a touch; a step issues behind its barrier):
    //   t = 0            wait vmcnt(0): [p(previous tile's last step)] (first tile: [T p T] of the prologue)          issues p T
    //   t = 1 .. nt - 5  wait vmcnt(1): [T(t - 2)] p(t - 1) | T(t - 1) stays out                                    issues p T
    //   t = nt - 4       wait vmcnt(1)                                                                              issues p T T  (next tile's steps 0, 2)
    //   t = nt - 3       wait vmcnt(2): the two touches of nt - 4 stay out                                           issues p T T  (next tile's steps 1, 3)
    //   t = nt - 2       wait vmcnt(2): the two touches of nt - 3 stay out                                           issues p
    //   t = nt - 1       wait vmcnt(0): everything has landed -- NO touch is in flight when the epilogue writes the staging images   issues p
    // (nt >= 5: kmb_gemm_lean_ok.)  Dynamic hand-out (nt >= 8): the counter is read at the top of step 1 (behind step 0's touch: it is the
    // youngest operation at step 1's wait, which then covers step 0's touch too, and older than everything step 2's wait leaves out),
    // published at the top of step 3, and read by the cursor behind the barrier of step nt - 3 >= 5.
    int w = 0;
    for (int t = 0; t + 1 < nt; ++t, ++it) {
      if (dyn && t == 1 && tid == 0) fetched = atomicAdd(my_ctr, 1u);
      if (dyn && t == 3 && tid == 0) *next_slot = dyn_base + (int)fetched;
      const int ntouch = TCH == 0 ? 0 : t + 4 < nt ? 1 : t + 2 < nt ? 2 : 0;
      kstep(smem + (it & 1) * LN_STG, smem + ((it + 1) & 1) * LN_STG, No{}, w, ntouch);
      w = ntouch;
    }
    kstep(smem + (it & 1) * LN_STG, smem + ((it + 1) & 1) * LN_STG, Yes{}, 0, 0);
    ++it;
    if constexpr (TRASM) KMB_TR_SYNC();
    if (KMB_DIAG_BIT(p.tile_order, 512)) {   // epilogue ablation (diagnostic build only, tools/kloop_time.py): keep the accumulators alive
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else {
      int tm, tn;
      decode_tile(tile, tm, tn);
      const int row0w = tm * 256 + wm * 128, col0w = tn * 256 + wn * 64;
      const bool hs = col0w < p.col_scale_n, hd = p.drop_thr16 != 0u, hr = p.residual != nullptr;
#define KMB_LN_LEAN(B, S, A, R, D, C) v11_epilogue_lean<B, S, A, R, D, C, 128, false, NJ>(p, acc, ef, lane, r, g, row0w, col0w)
      if constexpr (EC == LN_BIAS) { if (hs) KMB_LN_LEAN(true, true, 0, false, false, false); else KMB_LN_LEAN(true, false, 0, false, false, false); }
      if constexpr (EC == LN_BIAS_RES) { if (hd) KMB_LN_LEAN(true, false, 0, true, true, false); else KMB_LN_LEAN(true, false, 0, true, false, false); }
      if constexpr (EC == LN_PLAIN) { if (hr) KMB_LN_LEAN(false, false, 0, true, false, false); else KMB_LN_LEAN(false, false, 0, false, false, false); }
      if constexpr (EC == LN_GELU) KMB_LN_LEAN(true, false, 1, false, false, false);
      if constexpr (EC == LN_DGELU_CS) KMB_LN_LEAN(false, false, 2, false, false, true);
      if constexpr (EC == LN_CE) KMB_LN_LEAN(true, false, 5, false, false, false);
#undef KMB_LN_LEAN
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next tile's first fragments (its stage 0 landed before the last step's barrier)
    read_b(smem + (it & 1) * LN_STG, K0{}, fb[0]);
    read_a(smem + (it & 1) * LN_STG, 0, 0, fa[0]);
  }
  if constexpr (TRASM) KMB_TR_SYNC();   // (the last tile's look-ahead fragments are never used: their registers must not be handed out before they land)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): nothing of this workgroup is in flight when it ends
  retire();
}

int ln_class(const KmbGemm& p) {
  const bool hb = p.bias != nullptr, hr = p.residual != nullptr, hd = p.drop_thr16 != 0u, hc = p.colsum != nullptr;
  const bool hs = p.col_scale_n > 0;
  if (p.act == 0 && hb && !hr && !hd && !hc) return LN_BIAS;
  if (p.act == 0 && hb && hr && !hc && !hs) return LN_BIAS_RES;
  if (p.act == 0 && !hb && !hd && !hc && !hs) return LN_PLAIN;
  if (p.act == 1 && hb && !hr && !hd && !hc && !hs) return LN_GELU;
  if (p.act == 2 && !hb && !hr && !hd && hc && !hs) return LN_DGELU_CS;
  if (p.act == 5 && hb && !hr && !hd && !hc && !hs) return LN_CE;
  return -1;
}

template <bool B_KC>
hipError_t ln_launch_layout(int ec, const KmbGemm& p, dim3 grid, hipStream_t stream, uint32_t* sched, int dyn_first) {
#define KMB_LN_CASE(E)                                                                                                          \
  case E: {                                                                                                                     \
    static bool attr = false;                                                                                                   \
    if (!attr) {                                                                                                                \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel_lean<B_KC, E>, hipFuncAttributeMaxDynamicSharedMemorySize, LN_LDS); \
      if (e != hipSuccess) return e;                                                                                            \
      attr = true;                                                                                                              \
    }                                                                                                                           \
    hipLaunchKernelGGL((gemm_kernel_lean<B_KC, E>), grid, dim3(512), LN_LDS, stream, p, sched, dyn_first);                                         \
    break;                                                                                                                      \
  }
  switch (ec) {
    KMB_LN_CASE(LN_BIAS)
    KMB_LN_CASE(LN_BIAS_RES)
    KMB_LN_CASE(LN_PLAIN)
    KMB_LN_CASE(LN_GELU)
    KMB_LN_CASE(LN_DGELU_CS)
    KMB_LN_CASE(LN_CE)
    default: return hipErrorInvalidValue;
  }
#undef KMB_LN_CASE
  return hipGetLastError();
}

}  // namespace

bool kmb_gemm_lean_ok(const KmbGemm& p) {
  if (!p.a_kc || p.split_k > 1 || (p.K % BK) != 0 || p.K / BK < 5) return false;   // >= 5 K steps: the tile schedule's peeled steps
  if ((p.M % 256) != 0 || (p.N % 256) != 0) return false;
  if ((long)(p.M / 256) * (p.N / 256) < 128) return false;
  if (p.out_bf16 == nullptr || p.out_f32 != nullptr || p.beta != 0.f) return false;
  if (p.col_scale_n > 0 && (p.col_scale_n % 64) != 0) return false;
  if (p.act == 5 && (p.row_shift == nullptr || p.row_sums == nullptr)) return false;
  if ((long)p.lda * 2 * 256 >= (1L << 31) || (long)p.ldb * 2 * 256 >= (1L << 31)) return false;   // 32-bit piece offsets
  return ln_class(p) >= 0;
}

// sched: the launch's tile counters (16 zeroed words of the caller's ring, gemm.hip v11_sched_slot) or nullptr for static tiles;
// dynamic hand-out needs K >= 8 * 64 (the counter's round trip runs under the plain steps 1 .. 3 of a tile)
hipError_t kmb_gemm_lean_launch(const KmbGemm& p, hipStream_t stream, uint32_t* sched, int dyn_first) {
  if (!kmb_gemm_lean_ok(p)) return hipErrorInvalidValue;
  if (p.K / BK < 8) sched = nullptr;
  const long tiles = (long)(p.M / 256) * (p.N / 256);
  const dim3 grid(tiles >= 256 ? 256u : (unsigned)(tiles & ~7L));
  const int ec = ln_class(p);
  return p.b_kc ? ln_launch_layout<true>(ec, p, grid, stream, sched, dyn_first) : ln_launch_layout<false>(ec, p, grid, stream, sched, dyn_first);
}
