// Variant 10: ROLE-SPLIT persistent GEMM (forward X W^T and data-gradient dY W layouts, K-contiguous A).
//
// Why.  In the persistent kernels of gemm.hip (v11: one 4-wave workgroup per CU, all 256 accumulator registers of a wave
// live) a tile's epilogue runs with the matrix pipe idle -- 3.8 us of a 22 us K = 768 tile for a plain store, 10-15 us
// with residual / dropout / GeLU (20.5 % of the in-step GEMM time) -- and the same waves that issue the
// MFMAs also issue the stage's LDS-DMA pieces at 60-180 stalled cycles each.  One wave per SIMD cannot overlap either.
//
// Structure.  One 8-wave workgroup per CU = two GROUPS of four waves (one wave of each group per SIMD, 256 registers
// each).  A workgroup walks its tiles (256 x 128) as one sequence of rounds; in round j
//   * group (j & 1) is the MATRIX group: per 64-deep K step 24 fragment reads + 64 MFMAs per wave and nothing else
//     (wave block 128 x 64, 128 accumulator registers: the K loop of the eight-wave persistent kernel without its DMA);
//   * the other group is the MEMORY group: it issues every LDS-DMA piece of the stage two K steps ahead (12 per wave)
//     and, between them, drains ITS accumulators of tile j - 1 through the lean epilogue in eight units of 16 rows x 64
//     columns per wave, one unit per K step -- so tile j - 1's epilogue runs under tile j's MFMAs on the same SIMDs
//     (matrix pipe beside memory / VALU work of the partner wave: MI355X_MICROARCH.md "Two waves per SIMD", item 5);
//   then the groups swap.  Each K step has ONE workgroup barrier; the role branch sits INSIDE the step, so both groups
//   execute the same barrier instruction the same number of times by construction.
// LDS: three 48 KB stages (A 256 x 64, B 128 x 64: the images, swizzles and fragment reads of gemm.hip) + one 4 KB fp32
// staging image per memory-group wave = 160 KB.  Stage L + 2 is fetched right after the barrier that opens step L into the
// buffer step L - 1 read, and must have landed at the barrier that opens step L + 2: TWO stages are in flight and each has
// two full K steps to land (the memory group's wait is vmcnt(12): its twelve youngest operations -- at most the pieces
// of the newest stage -- stay outstanding).  With one stage in flight (first version: wait for the pieces at the end of
// the step that issued them) a K step lasted as long as an LDS-DMA round trip, 1.5 us, with 0.5 us of MFMAs in it.
// The matrix group's step is rotated by a quarter: the barrier sits in front of the LAST 16 MFMAs of the previous step,
// whose operands are already in registers, so the first fragment reads of the new stage hide behind them.
// Same MFMA, same k order, same epilogue arithmetic as every other variant: bit-identical results
// (tests/test_gemm_variants_gpu.py, tools/gemm_v11_check.py).
//
// Limits (kmb_gemm_rs_ok): K-contiguous A, M % 256 == 0, N % 128 == 0, K % 64 == 0 with >= 8 K steps, no split-K, bf16
// output, one of the lean epilogue classes below, >= 128 tiles.  Everything else stays on the other variants.
#define KMB_GEMM_DEVICE_ONLY
#include "gemm.hip"

namespace {

constexpr int RS_BM = 256, RS_BN = 128;
constexpr int RS_A_BYTES = RS_BM * BK * 2;              // 32 KB
constexpr int RS_STG = (RS_BM + RS_BN) * BK * 2;        // 48 KB
constexpr int RS_NSTG = 3;
constexpr int RS_EPW = 16 * 64 * 4;                     // 4 KB: 16 rows x 64 columns fp32, XOR-swizzled
constexpr int RS_LDS = RS_NSTG * RS_STG + 4 * RS_EPW;   // 160 KB

// epilogue classes (the lean classes of gemm_kernel_v11)
enum { RS_BIAS = 0, RS_BIAS_RES = 1, RS_PLAIN = 2, RS_GELU = 3, RS_DGELU_CS = 4, RS_CE = 5 };

// The epilogue of a wave's 128 x 64 block is cut into 16 MICRO-UNITS of 8 rows x 64 columns (one row-iteration of
// v11_epilogue_lean's 64-column form: CL = 8 column lanes x 8 rows), dealt evenly over the K steps of the next tile: what
// bounds the memory group is the SIMD's vector-issue slots the matrix wave's MFMAs leave free (8 of every 16 cycles: ~128
// instructions per K step), so its work has to be spread, not bunched into the first steps.  The arithmetic and its order
// are v11_epilogue_lean's.  Latencies are taken out of the stream: a 16-row chunk's accumulators are written to the wave's
// staging image one micro-unit before they are read back, the side operand (residual / GeLU' / row shift) of micro-unit
// m + 2 is requested in micro-unit m.
constexpr int RS_MICRO = 16;

// The memory role's global LOADS (bias, residual / GeLU' rows, row shifts).  Beside LDS-DMA in flight hipcc waits vmcnt(0)
// before the first use of any ordinary load's result (cdna_hip_programming.md section 5, "Three .s-level traps" (b)): every
// such wait drains both stages in flight, and the classes with a side operand run 1.9-2.5 us per K step because of it.
// The loads are therefore ORDERED so that the wait a K step has anyway would cover them -- the side operand of the
// micro-units of step s + 1 is requested in step s BEFORE that step's twelve pieces, and the memory group's vmcnt(12) at
// the top of step s + 1 leaves only its twelve youngest operations outstanding -- which lets them be inline-asm loads the
// compiler does not see (-DKMB_RS_ASMLOADS).  MEASURED AND NOT SHIPPED: the asm form faults (memory aperture violation)
// although every destination is one "+v" register web initialised before the first round; under the register pressure
// of this kernel the allocator still splits the web (v_mov of the state at loop headers), a load in flight then lands
// in a register that has been handed to another value.  Without fixed physical registers HIP source cannot express it;
// the product uses ordinary loads and pays the drains (the classes without loads -- bias, GeLU, plain -- do not).
__device__ __forceinline__ void rs_ld16(u32x4& dst, const void* ptr) {
#ifndef KMB_RS_ASMLOADS   // default: ordinary loads the compiler waits for (see the note above rs_ld16)
  dst = *reinterpret_cast<const u32x4*>(ptr);
#else
  asm volatile("global_load_dwordx4 %0, %1, off\n\ts_nop 1" : "+v"(dst) : "v"(ptr) : "memory");   // (s_nop: the compiler pads no hazard of an asm statement; its next instruction may rewrite the address registers)
#endif
}

struct RsEpiState {
  u32x4 bias_raw[2];  // eight fp32 bias values of this lane's columns
  kmb_f32x2 csum2[4];
  u32x4 side0, side1, side2, side3;        // residual / aux rows of the micro-units in flight (slot = m & 3); separate
  float shift0, shift1, shift2, shift3;    // members, not arrays: as arrays the state stayed in scratch memory
};

// the memory group's wait at the top of a K step; only the state the epilogue class actually loads is named (every named
// register is live from here on)
template <bool ALL, bool BIAS, bool SIDE, bool SHIFT>
__device__ __forceinline__ void rs_wait_loads(RsEpiState& st) {
#define KMB_RS_WAIT(...)                                                          \
  do {                                                                            \
    if (ALL) asm volatile("s_waitcnt vmcnt(0)" : __VA_ARGS__ : : "memory");       \
    else asm volatile("s_waitcnt vmcnt(12)" : __VA_ARGS__ : : "memory");          \
  } while (0)
  if constexpr (BIAS && SIDE) KMB_RS_WAIT("+v"(st.bias_raw[0]), "+v"(st.bias_raw[1]), "+v"(st.side0), "+v"(st.side1), "+v"(st.side2), "+v"(st.side3));
  else if constexpr (BIAS && SHIFT) KMB_RS_WAIT("+v"(st.bias_raw[0]), "+v"(st.bias_raw[1]), "+v"(st.shift0), "+v"(st.shift1), "+v"(st.shift2), "+v"(st.shift3));
  else if constexpr (BIAS) KMB_RS_WAIT("+v"(st.bias_raw[0]), "+v"(st.bias_raw[1]));
  else if constexpr (SIDE) KMB_RS_WAIT("+v"(st.side0), "+v"(st.side1), "+v"(st.side2), "+v"(st.side3));
  else {
    if (ALL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  }
#undef KMB_RS_WAIT
}

template <int NJ_>
__device__ __forceinline__ void rs_stage_chunk(f32x4 (&acc)[8][NJ_], float* ef, int u, int r, int g) {
  // accumulators of row chunk u -> the wave's staging image (transposed accumulators: lane (r, g) holds
  // C[16 u + r][16 j + 4 g .. + 3]); static accumulator indices in every arm
  float* const wbase = ef + r * 64;
  const int sw = (r & 7) << 3;
  auto stage = [&](const f32x4 (&a)[NJ_]) {
#pragma unroll
    for (int j = 0; j < NJ_; ++j) *reinterpret_cast<f32x4*>(wbase + ((j * 16 + g * 4) ^ sw)) = a[j];
    asm volatile("" ::: "memory");
  };
  switch (u) {
    case 0: stage(acc[0]); break;
    case 1: stage(acc[1]); break;
    case 2: stage(acc[2]); break;
    case 3: stage(acc[3]); break;
    case 4: stage(acc[4]); break;
    case 5: stage(acc[5]); break;
    case 6: stage(acc[6]); break;
    default: stage(acc[7]); break;
  }
}

// round start of a draining group: bias, zeroed column sums
template <bool BIAS>
__device__ __forceinline__ void rs_epi_begin(const KmbGemm& p, RsEpiState& st, int lane, int col0w) {
  const int gcol = col0w + (lane & 7) * 8;
  if (BIAS) {
    rs_ld16(st.bias_raw[0], p.bias + gcol);
    rs_ld16(st.bias_raw[1], p.bias + gcol + 4);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) st.csum2[e] = kmb_f32x2{0.f, 0.f};
}

// the side operand (SIDE: 1 residual, 2 aux) / row shift (act 5) of micro-unit m into slot m & 3
template <int SIDE, bool SHIFT>
__device__ __forceinline__ void rs_epi_load(const KmbGemm& p, RsEpiState& st, int m, int lane, int row0w, int col0w) {
  const int lr = lane >> 3, gcol = col0w + (lane & 7) * 8;
  const int grow = row0w + lr + 8 * m;
  // (one DISTINCT asm statement per slot: with identical statements in the arms the compiler hoists the load out of the
  //  switch and turns the four member stores into one store to a run-time indexed stack slot -- the state went to scratch)
#ifndef KMB_RS_ASMLOADS
#define KMB_RS_LD16(slot, tag) do { slot = *reinterpret_cast<const u32x4*>(ptr); asm volatile("; side slot " tag : "+v"(slot)); } while (0)
#define KMB_RS_LD4(slot, tag) do { slot = *ptr; asm volatile("; shift slot " tag : "+v"(slot)); } while (0)
#else
#define KMB_RS_LD16(slot, tag) asm volatile("global_load_dwordx4 %0, %1, off ; side slot " tag "\n\ts_nop 1" : "+v"(slot) : "v"(ptr) : "memory")
#define KMB_RS_LD4(slot, tag) asm volatile("global_load_dword %0, %1, off ; shift slot " tag "\n\ts_nop 1" : "+v"(slot) : "v"(ptr) : "memory")
#endif
  if (SIDE != 0) {
    const bf16_t* ptr = SIDE == 1 ? p.residual + (size_t)grow * p.ld_res + gcol : p.aux + (size_t)grow * p.ld_aux + gcol;
    switch (m & 3) {
      case 0: KMB_RS_LD16(st.side0, "0"); break;
      case 1: KMB_RS_LD16(st.side1, "1"); break;
      case 2: KMB_RS_LD16(st.side2, "2"); break;
      default: KMB_RS_LD16(st.side3, "3"); break;
    }
  }
  if (SHIFT) {
    const float* ptr = p.row_shift + grow;
    switch (m & 3) {
      case 0: KMB_RS_LD4(st.shift0, "0"); break;
      case 1: KMB_RS_LD4(st.shift1, "1"); break;
      case 2: KMB_RS_LD4(st.shift2, "2"); break;
      default: KMB_RS_LD4(st.shift3, "3"); break;
    }
  }
#undef KMB_RS_LD16
#undef KMB_RS_LD4
}

template <bool BIAS, bool SCALE, int ACT, bool RES, bool DROP, bool CS>
__device__ __forceinline__ void rs_epi_micro(const KmbGemm& p, RsEpiState& st, f32x4 (&acc)[8][4], float* ef, int m, int lane,
                                             int r, int g, int row0w, int col0w) {
  constexpr int LDE = 64;
  const int u = m >> 1, it = m & 1;
  const int lr = lane >> 3, c8 = (lane & 7) * 8, gcol = col0w + c8;
  const int row = lr + 8 * it;                                   // row inside the staged 16-row chunk
  const float* rd = ef + row * LDE + (c8 ^ ((row & 7) << 3));
  const f32x4 lo = *reinterpret_cast<const f32x4*>(rd);
  const f32x4 hi = *reinterpret_cast<const f32x4*>(rd + 4);
  asm volatile("" ::: "memory");
  // the next chunk's accumulators into the same image: the LDS executes a wave's accesses in order, so the writes queue
  // behind the reads above and have landed long before the next micro-unit reads them
  if (it == 1 && u + 1 < 8) rs_stage_chunk<4>(acc, ef, u + 1, r, g);
  u32x4 s0 = {0u, 0u, 0u, 0u};
  float h0 = 0.f;
  // (distinct asm statements in the arms again: a plain select over the four members became a run-time indexed stack load)
  if constexpr (RES || ACT == 2) {
    switch (m & 3) {
      case 0: asm volatile("; use side slot 0" : "+v"(st.side0)); s0 = st.side0; break;
      case 1: asm volatile("; use side slot 1" : "+v"(st.side1)); s0 = st.side1; break;
      case 2: asm volatile("; use side slot 2" : "+v"(st.side2)); s0 = st.side2; break;
      default: asm volatile("; use side slot 3" : "+v"(st.side3)); s0 = st.side3; break;
    }
  }
  if constexpr (ACT == 5) {
    switch (m & 3) {
      case 0: asm volatile("; use shift slot 0" : "+v"(st.shift0)); h0 = st.shift0; break;
      case 1: asm volatile("; use shift slot 1" : "+v"(st.shift1)); h0 = st.shift1; break;
      case 2: asm volatile("; use shift slot 2" : "+v"(st.shift2)); h0 = st.shift2; break;
      default: asm volatile("; use shift slot 3" : "+v"(st.shift3)); h0 = st.shift3; break;
    }
  }
  const kmb_f32x2 scale2 = {p.col_scale, p.col_scale};
  const kmb_f32x2 dscale2 = {p.drop_scale, p.drop_scale};
  const size_t roff = (size_t)(8 * m);
  const int grow = row0w + lr + 8 * m;
  bf16_t* const out = p.out_bf16 + (size_t)(row0w + lr) * p.ld_out_bf16 + gcol;
  kmb_f32x2 v[4] = {{lo[0], lo[1]}, {lo[2], lo[3]}, {hi[0], hi[1]}, {hi[2], hi[3]}};
  if (BIAS) {
    const kmb_f32x2 b0 = {__uint_as_float(st.bias_raw[0][0]), __uint_as_float(st.bias_raw[0][1])};
    const kmb_f32x2 b1 = {__uint_as_float(st.bias_raw[0][2]), __uint_as_float(st.bias_raw[0][3])};
    const kmb_f32x2 b2 = {__uint_as_float(st.bias_raw[1][0]), __uint_as_float(st.bias_raw[1][1])};
    const kmb_f32x2 b3 = {__uint_as_float(st.bias_raw[1][2]), __uint_as_float(st.bias_raw[1][3])};
    v[0] = v[0] + b0; v[1] = v[1] + b1; v[2] = v[2] + b2; v[3] = v[3] + b3;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (SCALE) v[e] = v[e] * scale2;
  if constexpr (ACT == 5) {
    const kmb_f32x2 c2 = {h0, h0};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] - c2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const kmb_f32x2 t = v[e] * 1.4426950408889634f;
      v[e] = kmb_f32x2{__builtin_amdgcn_exp2f(fminf(t[0], 115.f)), __builtin_amdgcn_exp2f(fminf(t[1], 115.f))};
    }
    float sum = (v[0][0] + v[0][1]) + (v[1][0] + v[1][1]) + ((v[2][0] + v[2][1]) + (v[3][0] + v[3][1]));
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o);
    if ((lane & 7) == 0) p.row_sums[(size_t)grow * p.row_sums_ld + (col0w >> 6)] = sum;
  } else if (ACT == 1) {
    bf16_t* const pre = p.preact != nullptr ? p.preact + (size_t)(row0w + lr) * p.ld_preact + gcol : nullptr;
    if (pre != nullptr) {
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
    float uu[8];
    unpack8(s0, uu);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] * kmb_f32x2{uu[2 * e], uu[2 * e + 1]};
  }
  if (DROP) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const kmb_f32x2 kept = v[e] * dscale2;
      v[e][0] = drop_keep(p.drop_seed, (uint32_t)grow, (uint32_t)(gcol + 2 * e), p.drop_thr16) ? kept[0] : 0.f;
      v[e][1] = drop_keep(p.drop_seed, (uint32_t)grow, (uint32_t)(gcol + 2 * e + 1), p.drop_thr16) ? kept[1] : 0.f;
    }
  }
  if (RES) {
    float rr[8];
    unpack8(s0, rr);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] + kmb_f32x2{rr[2 * e], rr[2 * e + 1]};
  }
  if (CS) {
#pragma unroll
    for (int e = 0; e < 4; ++e) st.csum2[e] = st.csum2[e] + v[e];
  }
  const u32x4 pk = {pack2bf(v[0][0], v[0][1]), pack2bf(v[1][0], v[1][1]), pack2bf(v[2][0], v[2][1]), pack2bf(v[3][0], v[3][1])};
  KMB_NT_STORE(pk, reinterpret_cast<u32x4*>(out + roff * p.ld_out_bf16));
  if (CS && m == RS_MICRO - 1) {
    // column sums over this wave's 128 rows: fold the row-lanes; one partial row per 64 rows of C (first filled, second zeroed)
    float csum[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { csum[2 * e] = st.csum2[e][0]; csum[2 * e + 1] = st.csum2[e][1]; }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      csum[e] += __shfl_xor(csum[e], 8);
      csum[e] += __shfl_xor(csum[e], 16);
      csum[e] += __shfl_xor(csum[e], 32);
    }
    if (lane < 8) {
      const int prow = row0w >> 6;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        p.colsum[(size_t)prow * p.N + gcol + e] = csum[e];
        p.colsum[(size_t)(prow + 1) * p.N + gcol + e] = 0.f;
      }
    }
  }
}

template <bool B_KC, int EC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_kernel_rs(const KmbGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;           // group, wave within the group
  const int wm = wq >> 1, wn = wq & 1;                // 2 x 2 wave blocks of 128 x 64
  const int r = lane & 15, g = lane >> 4;
  constexpr int NJ = 4;
  const int tiles_n = p.N / RS_BN, tiles_m = p.M / RS_BM, ntiles = tiles_m * tiles_n;
  // tile enumeration: row-major, or column-block-major for wide outputs (tile_order bit 3; see gemm_kernel_v11)
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
  // this workgroup's tiles: XCD x owns a contiguous range, its workgroups take every (grid / 8)-th tile of it
  const int per = (int)gridDim.x >> 3;
  const int xcd = (int)blockIdx.x & 7, loc = (int)blockIdx.x >> 3;
  const int tq = ntiles >> 3, trem = ntiles & 7;
  const int range0 = xcd < trem ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq;
  const int range1 = range0 + tq + (xcd < trem ? 1 : 0);
  const int first = range0 + loc;
  if (first >= range1) return;                                  // (all eight waves: no barrier has been executed)
  const int nrounds = (range1 - first + per - 1) / per;         // tiles of this workgroup: first + k * per
  const int nt = p.K / BK;                                      // >= 8 (launcher)

  const size_t stepA = (size_t)BK * 2;
  const size_t stepB = B_KC ? (size_t)BK * 2 : (size_t)BK * p.ldb * 2;
  auto uniform_ptr = [](const char* ptr) {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };

  // ---- memory role: DMA cursor (tile, K step) of the next stage to fetch ----
  // Per-lane source offsets of a wave's LDS-DMA pieces (the images and swizzles of dma_offsets256w4 / dma_offsets in
  // gemm.hip; tiles are whole here, so nothing is clamped): piece i of wave wq covers rows (wq * 8 + i) * 8 + (lane >> 3)
  // of a K-contiguous image, and the XOR swizzle of a row depends only on its low four bits -- so the lane part of the
  // offset takes TWO values (even / odd piece) and the rest, (wq * 64 + i * 8) rows, is a scalar added to the piece's base.
  // Token-major B image (data-gradient layout): piece i covers k-rows (wq * 4 + i) * 4 + (lane >> 4); the swizzle depends on
  // k-row bits 0, 1 and 3: two values again (i < 2 / i >= 2).  Four VGPRs for the whole kernel instead of twelve per tile.
  uint32_t voffA[2], voffB[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int rowl = lane >> 3;
    const int cA = (lane & 7) ^ (((rowl + 8 * f) >> 1) & 7);
    voffA[f] = (uint32_t)((rowl * p.lda + cA * 8) * 2);
    if (B_KC) {
      voffB[f] = (uint32_t)((rowl * p.ldb + cA * 8) * 2);
    } else {
      const int krl = lane >> 4, ps = lane & 15;
      const int c32 = (ps >> 1) ^ ((krl & 3) | (f << 2));
      voffB[f] = (uint32_t)((krl * p.ldb + c32 * 16 + (ps & 1) * 8) * 2);
    }
  }
  const size_t pieceA = (size_t)8 * p.lda * 2;                        // bytes between two pieces of A (8 rows)
  const size_t pieceB = B_KC ? (size_t)8 * p.ldb * 2 : (size_t)4 * p.ldb * 2;   // ... of B (8 rows / 4 k-rows)
  const char *gA_d = nullptr, *gB_d = nullptr;                         // this wave's first piece of the stage to fetch
  auto set_dma_tile = [&](int tile) {
    int tm, tn;
    decode_tile(tile, tm, tn);
    const int row0 = tm * RS_BM, col0 = tn * RS_BN;
    gA_d = uniform_ptr(reinterpret_cast<const char*>(p.A) + ((size_t)row0 + wq * 64) * p.lda * 2);
    gB_d = uniform_ptr(reinterpret_cast<const char*>(p.B) +
                       (B_KC ? ((size_t)col0 + wq * 32) * p.ldb * 2 : (size_t)col0 * 2 + (size_t)(wq * 16) * p.ldb * 2));
  };
  char* const dstA = smem + wq * (8 * 1024);
  char* const dstB = smem + RS_A_BYTES + wq * (4 * 1024);
  auto dma_stage = [&](int buf) {
    char* da = dstA + buf * RS_STG;
    char* db = dstB + buf * RS_STG;
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(uniform_ptr(gA_d + (size_t)i * pieceA), voffA[i & 1], da + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(uniform_ptr(gB_d + (size_t)i * pieceB), voffB[B_KC ? (i & 1) : (i >> 1)], db + i * 1024);
    gA_d = uniform_ptr(gA_d + stepA);
    gB_d = uniform_ptr(gB_d + stepB);
  };

  // ---- matrix role: fragment reads rebuilt per use from one lane constant per operand (gemm_kernel_v11, REMAT) ----
  const int rm_ka = r * 128 + ((g ^ ((r >> 1) & 7)) << 4);
  const int rm_sw = ((r >> 2) & 3) | ((g & 1) << 2);
  const int rm_l = (g * 8 + (r >> 2)) * (RS_BN * 2) + ((r & 3) << 3);
  const int rm_nb = rm_l | (rm_sw << 5);
  auto tr_read = [&](const char* ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
  };
  auto read_a = [&](const char* stage, int kk, int half, bf16x8 (&dst)[4]) {
    int c = rm_ka;
    asm volatile("" : "+v"(c));
    const char* base = stage + (c ^ (kk << 6)) + (wm * 8 + half * 4) * 2048;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(base + i * 2048);
  };
  auto read_b = [&](const char* stage, int kk, bf16x8 (&dst)[NJ]) {
    if constexpr (B_KC) {
      int c = rm_ka;
      asm volatile("" : "+v"(c));
      const char* base = stage + RS_A_BYTES + (c ^ (kk << 6)) + (wn * NJ) * 2048;
#pragma unroll
      for (int j = 0; j < NJ; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(base + j * 2048);
    } else {
      int c = rm_nb;
      asm volatile("" : "+v"(c));
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const char* pj = stage + RS_A_BYTES + kk * (32 * RS_BN * 2) + (c ^ ((wn * NJ + j) << 5));
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const s16x4 t = tr_read(pj + hh * (4 * RS_BN * 2));
          dst[j][hh * 4 + 0] = t[0]; dst[j][hh * 4 + 1] = t[1]; dst[j][hh * 4 + 2] = t[2]; dst[j][hh * 4 + 3] = t[3];
        }
      }
    }
  };
  constexpr int NDA = 4;                      // ds_read instructions per 4 A fragments
  constexpr int NDB = B_KC ? NJ : 2 * NJ;     // ... per NJ B fragments
  f32x4 acc[8][NJ];
  auto mma = [&](int half, const bf16x8 (&a)[4], const bf16x8 (&b)[NJ]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[half * 4 + i][j], 0, 0, 0);  // C^T tile
  };

  // ---- prologue: group 1 (the memory group of round 0) fetches stages 0 and 1 of the first tile ----
  if (grp == 1) {
    set_dma_tile(first);
    dma_stage(0);
    dma_stage(1);
  }

  float* const ef = reinterpret_cast<float*>(smem + RS_NSTG * RS_STG + wq * RS_EPW);
  RsEpiState est;
  est.bias_raw[0] = est.bias_raw[1] = est.side0 = est.side1 = est.side2 = est.side3 = u32x4{0u, 0u, 0u, 0u};
  est.shift0 = est.shift1 = est.shift2 = est.shift3 = 0.f;
  int row0w = 0, col0w = 0;          // the wave block this wave multiplied last (drained in its next memory round)
  constexpr int SIDE_KIND = EC == RS_DGELU_CS ? 2 : (EC == RS_BIAS_RES || EC == RS_PLAIN) ? 1 : 0;
  auto load_side = [&](int m) {
    if constexpr (EC == RS_PLAIN) { if (p.residual != nullptr) rs_epi_load<1, false>(p, est, m, lane, row0w, col0w); }
    else if constexpr (SIDE_KIND != 0 || EC == RS_CE) rs_epi_load<SIDE_KIND, EC == RS_CE>(p, est, m, lane, row0w, col0w);
  };
  int buf = 0;                       // stage buffer of the current K step: L % 3
  // One extra round drains the last tile; its K steps carry no MFMAs and no fetches, only the epilogue units.
  for (int round = 0; round <= nrounds; ++round) {
    const int tile = first + round * per;                        // the tile being multiplied in this round (if any)
    const bool mfma_role = (round & 1) == grp && round < nrounds;
    const bool drain = round >= 1 && ((round - 1) & 1) == grp;   // this group multiplied tile round - 1
    const int steps = round < nrounds ? nt : (nt < RS_MICRO ? nt : RS_MICRO);   // the drain round only carries micro-units
    int em = 0, eacc = 0;                                         // next micro-unit; Bresenham counter dealing 16 of them over `steps`
    int lm = 0, lacc = 0;                                         // the same schedule one step ahead: side operands to request
    int kd = 2;                                                  // K step (of `tile`) the next fetch asks for
    bool fetch_ok = round < nrounds;
    bool issued_prev = round == 0 && grp == 1;                   // a stage was requested in the previous step (the prologue's stage 1)
    if (mfma_role) {
      int tm, tn;
      decode_tile(tile, tm, tn);
      row0w = tm * RS_BM + wm * 128; col0w = tn * RS_BN + wn * 64;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      if (round < nrounds) {
        set_dma_tile(tile);
        gA_d = uniform_ptr(gA_d + 2 * stepA);
        gB_d = uniform_ptr(gB_d + 2 * stepB);
      }
      if (drain) {
        // bias (and zeroed column sums) of the block this wave drains, and the side operands of step 0's micro-units: the
        // round's first wait (vmcnt(0): nothing younger was requested) covers them
        rs_epi_begin<EC != RS_PLAIN && EC != RS_DGELU_CS>(p, est, lane, col0w);
        lacc = RS_MICRO;
        while (lacc >= steps && lm < RS_MICRO) { lacc -= steps; load_side(lm); ++lm; }
      }
    }
    // Two loops, one per role, each with `steps` iterations of ONE workgroup barrier (the hardware does not care which
    // s_barrier instruction a wave executes): written apart so that the register allocation of one role does not carry the
    // other's state (in one loop with the role test inside, the matrix role's fragment registers stayed live through the
    // memory role's code and the epilogue spilled -- scratch accesses are vector-memory operations that wait behind the
    // LDS-DMA pieces in flight).
    if (mfma_role) {
#ifdef KMB_RS_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      bf16x8 fa[2][4], fb[2][NJ];   // (declared here: the fragments must not be live through the other role's loop)
      for (int s = 0; s < steps; ++s) {
        // the barrier that opens K step L: this group is done reading step L - 1's buffer (fragments in registers) and
        // stage L has landed.  A group that has just turned matrix group issued the first two stages of its tile itself,
        // at the end of its memory round: in steps 0 and 1 it is still the one that has to wait for them.
        if (s == 0) __builtin_amdgcn_s_waitcnt(0x007C);         // lgkmcnt(0) vmcnt(12): stage 0 of the tile
        else if (s == 1) __builtin_amdgcn_s_waitcnt(0x0070);    // lgkmcnt(0) vmcnt(0): stage 1
        else __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* cur = smem + buf * RS_STG;
        // sub-phase 3 of the PREVIOUS step: A(k1, rows 64-127) x B(k1), operands in registers  ||  read B(k0), A(k0, rows 0-63)
        // of this step's stage (its latency hides behind these MFMAs)
        read_b(cur, 0, fb[0]);
        read_a(cur, 0, 0, fa[0]);
        if (s > 0) {
          mma(1, fa[1], fb[1]);
          __builtin_amdgcn_sched_group_barrier(0x100, (NDB + 1) / 2, 3);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 3);
          __builtin_amdgcn_sched_group_barrier(0x100, NDB / 2, 3);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 3);
          __builtin_amdgcn_sched_group_barrier(0x100, NDA, 3);
          __builtin_amdgcn_sched_group_barrier(0x008, 8, 3);
        }
        __builtin_amdgcn_sched_barrier(0);
        // sub-phase 0: A(k0, rows 0-63) x B(k0)  ||  read A(k0, rows 64-127)
        read_a(cur, 0, 1, fa[1]);
        mma(0, fa[0], fb[0]);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NDA / 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NDA / 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
        // sub-phase 1: A(k0, rows 64-127) x B(k0)  ||  read B(k1), A(k1, rows 0-63)
        read_b(cur, 1, fb[1]);
        read_a(cur, 1, 0, fa[0]);
        mma(1, fa[1], fb[0]);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, (NDB + 1) / 2, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, NDB / 2, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, NDA, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
        __builtin_amdgcn_sched_barrier(0);
        // sub-phase 2: A(k1, rows 0-63) x B(k1)  ||  read A(k1, rows 64-127) (used behind the next barrier)
        read_a(cur, 1, 1, fa[1]);
        mma(0, fa[0], fb[1]);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 2);
        __builtin_amdgcn_sched_group_barrier(0x100, NDA / 2, 2);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 2);
        __builtin_amdgcn_sched_group_barrier(0x100, NDA / 2, 2);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 2);
        __builtin_amdgcn_sched_barrier(0);
        buf = buf == RS_NSTG - 1 ? 0 : buf + 1;
      }
      // sub-phase 3 of the tile's last K step (operands in registers); every LDS read of this group is complete before it
      // reaches the next round's first barrier
      __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0)
      mma(1, fa[1], fb[1]);
    } else {
#ifdef KMB_RS_PRIO
      __builtin_amdgcn_s_setprio(KMB_RS_PRIO);   // the memory wave's few vector instructions first: one MFMA every 16 cycles leaves room
#endif
      for (int s = 0; s < steps; ++s) {
        // the barrier that opens K step L: this group's pieces of stage L have landed (its twelve youngest operations may
        // still be in flight: they are at most the pieces of stage L + 1)
        // (the wait statements name the epilogue state: the inline-asm loads that fill it are older than the twelve youngest)
        constexpr bool W_BIAS = EC != RS_PLAIN && EC != RS_DGELU_CS, W_SIDE = SIDE_KIND != 0, W_SHIFT = EC == RS_CE;
        if (issued_prev) rs_wait_loads<false, W_BIAS, W_SIDE, W_SHIFT>(est);   // vmcnt(12)
        else rs_wait_loads<true, W_BIAS, W_SIDE, W_SHIFT>(est);                // vmcnt(0): no younger stage was requested (round start, end of the tile list)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nb1 = buf == RS_NSTG - 1 ? 0 : buf + 1;      // (L + 1) % 3
        const int nb2 = nb1 == RS_NSTG - 1 ? 0 : nb1 + 1;      // (L + 2) % 3
        // the first 16-row chunk into the wave's staging image -- behind the round's first barrier: the partner wave of the
        // other group shares this image and was reading it until it reached that barrier
        if (drain && s == 0) rs_stage_chunk<4>(acc, ef, 0, r, g);
        // ---- side operands of the NEXT step's micro-units, requested before this step's pieces (see rs_ld16) ----
        if (drain) {
          lacc += RS_MICRO;
          while (lacc >= steps && lm < RS_MICRO) { lacc -= steps; load_side(lm); ++lm; }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- fetch stage L + 2 into the buffer step L - 1 read ----
        if (fetch_ok) {
          if (kd == nt) {   // this tile's K steps are all requested: on to the next tile of the workgroup
            kd = 0;
            fetch_ok = round + 1 < nrounds;
            if (fetch_ok) set_dma_tile(tile + per);
          }
          if (fetch_ok) {
#ifndef KMB_RS_NODMA   // (timing experiments only: -DKMB_RS_NODMA / -DKMB_RS_NOEPI builds compute garbage)
            dma_stage(nb2);
#endif
            ++kd;
          }
        }
        issued_prev = fetch_ok;
        __builtin_amdgcn_sched_barrier(0);
        // ---- this step's share of the 16 micro-units of the tile this group multiplied in the previous round ----
        eacc += RS_MICRO;
#ifdef KMB_RS_NOEPI
        if (drain && s == 0) {   // keep the accumulators (and with them the MFMAs) alive
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(acc[i][j]));
        }
        while (false) {
#else
        while (drain && eacc >= steps && em < RS_MICRO) {
#endif
          eacc -= steps;
          const bool hs = col0w < p.col_scale_n, hd = p.drop_thr16 != 0u, hr = p.residual != nullptr;
#define KMB_RS_UNIT(B, S, A, R, D, C) rs_epi_micro<B, S, A, R, D, C>(p, est, acc, ef, em, lane, r, g, row0w, col0w)
          if constexpr (EC == RS_BIAS) { if (hs) KMB_RS_UNIT(true, true, 0, false, false, false); else KMB_RS_UNIT(true, false, 0, false, false, false); }
          if constexpr (EC == RS_BIAS_RES) { if (hd) KMB_RS_UNIT(true, false, 0, true, true, false); else KMB_RS_UNIT(true, false, 0, true, false, false); }
          if constexpr (EC == RS_PLAIN) { if (hr) KMB_RS_UNIT(false, false, 0, true, false, false); else KMB_RS_UNIT(false, false, 0, false, false, false); }
          if constexpr (EC == RS_GELU) KMB_RS_UNIT(true, false, 1, false, false, false);
          if constexpr (EC == RS_DGELU_CS) KMB_RS_UNIT(false, false, 2, false, false, true);
          if constexpr (EC == RS_CE) KMB_RS_UNIT(true, false, 5, false, false, false);
#undef KMB_RS_UNIT
          ++em;
        }
        __builtin_amdgcn_sched_barrier(0);
        buf = nb1;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): nothing of this workgroup is in flight when it ends
}

int rs_class(const KmbGemm& p) {
  const bool hb = p.bias != nullptr, hr = p.residual != nullptr, hd = p.drop_thr16 != 0u, hc = p.colsum != nullptr;
  const bool hs = p.col_scale_n > 0;
  if (p.act == 0 && hb && !hr && !hd && !hc) return RS_BIAS;
  if (p.act == 0 && hb && hr && !hc && !hs) return RS_BIAS_RES;
  if (p.act == 0 && !hb && !hd && !hc && !hs) return RS_PLAIN;
  if (p.act == 1 && hb && !hr && !hd && !hc && !hs) return RS_GELU;
  if (p.act == 2 && !hb && !hr && !hd && hc && !hs) return RS_DGELU_CS;
  if (p.act == 5 && hb && !hr && !hd && !hc && !hs) return RS_CE;
  return -1;
}

template <bool B_KC>
hipError_t rs_launch_layout(int ec, const KmbGemm& p, dim3 grid, hipStream_t stream) {
#define KMB_RS_CASE(E)                                                                                                         \
  case E: {                                                                                                                    \
    static bool attr = false;                                                                                                  \
    if (!attr) {                                                                                                               \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel_rs<B_KC, E>, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS); \
      if (e != hipSuccess) return e;                                                                                           \
      attr = true;                                                                                                             \
    }                                                                                                                          \
    hipLaunchKernelGGL((gemm_kernel_rs<B_KC, E>), grid, dim3(512), RS_LDS, stream, p);                                          \
    break;                                                                                                                     \
  }
  switch (ec) {
    KMB_RS_CASE(RS_BIAS)
    KMB_RS_CASE(RS_BIAS_RES)
    KMB_RS_CASE(RS_PLAIN)
    KMB_RS_CASE(RS_GELU)
    KMB_RS_CASE(RS_DGELU_CS)
    KMB_RS_CASE(RS_CE)
    default: return hipErrorInvalidValue;
  }
#undef KMB_RS_CASE
  return hipGetLastError();
}

}  // namespace

bool kmb_gemm_rs_ok(const KmbGemm& p) {
  if (!p.a_kc || p.split_k > 1 || (p.K % BK) != 0 || p.K / BK < 8) return false;
  if ((p.M % RS_BM) != 0 || (p.N % RS_BN) != 0) return false;
  if ((long)(p.M / RS_BM) * (p.N / RS_BN) < 128) return false;
  if (p.out_bf16 == nullptr || p.out_f32 != nullptr || p.beta != 0.f) return false;
  if (p.col_scale_n > 0 && (p.col_scale_n % 64) != 0) return false;
  if (p.act == 5 && (p.row_shift == nullptr || p.row_sums == nullptr)) return false;
  return rs_class(p) >= 0;
}

hipError_t kmb_gemm_rs_launch(const KmbGemm& p, hipStream_t stream) {
  if (!kmb_gemm_rs_ok(p)) return hipErrorInvalidValue;
  const long tiles = (long)(p.M / RS_BM) * (p.N / RS_BN);
  const dim3 grid(tiles >= 256 ? 256u : (unsigned)(tiles & ~7L));
  const int ec = rs_class(p);
  return p.b_kc ? rs_launch_layout<true>(ec, p, grid, stream) : rs_launch_layout<false>(ec, p, grid, stream);
}
