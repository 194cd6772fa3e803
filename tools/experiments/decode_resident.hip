// EXPERIMENT (measured and lost; never in the product library): the resident decoder-layers kernel of round 5.
// Built by `python km-bart_amd/build.py --variant resident` (defines KMB_WITH_RESIDENT_DECODE for csrc/engine.cpp and links this file);
// `KMB_LIB_PATH=km-bart_amd/lib/libkmbart_hip_resident.so KMB_GEN_FUSED=2` selects it, tools/experiments/test_decode_resident.py checks it bit for bit
// against the six-launch blocks, tools/gen_resident_check.py / tools/decode_resident_stamps.py time it.  Result (profiles/r05_generation_resident_*):
// 12.0 ms per generate against 10.3 for the blocks -- an in-launch hand-off costs 3.5-4 us where a kernel boundary costs 1.7.
#define KMB_DECODE_DEVICE_ONLY
#include "decode.hip"   // device helpers of the product blocks (anonymous namespace: this translation unit gets its own copies)

namespace {

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

}  // namespace

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
