// LDS-DMA ingest rate of a CU by the SHAPE of a piece (measurement only; not part of the library).
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/dma_rate tools/dma_rate.hip && gpurun_out/dma_rate
//
// A GEMM K step of a K-contiguous operand ([rows, 768] bf16, 1536-byte rows) fetches, for each of the tile's rows, the
// BK-deep slice of that row: 128 bytes per row at BK = 64 (a 1 KiB LDS-DMA piece = 8 rows x one whole 128-byte line),
// 64 bytes per row at BK = 32 (a piece = 16 rows x HALF a line; the other half is asked for one K step later).  Question:
// does the CU take in half-line pieces at the same bytes per clock?  (A 256 x 128 tile with three 32-deep stages fits two
// workgroups per CU; with 64-deep stages it does not.)
//
// Every workgroup streams 256-row panels of a 100 MB matrix through LDS with `global_load_lds_dwordx4`, two stages in
// flight, two stage buffers, optionally a workgroup barrier per step; nothing is computed.  Printed: GB/s per CU and per chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void dma_piece(const char* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// SEG: contiguous bytes per row and step (128 or 64 or 256); ROWS rows per panel; BAR: workgroup barrier per step
template <int SEG, int ROWS, bool BAR>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ A, int panels, int ld, int kbytes, int rounds, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = ROWS * SEG;            // bytes per step
  constexpr int NP = STAGE / 1024 / 4;         // pieces per wave and step
  constexpr int RPP = 1024 / SEG;              // rows per piece
  constexpr int LPR = SEG / 16;                // lanes per row
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int steps = kbytes / SEG;
  uint32_t off[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) off[i] = (uint32_t)(((wave * NP + i) * RPP + lane / LPR) * ld + (lane % LPR) * 16);
  int buf = 0;
  for (int rd = 0; rd < rounds; ++rd) {
    // panels > 0: every workgroup walks the whole matrix (HBM / Infinity Cache);  panels < 0: -panels panels PER XCD, shared by
    // the workgroups of that XCD (blockIdx % 8) and resident in its L2 -- the case of a GEMM operand panel that 12 tiles share
    const int panel = panels > 0 ? ((int)blockIdx.x + rd * (int)gridDim.x) % panels
                                 : ((int)blockIdx.x & 7) * -panels + (((int)blockIdx.x >> 3) + rd) % -panels;
    const char* base = A + (size_t)panel * ROWS * ld;
    for (int s = 0; s < steps; ++s) {
      char* dst = smem + buf * STAGE + wave * (NP * 1024);
#pragma unroll
      for (int i = 0; i < NP; ++i) dma_piece(base + off[i] + s * SEG, dst + i * 1024);
      buf ^= 1;
      // the stage issued one step ago has landed; this one stays in flight
      if (NP == 8) __builtin_amdgcn_s_waitcnt(0x0F78);
      else if (NP == 4) __builtin_amdgcn_s_waitcnt(0x0F74);
      else if (NP == 2) __builtin_amdgcn_s_waitcnt(0x0F72);
      else if (NP == 16) __builtin_amdgcn_s_waitcnt(0x4F70);   // vmcnt(16): bits 14-15 hold vmcnt[5:4]
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      if (BAR) __syncthreads();
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<uint32_t*>(smem);
}

template <int SEG, int ROWS, bool BAR>
void run(const char* name, const char* A, int M, int ld, int kbytes, int grid, int wg_lds, uint32_t* sink, bool l2 = false) {
  const int panels = l2 ? -(4 * 256 / ROWS) : M / ROWS;      // L2 case: 1.5 MB per XCD
  const int rounds = l2 ? 64 * 256 / grid : 16 * (M / ROWS) / grid;
  auto fn = stream_kernel<SEG, ROWS, BAR>;
  CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, wg_lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), wg_lds, 0, A, panels, ld, kbytes, rounds, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)grid * rounds * ROWS * kbytes;
  printf("%-3s %-58s grid %4d  LDS %3d KB  %8.1f us  %6.1f GB/s per CU  %5.2f TB/s\n", l2 ? "L2" : "mem", name, grid, wg_lds >> 10, best * 1e3,
         bytes / (best * 1e-3) / 256 / 1e9, bytes / (best * 1e-3) / 1e12);
}

int main() {
  const int M = 65536, K = 768, ld = K * 2;
  char* A;
  uint32_t* sink;
  CK(hipMalloc(&A, (size_t)M * ld));
  CK(hipMemset(A, 1, (size_t)M * ld));
  CK(hipMalloc(&sink, 4096 * 4));
  printf("# LDS-DMA streaming of a %d x %d bf16 matrix (%d MB), 256-row panels, 256 threads per workgroup, two steps in flight\n", M, K, (M * ld) >> 20);
  for (int l2 = 0; l2 < 2; ++l2) {
    const bool L = l2 != 0;
    run<128, 256, true>("128 B per row and step (BK 64), barrier per step", A, M, ld, ld, 256, 64 << 10, sink, L);
    run<128, 256, false>("128 B per row and step (BK 64), no barrier", A, M, ld, ld, 256, 64 << 10, sink, L);
    run<64, 256, true>("64 B per row and step (BK 32), barrier per step", A, M, ld, ld, 256, 32 << 10, sink, L);
    run<64, 256, false>("64 B per row and step (BK 32), no barrier", A, M, ld, ld, 256, 32 << 10, sink, L);
    run<128, 256, true>("128 B per row and step, 2 workgroups per CU, barrier", A, M, ld, ld, 512, 64 << 10, sink, L);
    run<64, 256, true>("64 B per row and step, 2 workgroups per CU, barrier", A, M, ld, ld, 512, 32 << 10, sink, L);
    run<64, 256, false>("64 B per row and step, 2 workgroups per CU, no barrier", A, M, ld, ld, 512, 32 << 10, sink, L);
    run<128, 128, true>("128 B x 128-row panels, 2 workgroups per CU, barrier", A, M, ld, ld, 512, 32 << 10, sink, L);
    run<64, 256, true>("64 B per row and step, 3 workgroups per CU, barrier", A, M, ld, ld, 768, 32 << 10, sink, L);
  }
  return 0;
}
