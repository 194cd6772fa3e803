// What the memory system gives the ACCESS PATTERN of the single-tile attention backward, without its arithmetic: per (batch, head)
// item five 64 x 64 bf16 tiles are read (rows of 128 bytes at the q|k|v / o / dO row strides) and three are written.  If this copy
// runs at ~6 TB/s, attn_bwd_small_kernel (3.5 TB/s of the same bytes, profiles/r04_kernel_stats_b1024_serial.md) has headroom in its
// dependent chain; if it runs at ~4, the kernel is at the roofline of its pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/attn_pattern_copy.hip -o /tmp/apc && /tmp/apc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int ITEMS_IN_FLIGHT>
__global__ __launch_bounds__(256) void pattern_copy(const uint16_t* qkv, const uint16_t* o, const uint16_t* dO, uint16_t* dqkv, int B, int H, int T,
                                                    int nitems) {
  // thread t: row t / 8 (+ 32 for the second half of the tile), 16-byte piece t % 8 of the head's 128 bytes
  const int r0 = threadIdx.x >> 3, seg = threadIdx.x & 7;
  const int d = H * 64;
  for (int it = blockIdx.x; it < nitems; it += gridDim.x * ITEMS_IN_FLIGHT) {
    u32x4 v[ITEMS_IN_FLIGHT][5][2];
#pragma unroll
    for (int k = 0; k < ITEMS_IN_FLIGHT; ++k) {
      const int item = it + k * gridDim.x;
      if (item < nitems) {
        const int b = item / H, h = item % H;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const size_t row = (size_t)b * T + r0 + 32 * half;
          v[k][0][half] = *reinterpret_cast<const u32x4*>(qkv + row * 3 * d + h * 64 + seg * 8);
          v[k][1][half] = *reinterpret_cast<const u32x4*>(qkv + row * 3 * d + d + h * 64 + seg * 8);
          v[k][2][half] = *reinterpret_cast<const u32x4*>(qkv + row * 3 * d + 2 * d + h * 64 + seg * 8);
          v[k][3][half] = *reinterpret_cast<const u32x4*>(o + row * d + h * 64 + seg * 8);
          v[k][4][half] = *reinterpret_cast<const u32x4*>(dO + row * d + h * 64 + seg * 8);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < ITEMS_IN_FLIGHT; ++k) {
      const int item = it + k * gridDim.x;
      if (item < nitems) {
        const int b = item / H, h = item % H;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const size_t row = (size_t)b * T + r0 + 32 * half;
          u32x4 a = v[k][0][half], bb = v[k][1][half], c = v[k][2][half];
          a[0] ^= v[k][3][half][0]; bb[1] ^= v[k][4][half][1]; c[2] ^= v[k][3][half][2] ^ v[k][4][half][3];
          *reinterpret_cast<u32x4*>(dqkv + row * 3 * d + h * 64 + seg * 8) = a;
          *reinterpret_cast<u32x4*>(dqkv + row * 3 * d + d + h * 64 + seg * 8) = bb;
          *reinterpret_cast<u32x4*>(dqkv + row * 3 * d + 2 * d + h * 64 + seg * 8) = c;
        }
      }
    }
  }
}

int main() {
  const int B = 1024, H = 12, T = 64, d = H * 64;
  const size_t rows = (size_t)B * T;
  uint16_t *qkv, *o, *dO, *dqkv;
  hipMalloc(&qkv, rows * 3 * d * 2); hipMalloc(&o, rows * d * 2); hipMalloc(&dO, rows * d * 2); hipMalloc(&dqkv, rows * 3 * d * 2);
  hipMemset(qkv, 1, rows * 3 * d * 2); hipMemset(o, 2, rows * d * 2); hipMemset(dO, 3, rows * d * 2);
  const double bytes = (double)rows * d * 2 * 8;   // 5 tiles read + 3 written per item
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, int grid, const char* name) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, qkv, o, dO, dqkv, B, H, T, B * H);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, qkv, o, dO, dqkv, B, H, T, B * H);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s grid %5d: %7.1f us per pass = %.2f TB/s\n", name, grid, ms * 100.0, bytes / (ms * 1e-4) / 1e12);
  };
  for (int grid : {256, 512, 768, 1024, 2048, 4096, 12288}) run(pattern_copy<1>, grid, "one item in flight");
  for (int grid : {256, 512, 768, 1024, 2048}) run(pattern_copy<2>, grid, "two items in flight");
  return 0;
}
