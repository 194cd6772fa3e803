// Ablation of the top-k part kernel's phases at the decode shape (not part of the library): hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ unsigned long long topk_key(float v, int i) {
  const uint32_t b = __float_as_uint(v);
  const uint32_t ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ((unsigned long long)ord << 32) | (uint32_t)(0x7fffffff - i);
}
template <int NV, int MODE>
__global__ __launch_bounds__(256) void part(const float* __restrict__ logits, int ldv, int V, int cpp, float* out) {
  __shared__ float sh[8];
  const int p = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* row = logits + (size_t)r * ldv;
  f32x4 x[NV];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = tid + 256 * j, i = (p * cpp + c) * 4;
    if (c < cpp && i < ldv) {
      x[j] = *reinterpret_cast<const f32x4*>(row + i);
      for (int e = 0; e < 4; ++e) { if (i + e >= V) x[j][e] = -INFINITY; m = fmaxf(m, x[j][e]); }
    } else x[j] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  }
  m = wave_max(m);
  if (lane == 0) sh[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  float sum = 0.f;
  if (MODE >= 1) {
#pragma unroll
    for (int j = 0; j < NV; ++j) for (int e = 0; e < 4; ++e) sum += __expf(x[j][e] - m);
    sum = wave_sum(sum);
  }
  unsigned long long b1 = 0, b2 = 0;
  if (MODE >= 2) {
#pragma unroll
    for (int j = 0; j < NV; ++j) for (int e = 0; e < 4; ++e) {
      const int i = (p * cpp + tid + 256 * j) * 4 + e;
      const unsigned long long key = (tid + 256 * j < cpp && i < V) ? topk_key(x[j][e], i) : 0ull;
      const unsigned long long lo = key < b1 ? key : b1;
      b1 = key > b1 ? key : b1;
      b2 = lo > b2 ? lo : b2;
    }
  }
  if (tid == 0) out[r * 4 + p] = m + sum + (float)(b1 & 0xff) + (float)(b2 & 0xff);
}
template <int MODE> float run(const float* d, float* o, int rows, int ld, int V, int cpp) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((part<13, MODE>), dim3(4, rows), dim3(256), 0, 0, d, ld, V, cpp, o);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((part<13, MODE>), dim3(4, rows), dim3(256), 0, 0, d, ld, V, cpp, o);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1e3f;
}
__global__ void fillk(float* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (float)((i * 2654435761u) & 0xffff) * 1e-4f; }
__global__ void copyk(const f32x4* a, f32x4* b, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) b[i] = a[i]; }
int main() {
  const int rows = 320, V = 50320, ld = 50432, cpp = (ld / 4 + 3) / 4;
  float *d, *o, *d2; size_t n = (size_t)rows * ld;
  hipMalloc(&d, n * 4); hipMalloc(&d2, n * 4); hipMalloc(&o, rows * 16 * 4);
  hipLaunchKernelGGL(fillk, dim3((n + 255) / 256), dim3(256), 0, 0, d, n);
  hipDeviceSynchronize();
  printf("load+max            %6.1f us\n", run<0>(d, o, rows, ld, V, cpp));
  printf("load+max+expsum     %6.1f us\n", run<1>(d, o, rows, ld, V, cpp));
  printf("load+max+expsum+key %6.1f us\n", run<2>(d, o, rows, ld, V, cpp));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(copyk, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, (const f32x4*)d, (f32x4*)d2, n / 4);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("plain copy of the 64.5 MB  %6.1f us\n", ms / 20 * 1e3f);
  return 0;
}
