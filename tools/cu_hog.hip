// Test helper (tools/gemm_contention.py): `blocks` one-wave workgroups that sit on their CUs for `ticks` x 10 ns --
// a stand-in for a communication kernel that holds part of the chip while the GEMMs run.
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void hog_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int cu_hog(int blocks, unsigned long long ticks, void* stream) {
  hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, ticks);
  return (int)hipGetLastError();
}
