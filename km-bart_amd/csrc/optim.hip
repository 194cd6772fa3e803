// Fused multi-tensor AdamW (transformers 3.0.2 form, reference vcg_train.py:100) over the flat
// parameter arena, plus small HBM-bound movers (casts, fills, KV-cache append, row gather).
#include "common.h"
#include "kernels.h"
#include "diag.h"

namespace {

// 28 B/param: read p,g,m,v (16) + write p,m,v (12) + 2 B bf16 mirror
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ pb, size_t n4, size_t n, float lr,
                                                    float b1, float b2, float omb1, float omb2, float eps, float wd,
                                                    float step_size, float gscale) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gg[e] * gscale;
      mm[e] = mm[e] * b1 + gr * omb1;
      vv[e] = vv[e] * b2 + gr * gr * omb2;
      pp[e] = pp[e] - step_size * (mm[e] / (sqrtf(vv[e]) + eps));
      if (wd > 0.f) pp[e] = pp[e] - lr * wd * pp[e];
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (pb != nullptr) {
      uint2 o;
      o.x = pack2bf(pp[0], pp[1]);
      o.y = pack2bf(pp[2], pp[3]);
      reinterpret_cast<uint2*>(pb)[i] = o;
    }
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0) {
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      const float gr = g[i] * gscale;
      const float mm = m[i] * b1 + gr * omb1;
      const float vv = v[i] * b2 + gr * gr * omb2;
      float pp = p[i] - step_size * (mm / (sqrtf(vv) + eps));
      if (wd > 0.f) pp = pp - lr * wd * pp;
      p[i] = pp; m[i] = mm; v[i] = vv;
      if (pb != nullptr) pb[i] = f2bf(pp);
    }
  }
}

// The same update with U float4 per thread and loop turn in flight (4 U loads requested before the first is used) so that a
// SMALL grid -- one or two workgroups per CU, which leaves the CU's other wave slots to the backward pass this update runs
// beside -- still keeps the memory system busy.  Element-wise identical arithmetic.
template <int U>
__global__ __launch_bounds__(256) void adamw_kernel_u(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      bf16_t* __restrict__ pb, size_t n4, size_t n, float lr,
                                                      float b1, float b2, float omb1, float omb2, float eps, float wd,
                                                      float step_size, float gscale) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * U) {
    f32x4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride < n4 ? i0 + u * stride : i0;   // clamped address, no branch around the load
      pp[u] = reinterpret_cast<f32x4*>(p)[i];
      gg[u] = reinterpret_cast<const f32x4*>(g)[i];
      mm[u] = reinterpret_cast<f32x4*>(m)[i];
      vv[u] = reinterpret_cast<f32x4*>(v)[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      if (i >= n4) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gr = gg[u][e] * gscale;
        mm[u][e] = mm[u][e] * b1 + gr * omb1;
        vv[u][e] = vv[u][e] * b2 + gr * gr * omb2;
        pp[u][e] = pp[u][e] - step_size * (mm[u][e] / (sqrtf(vv[u][e]) + eps));
        if (wd > 0.f) pp[u][e] = pp[u][e] - lr * wd * pp[u][e];
      }
      reinterpret_cast<f32x4*>(p)[i] = pp[u];
      reinterpret_cast<f32x4*>(m)[i] = mm[u];
      reinterpret_cast<f32x4*>(v)[i] = vv[u];
      if (pb != nullptr) {
        uint2 o;
        o.x = pack2bf(pp[u][0], pp[u][1]);
        o.y = pack2bf(pp[u][2], pp[u][3]);
        reinterpret_cast<uint2*>(pb)[i] = o;
      }
    }
  }
  if (blockIdx.x == 0) {
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      const float gr = g[i] * gscale;
      const float mm1 = m[i] * b1 + gr * omb1;
      const float vv1 = v[i] * b2 + gr * gr * omb2;
      float pp1 = p[i] - step_size * (mm1 / (sqrtf(vv1) + eps));
      if (wd > 0.f) pp1 = pp1 - lr * wd * pp1;
      p[i] = pp1; m[i] = mm1; v[i] = vv1;
      if (pb != nullptr) pb[i] = f2bf(pp1);
    }
  }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[i];
    uint2 o;
    o.x = pack2bf(a[0], a[1]);
    o.y = pack2bf(a[2], a[3]);
    reinterpret_cast<uint2*>(y)[i] = o;
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) y[i] = f2bf(x[i]);
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ x, float v, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] = v;
}

__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ x, int ldx, bf16_t* __restrict__ y,
                                                        int ldy, int rows, int cols) {
  const size_t total = (size_t)rows * ldy;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / ldy), c = (int)(i % ldy);
    y[i] = c < cols ? f2bf(x[(size_t)r * ldx + c]) : (bf16_t)0;
  }
}

// cache[(r*Tmax + t)*HD + c] = src[r*ld + c]
__global__ __launch_bounds__(256) void kv_append_kernel(const bf16_t* __restrict__ src, int ld_src,
                                                        bf16_t* __restrict__ cache, int Tmax, int HD, int t, int R) {
  const int chunks = HD >> 3;
  const size_t total = (size_t)R * chunks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / chunks), c = (int)(i % chunks);
    *reinterpret_cast<u32x4*>(cache + ((size_t)r * Tmax + t) * HD + c * 8) =
        *reinterpret_cast<const u32x4*>(src + (size_t)r * ld_src + c * 8);
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ src, const int32_t* __restrict__ idx,
                                                          char* __restrict__ dst, int rows, int row_bytes,
                                                          size_t stride) {
  const int chunks = row_bytes >> 4;
  const size_t total = (size_t)rows * chunks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / chunks), c = (int)(i % chunks);
    *reinterpret_cast<u32x4*>(dst + (size_t)r * stride + c * 16) =
        *reinterpret_cast<const u32x4*>(src + (size_t)idx[r] * stride + c * 16);
  }
}

// the same row gather over up to 16 (src, dst) buffer pairs in ONE launch (beam reorder of every layer's K and V cache)
struct GatherSet { const char* src[16]; char* dst[16]; };
__global__ __launch_bounds__(256) void gather_rows_multi_kernel(GatherSet gs, const int32_t* __restrict__ idx, int rows,
                                                                int row_bytes, size_t stride) {
  const char* __restrict__ src = gs.src[blockIdx.y];
  char* __restrict__ dst = gs.dst[blockIdx.y];
  const int chunks = row_bytes >> 4;
  const size_t total = (size_t)rows * chunks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / chunks), c = (int)(i % chunks);
    *reinterpret_cast<u32x4*>(dst + (size_t)r * stride + c * 16) =
        *reinterpret_cast<const u32x4*>(src + (size_t)idx[r] * stride + c * 16);
  }
}

__global__ __launch_bounds__(256) void gather_i32_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ idx,
                                                         int32_t* __restrict__ dst, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

// dst[r][t] = src[idx[r]][t], t < nt: the beam reorder of the self-attention caches' HISTORY INDEX (rows of ld ints) -- the
// caches themselves stay where they are (KmbAttnDecode.hist)
__global__ __launch_bounds__(256) void gather_hist_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ idx,
                                                          int32_t* __restrict__ dst, int rows, int ld, int nt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * nt) return;
  const int r = i / nt, t = i - r * nt;
  dst[(size_t)r * ld + t] = src[(size_t)idx[r] * ld + t];
}

// out[i] = i / div  (the beam row -> batch item table of a generation)
__global__ __launch_bounds__(256) void iota_div_kernel(int32_t* __restrict__ out, int n, int div) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = i / div;
}

inline int grid_for(size_t work, int cap = 4096) {
  size_t b = (work + 255) / 256;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

hipError_t kmb_adamw_launch(float* p, const float* g, float* m, float* v, bf16_t* p_bf16, size_t n, KmbAdamW h,
                            hipStream_t stream) {
  if (n == 0) return hipSuccess;
  double step_size = h.lr;
  if (h.correct_bias) {
    const double bc1 = 1.0 - pow(h.beta1, (double)h.step);
    const double bc2 = 1.0 - pow(h.beta2, (double)h.step);
    step_size = step_size * sqrt(bc2) / bc1;
  }
  const size_t n4 = n >> 2;
  // (diagnostic build) KMB_ADAMW_GRID = cap of the grid, KMB_ADAMW_UNROLL = 1 | 2 | 4 float4 per thread and turn
  static const int grid_cap = KMB_DIAG_ENV("KMB_ADAMW_GRID") ? atoi(KMB_DIAG_ENV("KMB_ADAMW_GRID")) : 0;
  static const int unroll = KMB_DIAG_ENV("KMB_ADAMW_UNROLL") ? atoi(KMB_DIAG_ENV("KMB_ADAMW_UNROLL")) : 1;
  if (grid_cap > 0) {
    const dim3 grid(grid_for((n4 + unroll - 1) / unroll, grid_cap));
#define KMB_ADAMW_U(U) hipLaunchKernelGGL((adamw_kernel_u<U>), grid, dim3(256), 0, stream, p, g, m, v, p_bf16, n4, n, (float)h.lr, (float)h.beta1, \
                     (float)h.beta2, (float)(1.0 - h.beta1), (float)(1.0 - h.beta2), (float)h.eps, (float)h.weight_decay, (float)step_size, h.grad_scale)
    if (unroll >= 4) KMB_ADAMW_U(4);
    else if (unroll == 2) KMB_ADAMW_U(2);
    else KMB_ADAMW_U(1);
#undef KMB_ADAMW_U
    return hipGetLastError();
  }
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n4, 8192)), dim3(256), 0, stream, p, g, m, v, p_bf16, n4, n,
                     (float)h.lr, (float)h.beta1, (float)h.beta2, (float)(1.0 - h.beta1), (float)(1.0 - h.beta2),
                     (float)h.eps, (float)h.weight_decay, (float)step_size, h.grad_scale);
  return hipGetLastError();
}

hipError_t kmb_cast_f32_bf16_launch(const float* x, bf16_t* y, size_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n >> 2, 8192)), dim3(256), 0, stream, x, y, n);
  return hipGetLastError();
}

hipError_t kmb_fill_f32_launch(float* x, float v, size_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, v, n);
  return hipGetLastError();
}

hipError_t kmb_cast_rows_launch(const float* x, int ldx, bf16_t* y, int ldy, int rows, int cols, hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(cast_rows_kernel, dim3(grid_for((size_t)rows * ldy)), dim3(256), 0, stream, x, ldx, y, ldy, rows, cols);
  return hipGetLastError();
}

hipError_t kmb_kv_append_launch(const bf16_t* src, int ld_src, bf16_t* cache, int Tmax, int HD, int t, int R,
                                hipStream_t stream) {
  if (R <= 0) return hipSuccess;
  hipLaunchKernelGGL(kv_append_kernel, dim3(grid_for((size_t)R * (HD >> 3))), dim3(256), 0, stream, src, ld_src, cache, Tmax, HD, t, R);
  return hipGetLastError();
}

hipError_t kmb_gather_rows_launch(const void* src, const int32_t* idx, void* dst, int rows, int row_bytes,
                                  size_t stride_bytes, hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  if ((row_bytes & 15) || (stride_bytes & 15)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((size_t)rows * (row_bytes >> 4))), dim3(256), 0, stream,
                     (const char*)src, idx, (char*)dst, rows, row_bytes, stride_bytes);
  return hipGetLastError();
}

hipError_t kmb_iota_div_launch(int32_t* out, int n, int div, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (div <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(iota_div_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, out, n, div);
  return hipGetLastError();
}

hipError_t kmb_gather_hist_launch(const int32_t* src, const int32_t* idx, int32_t* dst, int rows, int ld, int nt, hipStream_t stream) {
  if (rows <= 0 || nt <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_hist_kernel, dim3((rows * nt + 255) / 256), dim3(256), 0, stream, src, idx, dst, rows, ld, nt);
  return hipGetLastError();
}

hipError_t kmb_gather_i32_launch(const int32_t* src, const int32_t* idx, int32_t* dst, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, idx, dst, n);
  return hipGetLastError();
}

hipError_t kmb_gather_rows_multi_launch(const void* const* src, void* const* dst, int n, const int32_t* idx, int rows,
                                        int row_bytes, size_t stride_bytes, hipStream_t stream) {
  if (rows <= 0 || n <= 0) return hipSuccess;
  if ((row_bytes & 15) || (stride_bytes & 15) || n > 16) return hipErrorInvalidValue;
  GatherSet gs;
  for (int i = 0; i < 16; ++i) { gs.src[i] = (const char*)src[i < n ? i : 0]; gs.dst[i] = (char*)dst[i < n ? i : 0]; }
  dim3 grid(grid_for((size_t)rows * (row_bytes >> 4), 256), n);
  hipLaunchKernelGGL(gather_rows_multi_kernel, grid, dim3(256), 0, stream, gs, idx, rows, row_bytes, stride_bytes);
  return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint32_t seed, uint32_t thr16, int rows, int cols,
                                                           uint8_t* __restrict__ keep) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
    keep[i] = drop_keep(seed, (uint32_t)(i / cols), (uint32_t)(i % cols), thr16) ? 1 : 0;
}
}  // namespace

hipError_t kmb_dropout_mask_launch(uint32_t seed, uint32_t thr16, int rows, int cols, uint8_t* keep,
                                   hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((size_t)rows * cols)), dim3(256), 0, stream, seed, thr16, rows, cols, keep);
  return hipGetLastError();
}

namespace {
// s_dev (optional): a device scalar multiplied into the scale; the kernels return at once when the product is 1 (the
// un-scaled loss.backward() of the API path passes autograd's device-side `ones` without a host round trip)
__global__ __launch_bounds__(256) void scale_bf16_kernel(bf16_t* __restrict__ x, size_t n8, float s, const float* s_dev) {
  if (s_dev != nullptr) s *= *s_dev;
  if (s == 1.f) return;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float v[8];
    unpack8(reinterpret_cast<const u32x4*>(x)[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= s;
    reinterpret_cast<u32x4*>(x)[i] = pack8(v);
  }
}
__global__ __launch_bounds__(256) void scale_f32_kernel(float* __restrict__ x, size_t n, float s, const float* s_dev) {
  if (s_dev != nullptr) s *= *s_dev;
  if (s == 1.f) return;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] *= s;
}
}  // namespace

// n must be a multiple of 8
hipError_t kmb_scale_bf16_launch(bf16_t* x, size_t n, float s, const float* s_dev, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(scale_bf16_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, stream, x, n >> 3, s, s_dev);
  return hipGetLastError();
}
// diagnostic (KMB_BWD_TRACE): order-independent 64-bit checksum of a buffer, accumulated into *out by integer atomics
__global__ __launch_bounds__(256) void hash_words_kernel(const uint32_t* __restrict__ x, size_t n, unsigned long long* out) {
  unsigned long long a = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned long long v = x[i] + 0x9E3779B97F4A7C15ull * (i + 1);
    v ^= v >> 29; v *= 0xBF58476D1CE4E5B9ull; v ^= v >> 32;
    a += v;
  }
  atomicAdd(out, a);
}
hipError_t kmb_hash_words_launch(const void* x, size_t nbytes, unsigned long long* out, hipStream_t stream) {
  (void)hipMemsetAsync(out, 0, sizeof(unsigned long long), stream);
  const size_t n = nbytes / 4;
  hipLaunchKernelGGL(hash_words_kernel, dim3(256), dim3(256), 0, stream, (const uint32_t*)x, n, out);
  return hipGetLastError();
}

hipError_t kmb_scale_f32_launch(float* x, size_t n, float s, const float* s_dev, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(scale_f32_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, stream, x, n, s, s_dev);
  return hipGetLastError();
}
