// Tied-vocabulary cross-entropy on fp32 logits (reference src/model/model.py:397-403:
// CrossEntropyLoss(), mean over labels != -100) and the generation-side log_softmax + top-k.
// One 256-thread block per logits row; the row (V = 50320 fp32 = 197 KB) is read twice and is
// L2 / Infinity-Cache resident for the second pass.
#include "common.h"
#include "kernels.h"
#include "embed_row.h"

namespace {

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
  const float mn = fmaxf(m, m2);
  if (mn == -INFINITY) { m = mn; s = 0.f; return; }
  s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
  m = mn;
}

// block-wide (max, sumexp) of row[0:V]; result broadcast to all threads
__device__ __forceinline__ void row_lse(const float* __restrict__ row, int V, float* sh, float& m_out, float& s_out) {
  const int tid = threadIdx.x;
  float m = -INFINITY, s = 0.f;
  const int V4 = V & ~3;
  for (int i = tid * 4; i < V4; i += 1024) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(row + i);
    const float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
    if (mx > m) { s *= __expf(m - mx); m = mx; }
    if (m != -INFINITY) s += __expf(x[0] - m) + __expf(x[1] - m) + __expf(x[2] - m) + __expf(x[3] - m);
  }
  for (int i = V4 + tid; i < V; i += 256) {
    const float x = row[i];
    if (x > m) { s *= __expf(m - x); m = x; }
    if (m != -INFINITY) s += __expf(x - m);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    online_merge(m, s, m2, s2);
  }
  const int wave = tid >> 6;
  if ((tid & 63) == 0) { sh[wave * 2] = m; sh[wave * 2 + 1] = s; }
  __syncthreads();
  m = sh[0]; s = sh[1];
#pragma unroll
  for (int w = 1; w < 4; ++w) online_merge(m, s, sh[w * 2], sh[w * 2 + 1]);
  __syncthreads();
  m_out = m; s_out = s;
}

// one workgroup of 1024 threads, eight independent loads in flight per thread (a 256-thread loop of dependent loads took
// 54 us for the 32768 labels of the benchmark batch)
// A label is VALID when 0 <= label < V; -100 is ignored (CrossEntropyLoss(ignore_index=-100), reference
// src/model/model.py:400-402); any other value is an input error the reference's loss raises on: such rows are treated as
// ignored by every cross-entropy kernel here, are NOT counted, and set bit 1 (value 2) of `status` (may be null) so that the
// host's input check reports them.
__global__ __launch_bounds__(1024) void count_valid_kernel(const int64_t* __restrict__ labels, int n, int V,
                                                           int32_t* __restrict__ count, int32_t* __restrict__ status) {
  __shared__ int sh[16];
  int c = 0, bad = 0;
  int i = threadIdx.x;
  for (; i + 7 * 1024 < n; i += 8 * 1024) {
    int64_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = labels[i + k * 1024];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool ok = v[k] >= 0 && v[k] < V;
      c += ok ? 1 : 0;
      bad |= (!ok && v[k] != -100) ? 1 : 0;
    }
  }
  for (; i < n; i += 1024) {
    const int64_t v = labels[i];
    const bool ok = v >= 0 && v < V;
    c += ok ? 1 : 0;
    bad |= (!ok && v != -100) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  if (bad && status != nullptr) atomicOr(status, 2);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k];
    count[0] = t;
  }
}

__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, int ldv, int V,
                                                 const int64_t* __restrict__ labels, const int32_t* __restrict__ count,
                                                 float grad_scale, float* __restrict__ loss_rows,
                                                 bf16_t* __restrict__ dlogits) {
  __shared__ float sh[8];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* row = logits + (size_t)r * ldv;
  const int64_t label = labels[r];
  const bool valid = label >= 0 && label < V;   // ignored: -100; out of range: flagged by count_valid
  float m = 0.f, s = 1.f;
  if (valid) row_lse(row, V, sh, m, s);  // block-uniform branch
  const float lse = m + __logf(s);
  if (tid == 0) loss_rows[r] = valid ? (lse - row[label]) : 0.f;
  if (dlogits == nullptr) return;
  bf16_t* drow = dlogits + (size_t)r * ldv;
  const int n = count[0];
  const float gs = (valid && n > 0) ? grad_scale / (float)n : 0.f;
  for (int i = tid * 8; i < ldv; i += 2048) {
    float o[8];
    if (valid) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = i + e;
        float p = 0.f;
        if (c < V) {
          p = __expf(row[c] - lse);
          if (c == (int)label) p -= 1.f;
        }
        o[e] = p * gs;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
    }
    *reinterpret_cast<u32x4*>(drow + i) = pack8(o);
  }
}

// Register-resident form for rows of up to NV * 4096 columns: 1024 threads hold the whole fp32 row (NV float4 each),
// so the logits are read from memory once (max, sum-exp and the gradient all come from registers).
template <int NV>
__global__ __launch_bounds__(1024) void ce_kernel_reg(const float* __restrict__ logits, int ldv, int V,
                                                      const int64_t* __restrict__ labels,
                                                      const int32_t* __restrict__ count, float grad_scale,
                                                      float* __restrict__ loss_rows, bf16_t* __restrict__ dlogits) {
  __shared__ float sh[32];
  const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* row = logits + (size_t)r * ldv;
  const int64_t label = labels[r];
  const bool valid = label >= 0 && label < V;  // block-uniform (ignored: -100; out of range: flagged by count_valid)
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
  if (!valid) {
    if (tid == 0) loss_rows[r] = 0.f;
    if (dlogits != nullptr) {
      const u32x2 zero = {0u, 0u};
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int i = (tid + 1024 * j) * 4;
        if (i < ldv) *reinterpret_cast<u32x2*>(dlogits + (size_t)r * ldv + i) = zero;
      }
    }
    return;
  }
  f32x4 x[NV];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = (tid + 1024 * j) * 4;
    if (i < ldv) {
      x[j] = *reinterpret_cast<const f32x4*>(row + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (i + e >= V) x[j][e] = -INFINITY;
        m = fmaxf(m, x[j][e]);
      }
    } else {
      x[j] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
  }
  m = wave_max(m);
  if (lane == 0) sh[wave] = m;
  __syncthreads();
  m = sh[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) m = fmaxf(m, sh[w]);
  float s = 0.f;
  if (m != -INFINITY) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += __expf(x[j][e] - m);
  }
  s = wave_sum(s);
  if (lane == 0) sh[16 + wave] = s;
  __syncthreads();
  s = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) s += sh[16 + w];
  const float lse = m + __logf(s);
  if (tid == 0) loss_rows[r] = lse - row[label];
  if (dlogits == nullptr) return;
  const int n = count[0];
  const float gs = n > 0 ? grad_scale / (float)n : 0.f;
  bf16_t* drow = dlogits + (size_t)r * ldv;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = (tid + 1024 * j) * 4;
    if (i < ldv) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float p = __expf(x[j][e] - lse);  // exp(-inf) = 0 in the pad columns
        if (i + e == (int)label) p -= 1.f;
        o[e] = p * gs;
      }
      const u32x2 pk = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
      *reinterpret_cast<u32x2*>(drow + i) = pk;
    }
  }
}

__global__ __launch_bounds__(1024) void loss_finish_kernel(const float* __restrict__ loss_rows, int rows,
                                                           const int32_t* __restrict__ count, float* __restrict__ loss) {
  __shared__ float sh[16];
  float a = 0.f;
  int i = threadIdx.x;
  for (; i + 7 * 1024 < rows; i += 8 * 1024) {   // eight independent loads in flight per thread
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = loss_rows[i + k * 1024];
    a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  for (; i < rows; i += 1024) a += loss_rows[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int n = count[0];
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k];
    // torch CrossEntropyLoss(mean) over zero valid targets is NaN
    loss[0] = n > 0 ? t / (float)n : __uint_as_float(0x7fc00000u);
  }
}

// The training head keeps its logits in bf16 (the reference's AMP path holds them in fp16, src/training.py:118-134 with
// autocast; CE itself runs in fp32 on the up-cast values, as here): 1024 threads hold the row as NV8 chunks of 8 bf16
// (read once, 16-byte loads), max / sum-exp / loss in fp32 registers, and the gradient (softmax - onehot) * scale / count
// is written back IN PLACE over the logits (dlogits == logits is allowed: every element is read before the first
// store of its thread, and a thread only rewrites what it read).  Halves the head GEMM's store and the CE's read
// against fp32 logits: 3.3 GB -> 1.65 GB each per step at the benchmark batch.
template <int NV8>
__global__ __launch_bounds__(1024) void ce_kernel_reg_bf16(const bf16_t* logits, int ldv, int V,
                                                           const int64_t* __restrict__ labels,
                                                           const int32_t* __restrict__ count, float grad_scale,
                                                           float* __restrict__ loss_rows, bf16_t* dlogits) {
  __shared__ float sh[33];
  const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bf16_t* row = logits + (size_t)r * ldv;
  const int64_t label = labels[r];
  const bool valid = label >= 0 && label < V;  // block-uniform (ignored: -100; out of range: flagged by count_valid)
  if (!valid) {
    if (tid == 0) loss_rows[r] = 0.f;
    if (dlogits != nullptr) {
      const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < NV8; ++j) {
        const int i = (tid + 1024 * j) * 8;
        if (i < ldv) *reinterpret_cast<u32x4*>(dlogits + (size_t)r * ldv + i) = zero;
      }
    }
    return;
  }
  float x[NV8][8];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV8; ++j) {
    const int i = (tid + 1024 * j) * 8;
    if (i < ldv) {
      unpack8(*reinterpret_cast<const u32x4*>(row + i), x[j]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (i + e >= V) x[j][e] = -INFINITY;
        m = fmaxf(m, x[j][e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[j][e] = -INFINITY;
    }
  }
  m = wave_max(m);
  if (lane == 0) sh[wave] = m;
  __syncthreads();
  m = sh[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) m = fmaxf(m, sh[w]);
  float s = 0.f;
  if (m != -INFINITY) {
#pragma unroll
    for (int j = 0; j < NV8; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s += __expf(x[j][e] - m);
        // the label's logit comes from the registers of the one thread that holds it (the row may be overwritten below)
        if ((tid + 1024 * j) * 8 + e == (int)label) sh[32] = x[j][e];
      }
  }
  s = wave_sum(s);
  if (lane == 0) sh[16 + wave] = s;
  __syncthreads();
  s = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) s += sh[16 + w];
  const float lse = m + __logf(s);
  if (tid == 0) loss_rows[r] = lse - sh[32];
  if (dlogits == nullptr) return;
  const int n = count[0];
  const float gs = n > 0 ? grad_scale / (float)n : 0.f;
  bf16_t* drow = dlogits + (size_t)r * ldv;
#pragma unroll
  for (int j = 0; j < NV8; ++j) {
    const int i = (tid + 1024 * j) * 8;
    if (i < ldv) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float p = __expf(x[j][e] - lse);  // exp(-inf) = 0 in the pad columns
        if (i + e == (int)label) p -= 1.f;
        o[e] = p * gs;
      }
      *reinterpret_cast<u32x4*>(drow + i) = pack8(o);
    }
  }
}

// Per row: logp = log_softmax(row) (or the forced-token distribution), then the k best of logp + add[row] in
// (value desc, index asc) order.  Thread t owns elements t, t+256, ...; it keeps its own best in registers, a round
// is one block-wide arg-max over those 256 candidates, and only the winner's wave rescans the winner's ~V/256
// elements (64 lanes wide, from L2) for its next candidate -- the row is swept 3 times in total instead of k+2.
__global__ __launch_bounds__(256) void logsoftmax_topk_kernel(const float* __restrict__ logits, int ldv, int V,
                                                              const float* __restrict__ add, int force_token,
                                                              int ban, int k, float* __restrict__ out_val,
                                                              int32_t* __restrict__ out_idx) {
  __shared__ float sh[8];
  __shared__ float shv[4];
  __shared__ int shi[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const float* row = logits + (size_t)r * ldv;
  const float a = add != nullptr ? add[r] : 0.f;
  if (force_token >= 0) {
    // every other logit is -inf: log_softmax is 0 at the forced token, -inf elsewhere (ties in index order)
    for (int j = tid; j < k; j += 256) {
      out_val[(size_t)r * k + j] = j == 0 ? a : -INFINITY;
      out_idx[(size_t)r * k + j] = j == 0 ? force_token : (j - 1 < force_token ? j - 1 : j);
    }
    return;
  }
  float m, s;
  row_lse(row, V, sh, m, s);
  const float lse = m + __logf(s);
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  // `ban`: that token scores -inf AFTER the normalisation (the log-sum-exp above includes it)
  for (int i = tid; i < V; i += 256) {
    const float x = i == ban ? -INFINITY : row[i];
    if (x > bv || bi == 0x7fffffff) { bv = x; bi = i; }
  }
  for (int j = 0; j < k; ++j) {
    float wv = bv;
    int wi = bi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(wv, o, 64);
      const int i2 = __shfl_xor(wi, o, 64);
      if (v2 > wv || (v2 == wv && i2 < wi)) { wv = v2; wi = i2; }
    }
    if (lane == 0) { shv[tid >> 6] = wv; shi[tid >> 6] = wi; }
    __syncthreads();
    wv = shv[0]; wi = shi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (shv[w] > wv || (shv[w] == wv && shi[w] < wi)) { wv = shv[w]; wi = shi[w]; }
    __syncthreads();
    if (tid == 0) {
      out_val[(size_t)r * k + j] = (wv - lse) + a;
      out_idx[(size_t)r * k + j] = wi;
    }
    if (wi == 0x7fffffff) continue;  // fewer than k elements in the row
    const int owner = wi & 255;
    if ((tid >> 6) == (owner >> 6)) {  // wave-uniform: the owner's wave finds the owner's next candidate
      float nv = -INFINITY;
      int ni = 0x7fffffff;
      for (int i = owner + 256 * lane; i < V; i += 256 * 64) {
        const float x = i == ban ? -INFINITY : row[i];
        const bool after = (x < wv) || (x == wv && i > wi);
        if (after && (x > nv || ni == 0x7fffffff || (x == nv && i < ni))) { nv = x; ni = i; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(nv, o, 64);
        const int i2 = __shfl_xor(ni, o, 64);
        if (i2 != 0x7fffffff && (ni == 0x7fffffff || v2 > nv || (v2 == nv && i2 < ni))) { nv = v2; ni = i2; }
      }
      if (tid == owner) { bv = nv; bi = ni; }
    }
  }
}

// Register-resident form (rows of up to NV * 4096 columns): 1024 threads hold the fp32 row, so the logits are read
// from memory ONCE -- max, sum-exp, every thread's candidate and the owner's rescans all come from registers (the
// 256-thread form above sweeps the row three times and rescans from L2; 102 us -> per decode step at 320 rows).
// Thread t owns the float4 chunks t, t + 1024, ...; same (value desc, index asc) order.
template <int NV>
__global__ __launch_bounds__(1024) void logsoftmax_topk_reg_kernel(const float* __restrict__ logits, int ldv, int V,
                                                                   const float* __restrict__ add, int force_token,
                                                                   int ban, int k, float* __restrict__ out_val,
                                                                   int32_t* __restrict__ out_idx) {
  __shared__ float sh[32];
  __shared__ float shv[16];
  __shared__ int shi[16];
  const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* row = logits + (size_t)r * ldv;
  const float a = add != nullptr ? add[r] : 0.f;
  if (force_token >= 0) {
    for (int j = tid; j < k; j += 1024) {
      out_val[(size_t)r * k + j] = j == 0 ? a : -INFINITY;
      out_idx[(size_t)r * k + j] = j == 0 ? force_token : (j - 1 < force_token ? j - 1 : j);
    }
    return;
  }
  f32x4 x[NV];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = (tid + 1024 * j) * 4;
    if (i < ldv) {
      x[j] = *reinterpret_cast<const f32x4*>(row + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (i + e >= V) x[j][e] = -INFINITY;
        m = fmaxf(m, x[j][e]);
      }
    } else {
      x[j] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
  }
  m = wave_max(m);
  if (lane == 0) sh[wave] = m;
  __syncthreads();
  m = sh[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) m = fmaxf(m, sh[w]);
  float s = 0.f;
  if (m != -INFINITY) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += __expf(x[j][e] - m);
  }
  s = wave_sum(s);
  if (lane == 0) sh[16 + wave] = s;
  __syncthreads();
  s = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) s += sh[16 + w];
  const float lse = m + __logf(s);
  if (ban >= 0) {   // banned token: -inf after the normalisation
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if ((tid + 1024 * j) * 4 + e == ban) x[j][e] = -INFINITY;
  }
  // Candidates are 64-bit keys: (order-preserving bits of the value) << 32 | (0x7fffffff - index), so "larger key" IS
  // (value desc, index asc) and every comparison is one branch-free unsigned compare.  (The first version compared
  // value and index with short-circuit logic: the compiler turned each element of the rescan into branches, 5.5 us
  // per round.)  Key 0 = no element.
  auto make_key = [](float v, int i) -> unsigned long long {
    const uint32_t b = __float_as_uint(v);
    const uint32_t ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)ord << 32) | (uint32_t)(0x7fffffff - i);
  };
  unsigned long long best = 0ull;
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = (tid + 1024 * j) * 4 + e;
      const unsigned long long key = i < V ? make_key(x[j][e], i) : 0ull;
      best = key > best ? key : best;
    }
  __shared__ unsigned long long shk[16];
  for (int jr = 0; jr < k; ++jr) {
    unsigned long long wk = best;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t lo = __shfl_xor((uint32_t)wk, o, 64);
      const uint32_t hi = __shfl_xor((uint32_t)(wk >> 32), o, 64);
      const unsigned long long k2 = ((unsigned long long)hi << 32) | lo;
      wk = k2 > wk ? k2 : wk;
    }
    if (lane == 0) shk[wave] = wk;
    __syncthreads();
    wk = shk[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) {
      const unsigned long long k2 = shk[w];
      wk = k2 > wk ? k2 : wk;
    }
    __syncthreads();
    const int wi = wk != 0ull ? 0x7fffffff - (int)(uint32_t)wk : 0x7fffffff;
    if (tid == 0) {
      const uint32_t ord = (uint32_t)(wk >> 32);
      const float wv = wk != 0ull ? __uint_as_float((ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord) : -INFINITY;
      out_val[(size_t)r * k + jr] = (wv - lse) + a;
      out_idx[(size_t)r * k + jr] = wi;
    }
    if (wk == 0ull) continue;  // fewer than k elements in the row
    const int owner = (wi >> 2) & 1023;
    if (wave == (owner >> 6)) {   // wave-uniform: the owner's next candidate, out of its registers
      unsigned long long nk = 0ull;
#pragma unroll
      for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = (tid + 1024 * j) * 4 + e;
          unsigned long long key = i < V ? make_key(x[j][e], i) : 0ull;
          key = key < wk ? key : 0ull;
          nk = key > nk ? key : nk;
        }
      if (tid == owner) best = nk;
    }
  }
}

// ---- two-launch form for decode-sized problems (a few hundred rows x 50k columns) ----
// One workgroup per row leaves the launch as long as the slowest CU: 320 rows on 256 CUs = 64 CUs with two 200 KB rows
// to pull in (41 us before the first candidate is out), and every one of the k selection rounds is a workgroup-wide
// arg-max plus a rescan of the winner's 52 registers (5.5 us per round: 90 us at k = 10).  Here a row is split into
// TOPK_PARTS parts, one 256-thread workgroup each (1280 workgroups: every CU streams): a part computes its own
// (max, sum-exp) and its own k best -- per wave, with no workgroup barrier in the rounds: every thread keeps its best
// TWO keys, so popping a winner is a register move and the 52-element rescan only runs for a thread that wins a second
// time -- and a second, tiny launch combines the parts of a row: log-sum-exp from the (max, sum) pairs, k best of the
// TOPK_PARTS * k candidates.  Same (value desc, index asc) order, same outputs.
constexpr int TOPK_PARTS = 4, TOPK_KMAX = 16, TOPK_PW = 2 + 2 * TOPK_KMAX;   // words per (row, part) record

__device__ __forceinline__ unsigned long long topk_key(float v, int i) {
  const uint32_t b = __float_as_uint(v);
  const uint32_t ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ((unsigned long long)ord << 32) | (uint32_t)(0x7fffffff - i);
}
__device__ __forceinline__ float topk_key_value(unsigned long long key) {
  const uint32_t ord = (uint32_t)(key >> 32);
  return __uint_as_float((ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord);
}
// Wave-wide maximum of a 64-bit key, result in every lane.  DPP moves instead of the shuffle tree (twelve dependent
// ds_bpermute per call through the LDS crossbar: 2.3 us per selection round with five waves per SIMD): rotations by
// 1, 2, 4, 8 inside each row of 16 lanes leave the row maximum in every lane of the row (max is idempotent), row_bcast:15
// folds row 0 into 1 and 2 into 3, row_bcast:31 folds lane 31 into rows 2 and 3; lane 63 then holds the wave maximum.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_step(unsigned long long v) {
  const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xf, false);
  const unsigned long long v2 = ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;   // lanes outside ROW_MASK: v itself
  return v2 > v ? v2 : v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = dpp_max_step<0x121, 0xf>(v);   // row_ror:1
  v = dpp_max_step<0x122, 0xf>(v);   // row_ror:2
  v = dpp_max_step<0x124, 0xf>(v);   // row_ror:4
  v = dpp_max_step<0x128, 0xf>(v);   // row_ror:8
  v = dpp_max_step<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_max_step<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
  const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, 63);
  const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

// wave-wide maximum of a float, result in every lane (same DPP pattern as wave_max_u64)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_maxf_step(float v) {
  const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
  return fmaxf(v, o);
}
__device__ __forceinline__ float wave_max_f32_dpp(float v) {
  v = dpp_maxf_step<0x121, 0xf>(v);
  v = dpp_maxf_step<0x122, 0xf>(v);
  v = dpp_maxf_step<0x124, 0xf>(v);
  v = dpp_maxf_step<0x128, 0xf>(v);
  v = dpp_maxf_step<0x142, 0xa>(v);
  v = dpp_maxf_step<0x143, 0xc>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Selection inside a wave (64 lanes x 4 NV values, k best as 64-bit keys into cand[0 .. TOPK_KMAX), 0 = none), round 4:
// the kernel is bound by its vector-ALU work (5 waves per SIMD x ~2500 instructions; the first form built a 64-bit key
// per element and kept every thread's best two keys: 14 instructions per element, 220 registers, three rounds of
// workgroups per launch), so the keys are not built at all for elements that cannot win:
//   1. every lane's two largest VALUES (three instructions per element),
//   2. at most k rounds of a wave-wide maximum over the lanes' current best, the winners popping to their second value,
//      until k values have been counted: the last maximum T is a LOWER bound of the wave's k-th largest value (a lane
//      holding three or more of the k best has contributed only two, the count was filled by smaller ones),
//   3. one pass over the elements: the few with value >= T (k, plus ties and the third-bests of step 2) get their key
//      and a slot of cand[] (ballot + prefix count).  The caller's merge of the waves orders by key as before.
// Fewer than k finite values in the wave, or more than TOPK_KMAX elements >= T (rows of equal logits): the exact, slow
// form below -- k rounds of "largest key below the previous winner" over all elements.  Same keys, same order of the
// final result either way.
template <int NV>
__device__ __forceinline__ void wave_topk(const f32x4 (&x)[NV], int idx0, int chunks_left, int V, int ban, int k, int lane,
                                          unsigned long long* cand) {
  // idx0: index of x[0][0] of this thread; element (j, e) has index idx0 + 1024 j + e; valid iff j * 256 < chunks_left
  // (chunks_left = chunks_per_part - tid), index < V and != ban.  Invalid elements hold -inf (the caller masked them).
  float v1 = -INFINITY, v2 = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = fminf(v1, x[j][e]);
      v1 = fmaxf(v1, x[j][e]);
      v2 = fmaxf(v2, t);
    }
  int cnt = 0;
  float T = -INFINITY;
  bool exact = false;
  for (int jr = 0; jr < k && cnt < k; ++jr) {
    const float wm = wave_max_f32_dpp(v1);
    if (wm == -INFINITY) { exact = true; break; }
    const bool mine = v1 == wm;
    cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(mine));
    T = wm;
    if (mine) { v1 = v2; v2 = -INFINITY; }
  }
  if (cnt < k) exact = true;
  int n = 0;
  if (!exact) {
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool hit = x[j][e] >= T;   // T > -inf: invalid elements never hit
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
        if (bal != 0ull) {
          const int slot = n + __builtin_popcountll(bal & below);
          if (hit && slot < TOPK_KMAX) cand[slot] = topk_key(x[j][e], idx0 + 1024 * j + e);
          n += __builtin_popcountll(bal);
        }
      }
    if (n > TOPK_KMAX) exact = true;
  }
  if (!exact) {
    if (lane >= n && lane < TOPK_KMAX) cand[lane] = 0ull;
    return;
  }
  unsigned long long last = ~0ull;
  for (int jr = 0; jr < TOPK_KMAX; ++jr) {
    unsigned long long best = 0ull;
    if (jr < k && last != 0ull) {
#pragma unroll
      for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = idx0 + 1024 * j + e;
          unsigned long long key = (j * 256 < chunks_left && i < V && i != ban) ? topk_key(x[j][e], i) : 0ull;
          key = key < last ? key : 0ull;
          best = key > best ? key : best;
        }
      best = wave_max_u64(best);
      last = best;
    }
    if (lane == 0) cand[jr] = best;
  }
}

#ifndef KMB_TOPK_OCC
#define KMB_TOPK_OCC 4   // workgroups per CU the register budget is cut for (128 registers: no spills; 2, 4 and 5 measure alike)
#endif
template <int NV>   // float4 chunks per thread: chunks per part <= NV * 256
__global__ __launch_bounds__(256, KMB_TOPK_OCC) void topk_part_kernel(const float* __restrict__ logits, int ldv, int V, int ban, int k,
                                                                      int chunks_per_part, float* __restrict__ part) {
  __shared__ float sh[8];
  __shared__ unsigned long long cand[4][TOPK_KMAX];
  const int p = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* row = logits + (size_t)r * ldv;
  f32x4 x[NV];
  float m = -INFINITY;
  // every load first, from a clamped address and with no branch around it: a load inside `if (in range) { load; mask; }` is
  // followed by its own s_waitcnt vmcnt(0), i.e. NV dependent memory round trips per thread (rounds 2-3 shipped that)
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = tid + 256 * j;
    const int i = (p * chunks_per_part + c) * 4;
    x[j] = *reinterpret_cast<const f32x4*>(row + ((c < chunks_per_part && i < ldv) ? i : 0));
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = tid + 256 * j;
    const int i = (p * chunks_per_part + c) * 4;
    const bool in = c < chunks_per_part && i < ldv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!in || i + e >= V) x[j][e] = -INFINITY;
      m = fmaxf(m, x[j][e]);
    }
  }
  m = wave_max(m);
  if (lane == 0) sh[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  float sum = 0.f;
  if (m != -INFINITY) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += __expf(x[j][e] - m);
  }
  sum = wave_sum(sum);
  if (lane == 0) sh[4 + wave] = sum;
  // the banned token is excluded AFTER the normalisation above
  const int idx0 = (p * chunks_per_part + tid) * 4;
  if (ban >= 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (idx0 + 1024 * j + e == ban) x[j][e] = -INFINITY;
  }
  wave_topk<NV>(x, idx0, chunks_per_part - tid, V, ban, k, lane, cand[wave]);
  __syncthreads();
  if (wave == 0) {
    float* out = part + ((size_t)r * TOPK_PARTS + p) * TOPK_PW;
    if (lane == 0) { out[0] = m; out[1] = (sh[4] + sh[5]) + (sh[6] + sh[7]); }
    unsigned long long key = cand[lane >> 4][lane & 15];   // 4 x TOPK_KMAX = 64
    for (int jr = 0; jr < k; ++jr) {
      const unsigned long long wk = wave_max_u64(key);
      if (lane == 0) {
        out[2 + 2 * jr] = wk != 0ull ? topk_key_value(wk) : -INFINITY;
        reinterpret_cast<int32_t*>(out)[3 + 2 * jr] = wk != 0ull ? 0x7fffffff - (int)(uint32_t)wk : 0x7fffffff;
      }
      if (key == wk) key = 0ull;
    }
  }
}

// one wave per row: combine the parts (val_row / idx_row: the row's k outputs, global memory or LDS)
__device__ __forceinline__ void topk_combine_row(const float* __restrict__ part, int r, const float* __restrict__ add, int force_token,
                                                 int k, float* val_row, int32_t* idx_row, int lane) {
  const float a = add != nullptr ? add[r] : 0.f;
  if (force_token >= 0) {   // every other logit is -inf: log_softmax is 0 at the forced token (ties in index order)
    for (int j = lane; j < k; j += 64) {
      val_row[j] = j == 0 ? a : -INFINITY;
      idx_row[j] = j == 0 ? force_token : (j - 1 < force_token ? j - 1 : j);
    }
    return;
  }
  const float* rec = part + (size_t)r * TOPK_PARTS * TOPK_PW;
  float M = -INFINITY;
#pragma unroll
  for (int q = 0; q < TOPK_PARTS; ++q) M = fmaxf(M, rec[q * TOPK_PW]);
  float S = 0.f;
#pragma unroll
  for (int q = 0; q < TOPK_PARTS; ++q) {
    const float mq = rec[q * TOPK_PW];
    if (mq != -INFINITY) S += rec[q * TOPK_PW + 1] * __expf(mq - M);
  }
  const float lse = M + __logf(S);
  unsigned long long key = 0ull;
  if (lane < TOPK_PARTS * k) {
    const float* c = rec + (lane / k) * TOPK_PW + 2 + 2 * (lane % k);
    const int i = reinterpret_cast<const int32_t*>(c)[1];
    if (i != 0x7fffffff) key = topk_key(c[0], i);
  }
  for (int jr = 0; jr < k; ++jr) {
    const unsigned long long wk = wave_max_u64(key);
    if (lane == 0) {
      val_row[jr] = ((wk != 0ull ? topk_key_value(wk) : -INFINITY) - lse) + a;
      idx_row[jr] = wk != 0ull ? 0x7fffffff - (int)(uint32_t)wk : 0x7fffffff;
    }
    if (key == wk) key = 0ull;
  }
}
__global__ __launch_bounds__(256) void topk_combine_kernel(const float* __restrict__ part, int rows, const float* __restrict__ add,
                                                           int force_token, int k, float* __restrict__ out_val,
                                                           int32_t* __restrict__ out_idx) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  topk_combine_row(part, r, add, force_token, k, out_val + (size_t)r * k, out_idx + (size_t)r * k, lane);
}

// Beam search, per batch item: the best `k` of the nb * k candidates its beams produced (val / idx from
// logsoftmax_topk_kernel, rows b*nb .. b*nb + nb - 1), ordered by (value desc, candidate position asc) -- what
// torch.topk over the [nb * V] scores of mixins.py's beam step returns, restricted to each beam's own top k (enough:
// the best k overall contain at most k from any beam).  Output, packed for ONE device-to-host copy:
// out[(b*k + j)*2] = score bits (fp32), out[(b*k + j)*2 + 1] = beam * V + token.
//
// next_scores != null: the same launch also picks the beams of the next step, so that the decode loop needs no host
// round trip: in candidate order, the first nb candidates whose token is not EOS (transformers 3.0.2
// _generate_beam_search: an EOS candidate either closes a hypothesis or is skipped, it never continues a beam) ->
// next_scores / next_tokens / next_beam_idx [b*nb + i] (beam index = row of the KV cache to continue from).  What the
// host does with the finished hypotheses and with `done` batch items does not feed back into the other rows.
// (val / idx: the nb * k candidates of batch item b -- global memory or LDS; one wave)
__device__ __forceinline__ void beam_merge_item(const float* val, const int32_t* idx, int b, int lane, int nb, int k, int V,
                                                int32_t* __restrict__ out, int eos, float* __restrict__ next_scores,
                                                int64_t* __restrict__ next_tokens, int32_t* __restrict__ next_beam_idx,
                                                int32_t* snext = nullptr) {
  // (snext != nullptr: the nb chosen beam rows also into that LDS array [0, 16) and their tokens into [16, 32), for the history gather
  //  and the next step's embedding that follow in the same launch)
  // one wave per batch item; lane l holds candidates l, l + 64, l + 128, l + 192 (n <= 256) as 64-bit keys
  // (value bits in order | ~position): a selection round is one DPP wave maximum (round 4; the first form ran a shuffle
  // tree over (value, position) pairs, a workgroup barrier and a global load of the winner's token per round: 1 us each)
  const int n = nb * k;
  unsigned long long key[4];
  int tokv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = lane + 64 * q;
    const bool in = i < n;
    const float v = val[in ? i : 0];
    tokv[q] = idx[in ? i : 0];
    key[q] = in ? topk_key(v, i) : 0ull;     // topk_key: larger value first, then smaller position; never 0 for a real entry
  }
  int n_sel = 0;
  for (int j = 0; j < k; ++j) {
    unsigned long long best = key[0] > key[1] ? key[0] : key[1];
    const unsigned long long b23 = key[2] > key[3] ? key[2] : key[3];
    best = best > b23 ? best : b23;
    const unsigned long long wk = wave_max_u64(best);   // k <= n: there is always a candidate left
    const int bi = 0x7fffffff - (int)(uint32_t)wk;
    const int q = bi >> 6;
    int tok = 0;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      if (qq == q && lane == (bi & 63)) { tok = tokv[qq]; key[qq] = 0ull; }
    }
    tok = __shfl(tok, bi & 63, 64);   // the owner's token to every lane
    if (lane == 0) {
      const float bv = topk_key_value(wk);
      const int beam = bi / k;
      out[((size_t)b * k + j) * 2] = __float_as_int(bv);
      out[((size_t)b * k + j) * 2 + 1] = beam * V + tok;
      if (next_scores != nullptr && n_sel < nb && tok != eos) {
        if (snext != nullptr) { snext[n_sel] = b * nb + beam; snext[16 + n_sel] = tok; }
        const size_t o = (size_t)b * nb + n_sel++;
        next_scores[o] = bv; next_tokens[o] = tok; next_beam_idx[o] = b * nb + beam;
      }
    }
  }
  if (lane == 0 && next_scores != nullptr) {
    for (; n_sel < nb; ++n_sel) {   // cannot happen with k >= 2 * nb (at most one EOS candidate per beam); keep the rows defined
      if (snext != nullptr) { snext[n_sel] = b * nb; snext[16 + n_sel] = eos >= 0 ? eos : 0; }
      const size_t o = (size_t)b * nb + n_sel;
      next_scores[o] = -1e9f; next_tokens[o] = eos >= 0 ? eos : 0; next_beam_idx[o] = b * nb;
    }
  }
}
// The beam reorder of the self-attention caches' history index (optim.hip gather_hist_kernel: dst[r][t] = src[next_beam_idx[r]][t], t < nt)
// for the item's nb rows, by the wave that has just chosen them: the decode loop's reorder launch folded into its beam step (round 6).
__device__ __forceinline__ void beam_hist_gather(const KmbHistGather& hg, const int32_t* snext, int b, int nb, int lane) {
  if (hg.dst == nullptr) return;
  __builtin_amdgcn_wave_barrier();   // snext was written by lane 0 of this wave
  for (int e = lane; e < nb * hg.nt; e += 64) {
    const int i = e / hg.nt, t = e - i * hg.nt;
    hg.dst[(size_t)(b * nb + i) * hg.ld + t] = hg.src[(size_t)snext[i] * hg.ld + t];
  }
}

// ... and the next decode step's input rows: embedding of the chosen tokens + position + LayerNorm (embed.hip's embed_ln_fwd_kernel, the
// same row code: embed_row.h), one wave per beam row; all the workgroup's threads call this (it has a barrier)
__device__ __forceinline__ void beam_embed_next(const KmbEmbedNext& en, const int32_t* snext, int b, int nb, int wave, int lane) {
  if (en.E == nullptr) return;
  __syncthreads();   // snext[16 ..] was written by wave 0
  if (wave < nb) {
    const int row = b * nb + wave;
    int tok = snext[16 + wave];
    tok = tok < 0 ? 0 : (tok >= en.V ? en.V - 1 : tok);   // (a real column unless V < k; never read outside the table)
    embed_ln_row<2>(en.E + (size_t)tok * en.D, en.prow, en.scale, en.gamma, en.beta, nullptr, en.y, nullptr, nullptr, row,
                    en.D, en.eps, KmbDrop{0u, 0u, 1.f}, lane);
  }
}

__global__ __launch_bounds__(64) void beam_merge_kernel(const float* __restrict__ val, const int32_t* __restrict__ idx,
                                                        int nb, int k, int V, int32_t* __restrict__ out, int eos,
                                                        float* __restrict__ next_scores, int64_t* __restrict__ next_tokens,
                                                        int32_t* __restrict__ next_beam_idx) {
  const int b = blockIdx.x, n = nb * k;
  beam_merge_item(val + (size_t)b * n, idx + (size_t)b * n, b, threadIdx.x, nb, k, V, out, eos, next_scores, next_tokens, next_beam_idx);
}
// The two in one launch per batch item (the decode loop): wave w < nb combines the parts of beam row b * nb + w into LDS, wave 0
// then merges the item's nb * k candidates.  Same arithmetic and order as topk_combine_kernel + beam_merge_kernel.
__global__ __launch_bounds__(1024) void beam_combine_merge_kernel(const float* __restrict__ part, const float* __restrict__ add,
                                                                  int force_token, int nb, int k, int V, int32_t* __restrict__ out,
                                                                  int eos, float* __restrict__ next_scores,
                                                                  int64_t* __restrict__ next_tokens, int32_t* __restrict__ next_beam_idx,
                                                                  const KmbHistGather hg, const KmbEmbedNext en) {
  __shared__ float sval[256];
  __shared__ int32_t sidx[256];
  __shared__ int32_t snext[32];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < nb) topk_combine_row(part, b * nb + wave, add, force_token, k, sval + wave * k, sidx + wave * k, lane);
  __syncthreads();
  if (wave == 0) {
    beam_merge_item(sval, sidx, b, lane, nb, k, V, out, eos, next_scores, next_tokens, next_beam_idx, snext);
    beam_hist_gather(hg, snext, b, nb, lane);
  }
  beam_embed_next(en, snext, b, nb, wave, lane);
}


// ---- the beam step behind the all-rows vocabulary projection's STATS epilogue (round 6) ----
// gemm.hip's gemm_kernel_allrows<true> leaves, per row and 256-column block, the block's maximum logit and its sum of exp(v - maximum):
// stats[(row * nblk + blk) * 2], [... + 1].  With them ONE launch does what topk_part_kernel +
// beam_combine_merge_kernel did, without streaming the 64 MB of logits again (topk_part: 22.9 us of a 553 us decode step):
//   * log-sum-exp of the row from the nblk pairs;
//   * T = the (k + 1 if a token is banned)-th largest block maximum: at least k admissible logits are >= T, so the k best are among
//     the elements >= T, and those sit in the blocks whose maximum is >= T -- typically k or k + 1 blocks of 1 KB, read HS_BATCH at a time;
//   * the k best of those candidates by (value desc, index asc) key, as everywhere else; then the item's merge (beam_merge_item).
// More than HS_CAND elements >= T (rows of equal logits): k rounds of "largest key below the previous winner" over the selected blocks.
// Same selection as the two-launch path; the scores differ from it in the last bits only through the log-sum-exp's grouping
// (197 blocks of 256 columns instead of 4 parts).
constexpr int HS_ROWS = 320, HS_COLS = 256, HS_CAND = 128, HS_BLK_MAX = 256, HS_BATCH = 12;   // HS_BATCH: blocks read at a time (k = 10: 10 or 11 blocks)

__device__ __forceinline__ void beam_stats_row(const float* __restrict__ logits, int ldv, int V, const float* __restrict__ stats, int nblk,
                                               int r, const float* __restrict__ add, int force_token, int ban, int k, float* val_row,
                                               int32_t* idx_row, unsigned long long* cand, uint16_t* sel, int lane) {
  const float a = add != nullptr ? add[r] : 0.f;
  if (force_token >= 0) {   // (topk_combine_row)
    for (int j = lane; j < k; j += 64) {
      val_row[j] = j == 0 ? a : -INFINITY;
      idx_row[j] = j == 0 ? force_token : (j - 1 < force_token ? j - 1 : j);
    }
    return;
  }
  float mq[4], sq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int blk = lane + 64 * q;
    const bool in = blk < nblk;
    const float2 ms = reinterpret_cast<const float2*>(stats)[(size_t)r * nblk + (in ? blk : 0)];
    mq[q] = in ? ms.x : -INFINITY;
    sq[q] = in ? ms.y : 0.f;
  }
  const float M = wave_max_f32_dpp(fmaxf(fmaxf(mq[0], mq[1]), fmaxf(mq[2], mq[3])));
  float S = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (mq[q] != -INFINITY) S += sq[q] * __expf(mq[q] - M);
  S = wave_sum(S);
  const float lse = M + __logf(S);
  // T: the keff-th largest block maximum (a lane's four, sorted, pop one per round: equal maxima are counted one per lane and round)
  const int keff = k + (ban >= 0 ? 1 : 0);
  float h0 = mq[0], h1 = mq[1], h2 = mq[2], h3 = mq[3];
  {
    float t;
#define KMB_CSWAP(x, y) t = fminf(x, y); x = fmaxf(x, y); y = t;
    KMB_CSWAP(h0, h1) KMB_CSWAP(h2, h3) KMB_CSWAP(h0, h2) KMB_CSWAP(h1, h3) KMB_CSWAP(h1, h2)
#undef KMB_CSWAP
  }
  float T = -INFINITY;
  int cnt = 0;
  for (int jr = 0; jr < keff && cnt < keff; ++jr) {
    const float wm = wave_max_f32_dpp(h0);
    if (wm == -INFINITY) break;
    const bool mine = h0 == wm;
    cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(mine));
    T = wm;
    if (mine) { h0 = h1; h1 = h2; h2 = h3; h3 = -INFINITY; }
  }
  if (cnt < keff) T = -INFINITY;   // fewer blocks than keff: every block is read, every admissible element is a candidate
  const unsigned long long below = (1ull << lane) - 1ull;
  int n_sel = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool pick = lane + 64 * q < nblk && mq[q] >= T;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(pick);
    if (pick) sel[n_sel + __builtin_popcountll(bal & below)] = (uint16_t)(lane + 64 * q);
    n_sel += __builtin_popcountll(bal);
  }
  __builtin_amdgcn_wave_barrier();
  const float* row = logits + (size_t)r * ldv;
  int n = 0;
  for (int base = 0; base < n_sel; base += HS_BATCH) {
    f32x4 x[HS_BATCH];
    int c0[HS_BATCH];
#pragma unroll
    for (int u = 0; u < HS_BATCH; ++u) {   // every load first, clamped
      const int blk = sel[base + u < n_sel ? base + u : base];
      c0[u] = blk * HS_COLS + lane * 4;
      const bool in = base + u < n_sel && c0[u] < V;     // V % 4 == 0, ldv >= V: a group of four is inside or outside
      x[u] = *reinterpret_cast<const f32x4*>(row + (in ? c0[u] : 0));
      if (!in) c0[u] = -1;
    }
#pragma unroll
    for (int u = 0; u < HS_BATCH; ++u) {
      if (base + u < n_sel) {   // wave-uniform (no `break`: the loop must unroll, x[] lives in registers)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool hit = c0[u] >= 0 && !(x[u][e] < T) && c0[u] + e != ban;   // (!(x < T): a NaN is a candidate, as in the two-launch path)
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
        if (bal != 0ull) {
          const int slot = n + __builtin_popcountll(bal & below);
          if (hit && slot < HS_CAND) cand[slot] = topk_key(x[u][e], c0[u] + e);
          n += __builtin_popcountll(bal);
        }
      }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (n >= k && n <= HS_CAND) {
    unsigned long long k0 = lane < n ? cand[lane] : 0ull;
    unsigned long long k1 = lane + 64 < n ? cand[lane + 64] : 0ull;
    for (int jr = 0; jr < k; ++jr) {
      const unsigned long long wk = wave_max_u64(k0 > k1 ? k0 : k1);
      if (lane == 0) {
        val_row[jr] = ((wk != 0ull ? topk_key_value(wk) : -INFINITY) - lse) + a;
        idx_row[jr] = wk != 0ull ? 0x7fffffff - (int)(uint32_t)wk : 0x7fffffff;
      }
      if (k0 == wk) k0 = 0ull;
      if (k1 == wk) k1 = 0ull;
    }
    return;
  }
  // The exact, slow form (keys are unique: the index is part of the key).  Also the way out when the fast form found FEWER than k
  // candidates, which takes NaNs in the row (block maxima skip them, `mq >= T` skips their blocks): then every block is read and every
  // admissible element is a candidate, so that -- as in the two-launch path -- the k indices handed on are always real columns.
  const bool all = n < k;
  if (all) T = -INFINITY;
  const int nb_read = all ? nblk : n_sel;
  unsigned long long last = ~0ull;
  for (int jr = 0; jr < k; ++jr) {
    unsigned long long best = 0ull;
    if (last != 0ull) {
      for (int b = 0; b < nb_read; ++b) {
        const int c = (all ? b : (int)sel[b]) * HS_COLS + lane * 4;
        if (c >= V) continue;
        const f32x4 x = *reinterpret_cast<const f32x4*>(row + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned long long key = (!(x[e] < T) && c + e != ban) ? topk_key(x[e], c + e) : 0ull;
          key = key < last ? key : 0ull;
          best = key > best ? key : best;
        }
      }
      best = wave_max_u64(best);
      last = best;
    }
    if (lane == 0) {
      val_row[jr] = ((best != 0ull ? topk_key_value(best) : -INFINITY) - lse) + a;
      idx_row[jr] = best != 0ull ? 0x7fffffff - (int)(uint32_t)best : 0x7fffffff;
    }
  }
}

__global__ __launch_bounds__(1024) void beam_stats_merge_kernel(const float* __restrict__ logits, int ldv, const float* __restrict__ stats,
                                                                int nblk, const float* __restrict__ add, int force_token, int ban, int nb,
                                                                int k, int V, int32_t* __restrict__ out, int eos,
                                                                float* __restrict__ next_scores, int64_t* __restrict__ next_tokens,
                                                                int32_t* __restrict__ next_beam_idx, const KmbHistGather hg,
                                                                const KmbEmbedNext en) {
  __shared__ float sval[256];
  __shared__ int32_t sidx[256];
  __shared__ int32_t snext[32];
  __shared__ unsigned long long cand[16][HS_CAND];
  __shared__ uint16_t sel[16][HS_BLK_MAX + 8];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < nb)
    beam_stats_row(logits, ldv, V, stats, nblk, b * nb + wave, add, force_token, ban, k, sval + wave * k, sidx + wave * k, cand[wave],
                   sel[wave], lane);
  __syncthreads();
  if (wave == 0) {
    beam_merge_item(sval, sidx, b, lane, nb, k, V, out, eos, next_scores, next_tokens, next_beam_idx, snext);
    beam_hist_gather(hg, snext, b, nb, lane);
  }
  beam_embed_next(en, snext, b, nb, wave, lane);
}


// ------------------------------------------------------------------------------------------------------------------
// Tied-head cross-entropy WITHOUT a pass over the logits (round 3).  Reference src/model/model.py:397-402:
//   loss = mean over valid rows of  lse(v_r) - v_r[label_r],   v_r = h_r E^T + b.
// The head GEMM (KmbGemm act 5) stores  P[r][j] = exp(v_rj - c_r)  in bf16 with c_r = the label's logit, so that
// P[r][label] = 1 exactly; a logit more than 80 above the label's (a row whose loss exceeds 80 nats) saturates at 2^115
// instead of overflowing (the epilogue clamps the exponent), ignored rows store zeros (their shift is +1e30); the GEMM
// leaves the fp32 row sums S_r = sum_j P[r][j] (per 64-column block) and d_r = v_r[label] - c_r.  Then
//   loss_r = log S_r - d_r,      softmax_rj = P[r][j] / S_r,
//   dlogits_rj = g (P[r][j] / S_r - [j == label_r]),          g = lm_factor / (number of valid rows)
// is never materialised: the row finish replaces the label's entry by P'[r][label_r] = P[r][label_r] - S_r (one bf16
// element per row: the difference rounded once, the relative error the two-kernel path's bf16 gradient row has there), and
// with a_r = g / S_r
//   dH_r  = a_r * sum_j P'[r][j] E_j            (the data-gradient GEMM runs on P', its slab finish applies a_r)
//   dE    = P'^T (a . H)                        (the weight-gradient GEMM runs on P' and the row-scaled copy a . H)
// so the 2 x 3.3 GB read-modify-write of ce_kernel_reg_bf16 (1.76 ms of a b = 1024 step) disappears; P is stored at
// 8 significant bits relative to the probability itself (the two-kernel path rounds the LOGIT to 8 bits: 3-6 % on p).
// (A first version kept P intact and corrected both products afterwards -- S_r E[label_r] in the finish and an atomic
// scatter of S_r (a . H)_r into dE: 115 us of fp32 atomics per b = 1024 step for the same precision.)

// c_r = h_r . E[label_r] + bias[label_r]; ignored / out-of-range rows get c_r = +1e30, so that their stored row is
// exp(v - 1e30) = 0 and their sum 0 whatever the logits are (nothing downstream has to rely on a zero factor); one wave per row
__global__ __launch_bounds__(256) void ce_label_logit_kernel(const bf16_t* __restrict__ H, int ldh, const bf16_t* __restrict__ E,
                                                             int lde, const float* __restrict__ bias,
                                                             const int64_t* __restrict__ labels, int rows, int d, int V,
                                                             float* __restrict__ shift) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const long long lab = labels[r];
  if (lab < 0 || lab >= V) {
    if (lane == 0) shift[r] = 1e30f;
    return;
  }
  const bf16_t* h = H + (size_t)r * ldh;
  const bf16_t* e = E + (size_t)lab * lde;
  float acc = 0.f;
  for (int i = lane * 8; i < d; i += 512) {
    float a[8], b[8];
    unpack8(*reinterpret_cast<const u32x4*>(h + i), a);
    unpack8(*reinterpret_cast<const u32x4*>(e + i), b);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = fmaf(a[k], b[k], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) shift[r] = acc + bias[lab];
}

// bias_pad[0, V) = bias, bias_pad[V, Vpad) = -1e30 (exp -> 0: the padded vocabulary columns contribute nothing)
__global__ __launch_bounds__(256) void ce_pad_bias_kernel(const float* __restrict__ bias, int V, int Vpad, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < Vpad) out[i] = i < V ? bias[i] : -1e30f;
}

// per row: S_r, loss_r, a_r, the row-scaled copy a . H (bf16), and the label's entry of the stored matrix turned into
// P[r, label_r] - S_r (so that a_r P'[r, :] IS the softmax gradient row and both gradient GEMMs need no correction term;
// the entry is exp(pick_r) ~ 1 against S_r >= 1: one bf16 rounding of the difference, the same relative error the bf16
// gradient row of the two-kernel path carries at the label); one wave per row
__global__ __launch_bounds__(256) void ce_rows_finish_kernel(const float* __restrict__ row_sums, int ld_sums, int nparts,
                                                             const float* __restrict__ pick, const int64_t* __restrict__ labels,
                                                             const int32_t* __restrict__ count, float lm_factor, int rows, int d, int V,
                                                             const bf16_t* __restrict__ H, int ldh, float* __restrict__ loss_rows,
                                                             float* __restrict__ srow, float* __restrict__ alpha,
                                                             bf16_t* __restrict__ ah, bf16_t* __restrict__ P, int ldp) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const long long lab = labels[r];
  const bool valid = lab != -100 && lab >= 0 && lab < V;
  float s = 0.f;
  for (int i = lane; i < nparts; i += 64) s += row_sums[(size_t)r * ld_sums + i];
  s = wave_sum(s);
  const int n = count[0];
  const float a = (valid && n > 0 && s > 0.f) ? lm_factor / ((float)n * s) : 0.f;
  if (lane == 0) {
    const float pk = pick != nullptr ? pick[r] : 0.f;   // v_r[label] - c_r: 0 up to summation order when c_r is the label's logit
    loss_rows[r] = valid ? __logf(s) - pk : 0.f;
    srow[r] = valid ? s : 0.f;
    alpha[r] = a;
    if (valid && P != nullptr) P[(size_t)r * ldp + lab] = f2bf(__expf(pk) - s);
  }
  if (ah != nullptr) {
    for (int i = lane * 8; i < d; i += 512) {
      float x[8];
      unpack8(*reinterpret_cast<const u32x4*>(H + (size_t)r * ldh + i), x);
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] *= a;
      *reinterpret_cast<u32x4*>(ah + (size_t)r * d + i) = pack8(x);
    }
  }
}

// dH_r = a_r * sum_s slab[s][r]  (the label's correction is inside the matrix the GEMM multiplied);  8 columns per thread
__global__ __launch_bounds__(256) void ce_dgrad_finish_kernel(const float* __restrict__ slab, int nslabs, size_t stride,
                                                              const float* __restrict__ alpha, bf16_t* __restrict__ out, int rows, int d) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;   // chunk of 8 columns
  const int per_row = d >> 3;
  const int r = (int)(idx / per_row);
  if (r >= rows) return;
  const int c = (int)(idx - (size_t)r * per_row) * 8;
  const float a = alpha[r];
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (a != 0.f) {
    const float* src = slab + (size_t)r * d + c;
    f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
    add_slabs2<4>(lo, hi, src, stride, nslabs);
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= a;
  }
  *reinterpret_cast<u32x4*>(out + (size_t)r * d + c) = pack8(v);
}

}  // namespace

hipError_t kmb_ce_label_logit_launch(const bf16_t* H, int ldh, const bf16_t* E, int lde, const float* bias, const int64_t* labels,
                                     int rows, int d, int V, float* shift, hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  if ((d & 7) || (ldh & 7) || (lde & 7)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ce_label_logit_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, H, ldh, E, lde, bias, labels, rows, d, V, shift);
  return hipGetLastError();
}
hipError_t kmb_ce_pad_bias_launch(const float* bias, int V, int Vpad, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(ce_pad_bias_kernel, dim3((Vpad + 255) / 256), dim3(256), 0, stream, bias, V, Vpad, out);
  return hipGetLastError();
}
hipError_t kmb_ce_rows_finish_launch(const float* row_sums, int ld_sums, int nparts, const float* pick, const int64_t* labels,
                                     const int32_t* count, float lm_factor, int rows, int d, int V, const bf16_t* H, int ldh,
                                     float* loss_rows, float* srow, float* alpha, bf16_t* ah, bf16_t* P, int ldp, hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  if ((d & 7) || (ldh & 7)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ce_rows_finish_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, row_sums, ld_sums, nparts, pick, labels, count,
                     lm_factor, rows, d, V, H, ldh, loss_rows, srow, alpha, ah, P, ldp);
  return hipGetLastError();
}
hipError_t kmb_ce_dgrad_finish_launch(const float* slab, int nslabs, size_t stride, const float* alpha, bf16_t* out, int rows, int d,
                                      hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  if ((d & 7) || (stride & 3)) return hipErrorInvalidValue;
  const size_t chunks = (size_t)rows * (d >> 3);
  hipLaunchKernelGGL(ce_dgrad_finish_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, stream, slab, nslabs, stride, alpha,
                     out, rows, d);
  return hipGetLastError();
}

hipError_t kmb_count_valid_launch(const int64_t* labels, int n, int V, int32_t* count, int32_t* status, hipStream_t stream) {
  hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(1024), 0, stream, labels, n, V, count, status);
  return hipGetLastError();
}

hipError_t kmb_ce_launch(const float* logits, int ldv, int V, const int64_t* labels, int rows, const int32_t* count,
                         float grad_scale, float* loss_rows, bf16_t* dlogits, hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  if ((ldv & 7) || ((uintptr_t)logits & 15)) return hipErrorInvalidValue;
  if (ldv <= 13 * 4096)
    hipLaunchKernelGGL((ce_kernel_reg<13>), dim3(rows), dim3(1024), 0, stream, logits, ldv, V, labels, count, grad_scale, loss_rows, dlogits);
  else
    hipLaunchKernelGGL(ce_kernel, dim3(rows), dim3(256), 0, stream, logits, ldv, V, labels, count, grad_scale, loss_rows, dlogits);
  return hipGetLastError();
}

hipError_t kmb_ce_bf16_launch(const bf16_t* logits, int ldv, int V, const int64_t* labels, int rows, const int32_t* count,
                              float grad_scale, float* loss_rows, bf16_t* dlogits, hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  if ((ldv & 7) || ((uintptr_t)logits & 15) || ldv > 8 * 8192) return hipErrorInvalidValue;
  if (ldv <= 7 * 8192)
    hipLaunchKernelGGL((ce_kernel_reg_bf16<7>), dim3(rows), dim3(1024), 0, stream, logits, ldv, V, labels, count, grad_scale, loss_rows, dlogits);
  else
    hipLaunchKernelGGL((ce_kernel_reg_bf16<8>), dim3(rows), dim3(1024), 0, stream, logits, ldv, V, labels, count, grad_scale, loss_rows, dlogits);
  return hipGetLastError();
}

hipError_t kmb_loss_finish_launch(const float* loss_rows, int rows, const int32_t* count, float* loss,
                                  hipStream_t stream) {
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1024), 0, stream, loss_rows, rows, count, loss);
  return hipGetLastError();
}

size_t kmb_logsoftmax_topk_scratch_floats(int rows) { return (size_t)(rows > 0 ? rows : 0) * TOPK_PARTS * TOPK_PW; }

hipError_t kmb_logsoftmax_topk_launch(const float* logits, int ldv, int V, int rows, const float* add,
                                      int force_token, int ban_token, int k, float* out_val, int32_t* out_idx,
                                      float* scratch, size_t scratch_floats, hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  const int chunks = (ldv / 4 + TOPK_PARTS - 1) / TOPK_PARTS;
  if (scratch != nullptr && scratch_floats >= kmb_logsoftmax_topk_scratch_floats(rows) && k >= 1 && k <= TOPK_KMAX &&
      (ldv & 3) == 0 && ((uintptr_t)logits & 15) == 0 && chunks <= 13 * 256 && rows <= 65535) {
    if (force_token < 0)
      hipLaunchKernelGGL((topk_part_kernel<13>), dim3(TOPK_PARTS, rows), dim3(256), 0, stream, logits, ldv, V, ban_token, k, chunks, scratch);
    hipLaunchKernelGGL(topk_combine_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, scratch, rows, add, force_token, k, out_val, out_idx);
    return hipGetLastError();
  }
  if (ldv <= 13 * 4096 && (ldv & 3) == 0 && ((uintptr_t)logits & 15) == 0)
    hipLaunchKernelGGL((logsoftmax_topk_reg_kernel<13>), dim3(rows), dim3(1024), 0, stream, logits, ldv, V, add, force_token, ban_token, k, out_val, out_idx);
  else
    hipLaunchKernelGGL(logsoftmax_topk_kernel, dim3(rows), dim3(256), 0, stream, logits, ldv, V, add, force_token, ban_token, k, out_val, out_idx);
  return hipGetLastError();
}

// log-softmax top-k of every beam row + the per-item merge + the next step's beams: topk_part (unless the token is forced) and ONE
// launch for the rest.  hipErrorNotSupported: the shape needs the separate launches (kmb_logsoftmax_topk_launch + kmb_beam_merge_launch).
hipError_t kmb_beam_step_launch(const float* logits, int ldv, int V, int B, int nb, const float* add, int force_token, int ban_token,
                                int k, int32_t* out, int eos, float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx,
                                float* scratch, size_t scratch_floats, hipStream_t stream, const KmbHistGather* hist,
                                const KmbEmbedNext* embed) {
  if (B <= 0) return hipSuccess;
  const int rows = B * nb;
  const int chunks = (ldv / 4 + TOPK_PARTS - 1) / TOPK_PARTS;
  if (!(scratch != nullptr && scratch_floats >= kmb_logsoftmax_topk_scratch_floats(rows) && k >= 1 && k <= TOPK_KMAX && nb >= 1 &&
        nb <= 16 && nb * k <= 256 && (ldv & 3) == 0 && ((uintptr_t)logits & 15) == 0 && chunks <= 13 * 256 && rows <= 65535))
    return hipErrorNotSupported;
  if (next_scores != nullptr && (!next_tokens || !next_beam_idx)) return hipErrorInvalidValue;
  const KmbHistGather hg = hist != nullptr && next_scores != nullptr ? *hist : KmbHistGather{nullptr, nullptr, 0, 0};
  KmbEmbedNext en = embed != nullptr && next_scores != nullptr ? *embed : KmbEmbedNext{};
  if (en.E != nullptr && ((en.D & 7) || en.D > 1024 || en.D <= 512)) return hipErrorInvalidValue;   // embed_ln_row<2>
  if (force_token < 0)
    hipLaunchKernelGGL((topk_part_kernel<13>), dim3(TOPK_PARTS, rows), dim3(256), 0, stream, logits, ldv, V, ban_token, k, chunks, scratch);
  hipLaunchKernelGGL(beam_combine_merge_kernel, dim3(B), dim3(64 * nb), 0, stream, scratch, add, force_token, nb, k, V, out, eos,
                     next_scores, next_tokens, next_beam_idx, hg, en);
  return hipGetLastError();
}

// The same step from the all-rows projection's per-block statistics (gemm.hip kmb_gemm_allrows_launch with stats): one launch.
// hipErrorNotSupported: the shape needs kmb_beam_step_launch.
hipError_t kmb_beam_step_stats_launch(const float* logits, int ldv, int V, int B, int nb, const float* add, int force_token, int ban_token,
                                      int k, int32_t* out, int eos, float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx,
                                      const float* stats, int nblk, hipStream_t stream, const KmbHistGather* hist,
                                      const KmbEmbedNext* embed) {
  if (B <= 0) return hipSuccess;
  if (!(stats != nullptr && nblk >= 1 && nblk <= HS_BLK_MAX && nblk == (V + HS_COLS - 1) / HS_COLS && B * nb <= HS_ROWS && k >= 1 &&
        k <= TOPK_KMAX && nb >= 1 && nb <= 16 && nb * k <= 256 && (ldv & 3) == 0 && (V & 3) == 0 && ldv >= V &&
        ((uintptr_t)logits & 15) == 0))
    return hipErrorNotSupported;
  if (next_scores != nullptr && (!next_tokens || !next_beam_idx)) return hipErrorInvalidValue;
  const KmbHistGather hg = hist != nullptr && next_scores != nullptr ? *hist : KmbHistGather{nullptr, nullptr, 0, 0};
  KmbEmbedNext en = embed != nullptr && next_scores != nullptr ? *embed : KmbEmbedNext{};
  if (en.E != nullptr && ((en.D & 7) || en.D > 1024 || en.D <= 512)) return hipErrorInvalidValue;   // embed_ln_row<2>
  hipLaunchKernelGGL(beam_stats_merge_kernel, dim3(B), dim3(64 * nb), 0, stream, logits, ldv, stats, nblk, add, force_token, ban_token, nb,
                     k, V, out, eos, next_scores, next_tokens, next_beam_idx, hg, en);
  return hipGetLastError();
}

hipError_t kmb_beam_merge_launch(const float* val, const int32_t* idx, int B, int nb, int k, int V, int32_t* out, int eos,
                                 float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx, hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  if (nb * k > 256 || k <= 0) return hipErrorInvalidValue;
  if (next_scores != nullptr && (!next_tokens || !next_beam_idx)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(64), 0, stream, val, idx, nb, k, V, out, eos, next_scores, next_tokens,
                     next_beam_idx);
  return hipGetLastError();
}
