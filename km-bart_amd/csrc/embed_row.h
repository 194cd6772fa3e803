// One wave: a row of the multimodal embedding -- (token | region) vector * scale + learned position, LayerNorm (+ dropout) -- shared by
// embed.hip's embed_ln_fwd_kernel and by the decode loop's beam step (loss.hip), which embeds the tokens it has just chosen for the next
// decode step in the same launch (round 6).  Same code, same order of operations: bit-identical rows either way.
// Reference: src/model/modules.py:89-102 (_embed_multi_modal), :133-137 (pos + LN + dropout).
#pragma once
#include "common.h"
#include "kernels.h"

template <int NCH>
__device__ __forceinline__ void embed_ln_row(const float* __restrict__ erow, const float* __restrict__ prow, float scale,
                                             const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ z,
                                             bf16_t* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int row, int D,
                                             float eps, KmbDrop drop, int lane) {
  const int nch = D >> 3;
  float v[NCH][8];
  float s = 0.f;
  // all loads first, from clamped addresses and outside any branch (see ln_fwd_kernel in norm.hip)
  f32x4 ev[NCH][2], pv[NCH][2], gv[NCH][2], bv[NCH][2];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i < nch ? lane + 64 * i : 0;
    ev[i][0] = *reinterpret_cast<const f32x4*>(erow + c * 8); ev[i][1] = *reinterpret_cast<const f32x4*>(erow + c * 8 + 4);
    pv[i][0] = *reinterpret_cast<const f32x4*>(prow + c * 8); pv[i][1] = *reinterpret_cast<const f32x4*>(prow + c * 8 + 4);
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i < nch ? lane + 64 * i : 0;
    gv[i][0] = *reinterpret_cast<const f32x4*>(gamma + c * 8); gv[i][1] = *reinterpret_cast<const f32x4*>(gamma + c * 8 + 4);
    bv[i][0] = *reinterpret_cast<const f32x4*>(beta + c * 8); bv[i][1] = *reinterpret_cast<const f32x4*>(beta + c * 8 + 4);
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const f32x4 e0 = ev[i][0], e1 = ev[i][1], p0 = pv[i][0], p1 = pv[i][1];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[i][e] = e0[e] * scale + p0[e]; v[i][4 + e] = e1[e] * scale + p1[e]; }
      if (z != nullptr) *reinterpret_cast<u32x4*>(z + (size_t)row * D + c * 8) = pack8(v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (lane + 64 * i < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
    }
  const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0 && mean != nullptr) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = (v[i][e] - mu) * rs * gv[i][e >> 2][e & 3] + bv[i][e >> 2][e & 3];
        if (drop.thr16 != 0u)
          o[e] = drop_keep(drop.seed, (uint32_t)row, (uint32_t)(c * 8 + e), drop.thr16) ? o[e] * drop.scale : 0.f;
      }
      *reinterpret_cast<u32x4*>(y + (size_t)row * D + c * 8) = pack8(o);
    }
  }
}
