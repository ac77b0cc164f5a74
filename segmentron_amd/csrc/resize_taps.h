// Source-index arithmetic of bilinear resampling, shared by resize.hip and loss.hip: the same
// float32 formulas as ATen's upsample_bilinear2d.
#pragma once
#include "common.h"

namespace seg {

__device__ __forceinline__ float src_index(float scale, int dst, int align) {
  if (align) return scale * (float)dst;
  const float s = scale * ((float)dst + 0.5f) - 0.5f;
  return s < 0.f ? 0.f : s;
}
__device__ __forceinline__ void taps(float scale, int dst, int in, int align, int& i0, int& i1,
                                     float& lam) {
  const float s = src_index(scale, dst, align);
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  lam = s - (float)i0;
}
static inline float host_scale(int in, int out, int align) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}
// candidate output range [lo, hi] whose taps may touch input index i
// (align_corners=False: src = scale*(dst+0.5)-0.5, so index i is touched up to
//  dst < (i+1.5)/scale - 0.5 — half a source pixel further than in the aligned mapping)
__device__ __forceinline__ void cand_range(float scale, int i, int out, int align, int& lo,
                                           int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
  const float inv = 1.f / scale;
  lo = (int)floorf(((float)i - 1.f) * inv) - 1;
  hi = (int)ceilf(((float)i + (align ? 1.f : 1.5f)) * inv) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}
__device__ __forceinline__ float tap_weight(float scale, int dst, int in, int align, int i) {
  int i0, i1; float lam;
  taps(scale, dst, in, align, i0, i1, lam);
  float w = 0.f;
  if (i0 == i) w += 1.f - lam;
  if (i1 == i) w += lam;
  return w;
}


}  // namespace seg
