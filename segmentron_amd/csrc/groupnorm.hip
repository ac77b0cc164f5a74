// nn.GroupNorm(min(32, C), C) — the `GN` choice of cfg.MODEL.BN_TYPE
// (/root/reference/segmentron/modules/batch_norm.py:105-108,129), forward and backward, NHWC.
//
// Group statistics are per SAMPLE, so the normalisation is a per-(sample, channel) affine
//     z = x * a[n][c] + b[n][c],   a = gamma[c] * rstd[n][g],  b = beta[c] - mean[n][g] * a
// and cannot ride in the per-channel prologue of the consuming convolution the way BatchNorm does
// (functional.py Act): GroupNorm layers are materialised.  Four kernels serve both directions:
//   gn_moments   per (sample, pixel chunk, channel): (sum u, sum u*v) — v = u forward (sum x,
//                sum x^2), (u, v) = (dz, x) backward; fp32 partial rows [N][chunks][2][C]
//   gn_fwd_finalize / gn_bwd_finalize   one block per (sample, group): float64 sums over the
//                chunks and the group's channels -> the coefficient rows [N][3][C]
//   gn_affine    out = k1[n][c] * u + k2[n][c] * v + k3[n][c]
// Backward (m = H*W*C/G elements per group, xh = (x - mean) * rstd, sums over the group):
//     A = sum dz*gamma, B = sum dz*gamma*xh
//     dx = rstd*gamma*dz - (rstd^2 * B / m) * x + (rstd^2 * B * mean - rstd * A) / m
//     dgamma[c] = sum_n rstd * (S2[n][c] - mean * S1[n][c]),  dbeta[c] = sum_n S1[n][c]
// (S1 = sum_p dz, S2 = sum_p dz*x per sample and channel; the sums over n: seg_colsum of the
// contribution rows this file writes).
#include "common.h"

namespace seg {

constexpr int GN_THREADS = 256, GN_COLS = 64, GN_ROWS = GN_THREADS / GN_COLS;

template <typename T, bool TWO>
__global__ __launch_bounds__(GN_THREADS) void gn_moments_kernel(const T* __restrict__ u, long ldu,
                                                                const T* __restrict__ v, long ldv,
                                                                long HW, int C, int chunks,
                                                                float* __restrict__ partial) {
  constexpr int VEC = Vec<T>::N;
  const int cx = blockIdx.z * GN_COLS + (threadIdx.x % GN_COLS), ry = threadIdx.x / GN_COLS;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const long rows = (HW + chunks - 1) / chunks;
  const long r0 = chunk * rows, r1 = min(HW, r0 + rows);
  const bool live = cx * VEC < C;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) s1[k] = s2[k] = 0.f;
  if (live) {
    const T* pu = u + (n * HW) * ldu + cx * VEC;
    const T* pv = TWO ? v + (n * HW) * ldv + cx * VEC : nullptr;
    for (long r = r0 + ry; r < r1; r += GN_ROWS) {
      float a[VEC], b[VEC];
      Vec<T>::unpack(Vec<T>::load_raw(pu + r * ldu), a);
      if (TWO) Vec<T>::unpack(Vec<T>::load_raw(pv + r * ldv), b);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        s1[k] += a[k];
        s2[k] = fmaf(a[k], TWO ? b[k] : a[k], s2[k]);
      }
    }
  }
  __shared__ float red[GN_ROWS][2][GN_COLS][VEC + 1];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    red[ry][0][threadIdx.x % GN_COLS][k] = s1[k];
    red[ry][1][threadIdx.x % GN_COLS][k] = s2[k];
  }
  __syncthreads();
  if (ry == 0 && live) {
    float* dst = partial + ((long)(n * chunks + chunk) * 2) * C + cx * VEC;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int j = 0; j < GN_ROWS; ++j) {
        t1 += red[j][0][threadIdx.x][k];
        t2 += red[j][1][threadIdx.x][k];
      }
      dst[k] = t1;
      dst[C + k] = t2;
    }
  }
}

// wave-wide sum of a double (64 lanes)
__device__ __forceinline__ double gn_wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// one wave per (sample, group)
__global__ __launch_bounds__(64) void gn_fwd_finalize_kernel(const float* __restrict__ partial,
                                                             long HW, int C, int G, int chunks,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             double eps,
                                                             float* __restrict__ mean_rstd,
                                                             float* __restrict__ coef) {
  const int n = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G, lane = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int e = lane; e < chunks * cpg; e += 64) {
    const int ch = e / cpg, c = g * cpg + e % cpg;
    const float* row = partial + ((long)(n * chunks + ch) * 2) * C;
    s1 += row[c];
    s2 += row[C + c];
  }
  s1 = gn_wave_sum(s1);
  s2 = gn_wave_sum(s2);
  const double m = (double)HW * cpg;
  const double mean = s1 / m;
  double var = s2 / m - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const double rstd = 1.0 / sqrt(var + eps);
  if (lane == 0) {
    mean_rstd[(n * G + g) * 2] = (float)mean;
    mean_rstd[(n * G + g) * 2 + 1] = (float)rstd;
  }
  float* k = coef + (long)n * 3 * C;
  for (int j = lane; j < cpg; j += 64) {
    const int c = g * cpg + j;
    const double a = (gamma ? (double)gamma[c] : 1.0) * rstd;
    k[c] = (float)a;
    k[C + c] = 0.f;
    k[2 * C + c] = (float)((beta ? (double)beta[c] : 0.0) - mean * a);
  }
}

__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(const float* __restrict__ partial,
                                                             long HW, int C, int G, int chunks,
                                                             const float* __restrict__ mean_rstd,
                                                             const float* __restrict__ gamma,
                                                             float* __restrict__ coef,
                                                             float* __restrict__ contrib) {
  const int n = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G, lane = threadIdx.x;
  const double mean = mean_rstd[(n * G + g) * 2], rstd = mean_rstd[(n * G + g) * 2 + 1];
  double A = 0.0, B = 0.0;
  for (int j = lane; j < cpg; j += 64) {  // (cpg <= 64 for every width of the model zoo: one trip)
    const int c = g * cpg + j;
    double s1 = 0.0, s2 = 0.0;
    for (int ch = 0; ch < chunks; ++ch) {
      const float* row = partial + ((long)(n * chunks + ch) * 2) * C;
      s1 += row[c];
      s2 += row[C + c];
    }
    const double gam = gamma ? (double)gamma[c] : 1.0;
    const double dxh = rstd * (s2 - mean * s1);  // sum_p dz * xh
    A += gam * s1;
    B += gam * dxh;
    contrib[(long)n * 2 * C + c] = (float)dxh;  // -> dgamma
    contrib[(long)n * 2 * C + C + c] = (float)s1;  // -> dbeta
  }
  A = gn_wave_sum(A);
  B = gn_wave_sum(B);
  const double m = (double)HW * cpg;
  const double k2 = -rstd * rstd * B / m, k3 = (rstd * rstd * B * mean - rstd * A) / m;
  float* k = coef + (long)n * 3 * C;
  for (int j = lane; j < cpg; j += 64) {
    const int c = g * cpg + j;
    k[c] = (float)((gamma ? (double)gamma[c] : 1.0) * rstd);
    k[C + c] = (float)k2;
    k[2 * C + c] = (float)k3;
  }
}

template <typename T, bool TWO>
__global__ __launch_bounds__(GN_THREADS) void gn_affine_kernel(const T* __restrict__ u, long ldu,
                                                               const T* __restrict__ v, long ldv,
                                                               const float* __restrict__ coef,
                                                               T* __restrict__ out, long ldo,
                                                               long HW, int C, int chunks) {
  constexpr int VEC = Vec<T>::N;
  const int cx = blockIdx.z * GN_COLS + (threadIdx.x % GN_COLS), ry = threadIdx.x / GN_COLS;
  const int n = blockIdx.y, chunk = blockIdx.x;
  if (cx * VEC >= C) return;
  const long rows = (HW + chunks - 1) / chunks;
  const long r0 = chunk * rows, r1 = min(HW, r0 + rows);
  float k1[VEC], k2[VEC], k3[VEC];
  const float* k = coef + (long)n * 3 * C + cx * VEC;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    k1[j] = k[j];
    k2[j] = k[C + j];
    k3[j] = k[2 * C + j];
  }
  const T* pu = u + (n * HW) * ldu + cx * VEC;
  const T* pv = TWO ? v + (n * HW) * ldv + cx * VEC : nullptr;
  T* po = out + (n * HW) * ldo + cx * VEC;
  for (long r = r0 + ry; r < r1; r += GN_ROWS) {
    float a[VEC], b[VEC], o[VEC];
    Vec<T>::unpack(Vec<T>::load_raw(pu + r * ldu), a);
    if (TWO) Vec<T>::unpack(Vec<T>::load_raw(pv + r * ldv), b);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      o[j] = fmaf(k1[j], a[j], k3[j]);
      if (TWO) o[j] = fmaf(k2[j], b[j], o[j]);
    }
    Vec<T>::store(po + r * ldo, o);
  }
}

static int gn_chunks(long HW) {
  long c = (HW + 63) / 64;  // >= 64 pixel rows per chunk
  return (int)(c < 1 ? 1 : (c > 128 ? 128 : c));
}

template <typename T>
static int gn_check(const char* what, long ld, int C) {
  constexpr int VEC = Vec<T>::N;
  SEG_REQUIRE(C % VEC == 0 && ld % VEC == 0, "%s: C = %d and the row pitch %ld must be multiples of %d",
              what, C, ld, VEC);
  return 0;
}

}  // namespace seg

extern "C" int seg_gn_chunks(long HW) { return seg::gn_chunks(HW); }

extern "C" int seg_gn_moments(int dtype, const void* u, long ldu, const void* v, long ldv, int N,
                              long HW, int C, float* partial, void* stream) {
  using namespace seg;
  const int chunks = gn_chunks(HW);
  const int vec = dtype == DT_BF16 ? 8 : 4;
  const dim3 grid(chunks, N, (C / vec + GN_COLS - 1) / GN_COLS), block(GN_THREADS);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DT_BF16) {
    if (gn_check<bf16_t>("seg_gn_moments", ldu, C) || (v && gn_check<bf16_t>("seg_gn_moments", ldv, C))) return 1;
    if (v) hipLaunchKernelGGL((gn_moments_kernel<bf16_t, true>), grid, block, 0, s, (const bf16_t*)u, ldu, (const bf16_t*)v, ldv, HW, C, chunks, partial);
    else hipLaunchKernelGGL((gn_moments_kernel<bf16_t, false>), grid, block, 0, s, (const bf16_t*)u, ldu, nullptr, 0L, HW, C, chunks, partial);
  } else {
    if (gn_check<float>("seg_gn_moments", ldu, C) || (v && gn_check<float>("seg_gn_moments", ldv, C))) return 1;
    if (v) hipLaunchKernelGGL((gn_moments_kernel<float, true>), grid, block, 0, s, (const float*)u, ldu, (const float*)v, ldv, HW, C, chunks, partial);
    else hipLaunchKernelGGL((gn_moments_kernel<float, false>), grid, block, 0, s, (const float*)u, ldu, nullptr, 0L, HW, C, chunks, partial);
  }
  return check_launch("seg_gn_moments");
}

extern "C" int seg_gn_fwd_finalize(const float* partial, int N, long HW, int C, int G,
                                   const float* gamma, const float* beta, double eps,
                                   float* mean_rstd, float* coef, void* stream) {
  using namespace seg;
  SEG_REQUIRE(G > 0 && C % G == 0, "seg_gn_fwd_finalize: %d channels are not divisible into %d groups", C, G);
  hipLaunchKernelGGL(gn_fwd_finalize_kernel, dim3(N * G), dim3(64), 0, (hipStream_t)stream, partial,
                     HW, C, G, gn_chunks(HW), gamma, beta, eps, mean_rstd, coef);
  return check_launch("seg_gn_fwd_finalize");
}

extern "C" int seg_gn_bwd_finalize(const float* partial, int N, long HW, int C, int G,
                                   const float* mean_rstd, const float* gamma, float* coef,
                                   float* contrib, void* stream) {
  using namespace seg;
  SEG_REQUIRE(G > 0 && C % G == 0, "seg_gn_bwd_finalize: %d channels are not divisible into %d groups", C, G);
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(N * G), dim3(64), 0, (hipStream_t)stream, partial,
                     HW, C, G, gn_chunks(HW), mean_rstd, gamma, coef, contrib);
  return check_launch("seg_gn_bwd_finalize");
}

extern "C" int seg_gn_affine(int dtype, const void* u, long ldu, const void* v, long ldv,
                             const float* coef, void* out, long ldo, int N, long HW, int C,
                             void* stream) {
  using namespace seg;
  const int chunks = gn_chunks(HW);
  const int vec = dtype == DT_BF16 ? 8 : 4;
  const dim3 grid(chunks, N, (C / vec + GN_COLS - 1) / GN_COLS), block(GN_THREADS);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DT_BF16) {
    if (gn_check<bf16_t>("seg_gn_affine", ldu, C) || gn_check<bf16_t>("seg_gn_affine", ldo, C) || (v && gn_check<bf16_t>("seg_gn_affine", ldv, C))) return 1;
    if (v) hipLaunchKernelGGL((gn_affine_kernel<bf16_t, true>), grid, block, 0, s, (const bf16_t*)u, ldu, (const bf16_t*)v, ldv, coef, (bf16_t*)out, ldo, HW, C, chunks);
    else hipLaunchKernelGGL((gn_affine_kernel<bf16_t, false>), grid, block, 0, s, (const bf16_t*)u, ldu, nullptr, 0L, coef, (bf16_t*)out, ldo, HW, C, chunks);
  } else {
    if (gn_check<float>("seg_gn_affine", ldu, C) || gn_check<float>("seg_gn_affine", ldo, C) || (v && gn_check<float>("seg_gn_affine", ldv, C))) return 1;
    if (v) hipLaunchKernelGGL((gn_affine_kernel<float, true>), grid, block, 0, s, (const float*)u, ldu, (const float*)v, ldv, coef, (float*)out, ldo, HW, C, chunks);
    else hipLaunchKernelGGL((gn_affine_kernel<float, false>), grid, block, 0, s, (const float*)u, ldu, nullptr, 0L, coef, (float*)out, ldo, HW, C, chunks);
  }
  return check_launch("seg_gn_affine");
}
