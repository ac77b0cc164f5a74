// Folding a LINEAR pending BatchNorm into a 1x1 convolution.
// In `relu_first` separable convs (segmentron/modules/basic.py:46-50: relu -> depthwise -> bn_depth
// -> pointwise -> bn_point) nothing non-linear sits between bn_depth and the pointwise conv, so
//     W (s .* x + t) = (W diag(s)) x + W t
// The forward GEMM and the weight-gradient GEMM then run on the RAW depthwise output with no
// per-element prologue work at all, and BatchNorm backward needs no pass over the activation:
//     dW' = dY^T x_raw ;  dW = dW' diag(s) (+ db (x) t) ;  ds = colsum(W .* dW') ;  dt = W^T db
//     dgamma = invstd (ds - mean dt) ; dbeta = dt
//     dx_raw = dY W'  -  c0 - c1 x_raw ,  c1 = gamma (ds - mean dt) invstd^3 / n ,
//                                          c0 = dt s / n - c1 mean
// (63 of the 79 GEMM-shaped convolutions of DeepLabv3+/xception65, ~80 % of its FLOPs.)
#include "common.h"

namespace seg {

// one wave per output row o
template <typename T>
__global__ __launch_bounds__(256) void fold_weights_kernel(const float* __restrict__ W,
                                                           const float* __restrict__ s,
                                                           const float* __restrict__ t,
                                                           T* __restrict__ Wp, T* __restrict__ WpT,
                                                           float* __restrict__ bprime, int O,
                                                           int C) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= O) return;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float w = W[(long)o * C + c];
    const float ws = w * s[c];
    acc = fmaf(w, t[c], acc);
    Vec<T>::store1(Wp + (long)o * C + c, ws);
    if (WpT) Vec<T>::store1(WpT + (long)c * O + o, ws);
  }
  acc = wave_sum(acc);
  if (lane == 0 && bprime) bprime[o] = acc;
}

// block: 8 columns (c) x 32 row groups (o) -> C/8 blocks, O/32 serial iterations
__global__ __launch_bounds__(256) void fold_bwd_reduce_kernel(
    const float* __restrict__ W, const float* __restrict__ dWp, const float* __restrict__ s,
    const float* __restrict__ t, const float* __restrict__ db, float* __restrict__ dW,
    float* __restrict__ dsdt, int O, int C) {
  __shared__ float red[2][32][9];
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cx;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    const float sc = s[c], tc = t[c];
    for (int o = ry; o < O; o += 32) {
      const float w = W[(long)o * C + c], g = dWp[(long)o * C + c];
      const float dbo = db ? db[o] : 0.f;
      dW[(long)o * C + c] = fmaf(g, sc, dbo * tc);
      a0 = fmaf(w, g, a0);
      a1 = fmaf(w, dbo, a1);
    }
  }
  red[0][ry][cx] = a0;
  red[1][ry][cx] = a1;
  __syncthreads();
  if (ry == 0 && c < C) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      s0 += red[0][k][cx];
      s1 += red[1][k][cx];
    }
    dsdt[c] = s0;
    dsdt[C + c] = s1;
  }
}

__global__ void fold_bwd_finalize_kernel(const float* __restrict__ dsdt, double count,
                                         const float* __restrict__ mean,
                                         const float* __restrict__ invstd,
                                         const float* __restrict__ gamma,
                                         const float* __restrict__ scale, float* dgamma,
                                         float* dbeta, float* c0, float* c1, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double ds = dsdt[c], dt = dsdt[C + c];
  const double mu = mean[c], is = invstd[c];
  const double g = gamma ? (double)gamma[c] : 1.0;
  const double u = ds - mu * dt;
  const double A = g * u * is * is * is / count;
  if (dgamma) dgamma[c] = (float)(is * u);
  if (dbeta) dbeta[c] = (float)dt;
  c1[c] = (float)A;
  c0[c] = (float)(dt * (double)scale[c] / count - A * mu);
}

}  // namespace seg

extern "C" int seg_fold_weights(int dtype, const float* W, const float* scale, const float* shift,
                                void* Wp, void* WpT, float* bprime, int O, int C, void* stream) {
  using namespace seg;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "fold_weights: bad dtype %d", dtype);
  SEG_REQUIRE(O >= 1 && C >= 1 && W && scale && shift && Wp, "fold_weights: bad arguments");
  const dim3 grid((O + 3) / 4);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((fold_weights_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, W,
                       scale, shift, (bf16_t*)Wp, (bf16_t*)WpT, bprime, O, C);
  else
    hipLaunchKernelGGL((fold_weights_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, W,
                       scale, shift, (float*)Wp, (float*)WpT, bprime, O, C);
  return check_launch("fold_weights");
}

extern "C" int seg_fold_bwd_reduce(const float* W, const float* dWp, const float* scale,
                                   const float* shift, const float* db, float* dW, float* dsdt,
                                   int O, int C, void* stream) {
  using namespace seg;
  SEG_REQUIRE(O >= 1 && C >= 1, "fold_bwd_reduce: empty");
  hipLaunchKernelGGL(fold_bwd_reduce_kernel, dim3((C + 7) / 8), dim3(256), 0,
                     (hipStream_t)stream, W, dWp, scale, shift, db, dW, dsdt, O, C);
  return check_launch("fold_bwd_reduce");
}

extern "C" int seg_fold_bwd_finalize(const float* dsdt, double count, const float* mean,
                                     const float* invstd, const float* gamma, const float* scale,
                                     float* dgamma, float* dbeta, float* c0, float* c1, int C,
                                     void* stream) {
  using namespace seg;
  SEG_REQUIRE(count >= 1.0 && C >= 1, "fold_bwd_finalize: bad count/C");
  hipLaunchKernelGGL(fold_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, dsdt, count, mean, invstd, gamma, scale, dgamma, dbeta,
                     c0, c1, C);
  return check_launch("fold_bwd_finalize");
}
