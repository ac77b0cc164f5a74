// Depthwise 3x3 convolution (groups == C), NHWC, any stride / dilation, pad = dil.
// Reference call sites: segmentron/modules/basic.py:38-40 (SeparableConv2d.depthwise),
// :152-153 (InvertedResidual dw).  68 instances in DeepLabv3+/xception65: 1.5 % of
// the MACs but as much HBM traffic as all 1x1 convs -> purely bandwidth-bound, so the kernel is
// built around 16-byte channel vectors (8 bf16 / 4 f32 per lane, consecutive lanes = consecutive
// channel vectors = fully coalesced NHWC rows) with the producer's BatchNorm(+ReLU) fused into
// the load path and the consumer BatchNorm's sum / sum-of-squares fused into the store path.
//
//   MODE_FWD   : y[n,ho,wo,c] = sum_{kh,kw} act(x[n, ho*s-p+kh*d, wo*s-p+kw*d, c]) * w[kh,kw,c]
//   MODE_DGRAD : dx[n,h,w,c]  = sum_{kh,kw} dy[n,(h+p-kh*d)/s,(w+p-kw*d)/s,c] * w[kh,kw,c]
//                (terms kept only when the division is exact and in range)
//   wgrad      : dw[kh,kw,c]  = sum_{n,ho,wo} dy[n,ho,wo,c] * act(x[n, ho*s-p+kh*d, ...,c])
//
// Thread layout: block = CVB channel-vectors (8/16/32, chosen by the host to fit C) x 256/CVB
// pixel strips; a strip is TW=4 consecutive output pixels of one row.  blockIdx.x tiles the
// channel vectors, blockIdx.y the strips (grid-stride), one row of statistics partials per
// blockIdx.y (deterministic: LDS tree over the strips of a block, fp64 finish in bn_finalize).
#include "common.h"
#include "dwconv_tiled.h"
#include "dwconv_slide.h"
#include <cstdlib>
#include <type_traits>

namespace seg {

constexpr bool g_dw_row = true;  // wide dilations on the row-chain kernels (dwconv_row.hip)

constexpr int DW_TW = 4;
constexpr int DW_THREADS = 256;
enum { MODE_FWD = 0, MODE_DGRAD = 1 };

struct DwArgs {
  const void* x;      // fwd: input; dgrad: dy
  const float* w;     // [9][C] fp32, tap-major
  void* y;            // fwd: output; dgrad: dx
  const float* pro_scale;
  const float* pro_shift;
  float* stat_partial;  // [gridDim.y][2][C] or null (fwd only)
  long ldx, ldy;
  int N, Hi, Wi, C;   // geometry of the tensor being READ
  int Ho, Wo;         // geometry of the tensor being WRITTEN
  int stride, pad, dil;
  int pro_mode;
  int cvb_log2;       // log2(CVB)
  int CV;             // C / VEC
  long strips;        // N * Ho * ceil(Wo / TW)
};

// q = s / d for 0 <= s < 2^24 via a float reciprocal and one fix-up (replaces a ~40-instruction
// integer division in the per-strip index decode)
__device__ __forceinline__ int fast_div(int s, int d, float inv) {
  int q = (int)((float)s * inv);
  if (q * d > s) --q;
  if ((q + 1) * d <= s) ++q;
  return q;
}

template <int VEC>
__device__ __forceinline__ void dw_act(float* f, int mode, const float* sc, const float* sh) {
  if (mode & PRO_AFFINE) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = fmaf(f[i], sc[i], sh[i]);
  }
  if (mode & PRO_RELU) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = fmaxf(f[i], 0.f);
  }
  if (mode & PRO_CLAMP6) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = fminf(f[i], 6.f);
  }
}

// Strip kernels: stride 2 and wide dilations (6/12/18) only — every stride-1, dil <= 2 layer runs
// on the LDS-tiled kernels of dwconv_tiled.hip (the sliding-window fast path these kernels once
// had for that case is gone with it).
template <typename T, int MODE>
__global__ __launch_bounds__(DW_THREADS, 2) void dwconv_kernel(const DwArgs a) {
  using V = Vec<T>;
  constexpr int VEC = V::N;
  extern __shared__ __attribute__((aligned(16))) float dw_smem[];
  const int tid = threadIdx.x;
  const int cvb = 1 << a.cvb_log2;
  const int cx = tid & (cvb - 1), sy = tid >> a.cvb_log2;
  const int spb = DW_THREADS >> a.cvb_log2;  // strips per block iteration
  const int cv = blockIdx.x * cvb + cx;
  const bool cok = cv < a.CV;
  const int c0 = cok ? cv * VEC : 0;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const int WQ = (a.Wo + DW_TW - 1) / DW_TW;
  const float inv_wq = 1.0f / (float)WQ, inv_ho = 1.0f / (float)a.Ho;

  float sc[VEC], sh[VEC];
  if (a.pro_mode & PRO_AFFINE) {
    load_params<VEC>(a.pro_scale, c0, sc);
    load_params<VEC>(a.pro_shift, c0, sh);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
  }
  float ssum[VEC], ssq[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) ssum[i] = ssq[i] = 0.f;

  const int nstrips = (int)a.strips;
  for (int s = blockIdx.y * spb + sy; s < nstrips; s += gridDim.y * spb) {
    if (!cok) continue;
    const int t = fast_div(s, WQ, inv_wq);
    const int wq = s - t * WQ;
    const int n = fast_div(t, a.Ho, inv_ho);
    const int ho = t - n * a.Ho;
    const int w0 = wq * DW_TW;
    float acc[DW_TW][VEC];
#pragma unroll
    for (int j = 0; j < DW_TW; ++j)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[j][i] = 0.f;

#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      int hi;
      if (MODE == MODE_FWD) {
        hi = ho * a.stride - a.pad + kh * a.dil;
      } else {
        const int hn = ho + a.pad - kh * a.dil;
        if (hn < 0 || (hn % a.stride) != 0) continue;
        hi = hn / a.stride;
      }
      if (hi < 0 || hi >= a.Hi) continue;
      const long rowbase = ((long)n * a.Hi + hi) * a.Wi;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        float wv[VEC];
        load_params<VEC>(a.w + (kh * 3 + kw) * a.C, c0, wv);
#pragma unroll
        for (int j = 0; j < DW_TW; ++j) {
          const int wo = w0 + j;
          int wi;
          bool ok = wo < a.Wo;
          if (MODE == MODE_FWD) {
            wi = wo * a.stride - a.pad + kw * a.dil;
          } else {
            const int wn = wo + a.pad - kw * a.dil;
            ok = ok && wn >= 0 && (wn % a.stride) == 0;
            wi = wn / a.stride;
          }
          ok = ok && wi >= 0 && wi < a.Wi;
          if (ok) {
            float f[VEC];
            V::unpack_raw(V::load_raw(X + (rowbase + wi) * a.ldx + c0), f);
            dw_act<VEC>(f, a.pro_mode, sc, sh);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[j][i] = fmaf(f[i], wv[i], acc[j][i]);
          }
        }
      }
    }
    const long orow = ((long)n * a.Ho + ho) * a.Wo;
#pragma unroll
    for (int j = 0; j < DW_TW; ++j) {
      if (w0 + j < a.Wo) {
        V::store(Y + (orow + w0 + j) * a.ldy + c0, acc[j]);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          ssum[i] += acc[j][i];
          ssq[i] += acc[j][i] * acc[j][i];
        }
      }
    }
  }

  if (a.stat_partial != nullptr) {
    // dw_smem: [spb][cvb][2*VEC]
    float* mine = dw_smem + ((long)sy * cvb + cx) * 2 * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      mine[i] = ssum[i];
      mine[VEC + i] = ssq[i];
    }
    __syncthreads();
    for (int e = tid; e < cvb * 2 * VEC; e += DW_THREADS) {
      float tot = 0.f;
      for (int r = 0; r < spb; ++r) tot += dw_smem[(long)r * cvb * 2 * VEC + e];
      const int lcx = e / (2 * VEC), k = e % (2 * VEC);
      const int which = k / VEC, ci = k % VEC;
      const int c = (blockIdx.x * cvb + lcx) * VEC + ci;
      if (c < a.C) a.stat_partial[((long)blockIdx.y * 2 + which) * a.C + c] = tot;
    }
  }
}

// ---- weight gradient: partial[blockIdx.y][9][C]
struct DwWgradArgs {
  const void* x;   // forward input (pre-activation raw tensor + prologue)
  const void* dy;  // grad wrt dw output
  float* partial;  // [gridDim.y][9][C]
  const float* pro_scale;
  const float* pro_shift;
  long ldx, lddy;
  int N, Hi, Wi, C, Ho, Wo;
  int stride, pad, dil;
  int pro_mode;
  int cvb_log2, CV;
  long strips;
};

template <typename T>
__global__ __launch_bounds__(DW_THREADS) void dwconv_wgrad_kernel(const DwWgradArgs a) {
  constexpr int VEC = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) float dw_smem[];
  const int tid = threadIdx.x;
  const int cvb = 1 << a.cvb_log2;
  const int cx = tid & (cvb - 1), sy = tid >> a.cvb_log2;
  const int spb = DW_THREADS >> a.cvb_log2;
  const int cv = blockIdx.x * cvb + cx;
  const bool cok = cv < a.CV;
  const int c0 = cok ? cv * VEC : 0;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ DY = reinterpret_cast<const T*>(a.dy);
  const int WQ = (a.Wo + DW_TW - 1) / DW_TW;
  const float inv_wq = 1.0f / (float)WQ, inv_ho = 1.0f / (float)a.Ho;

  float sc[VEC], sh[VEC];
  if (a.pro_mode & PRO_AFFINE) {
    load_params<VEC>(a.pro_scale, c0, sc);
    load_params<VEC>(a.pro_shift, c0, sh);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
  }
  float acc[9][VEC];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[k][i] = 0.f;

  const int nstrips = (int)a.strips;
  for (int s = blockIdx.y * spb + sy; s < nstrips; s += gridDim.y * spb) {
    if (!cok) continue;
    const int t = fast_div(s, WQ, inv_wq);
    const int wq = s - t * WQ;
    const int n = fast_div(t, a.Ho, inv_ho);
    const int ho = t - n * a.Ho;
    const int w0 = wq * DW_TW;
    const long orow = ((long)n * a.Ho + ho) * a.Wo;
    float g[DW_TW][VEC];
#pragma unroll
    for (int j = 0; j < DW_TW; ++j) {
      const int wc = min(w0 + j, a.Wo - 1);
      Vec<T>::unpack(ldg16(DY + (orow + wc) * a.lddy + c0), g[j]);
      const float keep = (w0 + j < a.Wo) ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) g[j][i] *= keep;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = ho * a.stride - a.pad + kh * a.dil;
      if (hi < 0 || hi >= a.Hi) continue;
      const long rowbase = ((long)n * a.Hi + hi) * a.Wi;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
        for (int j = 0; j < DW_TW; ++j) {
          const int wi = (w0 + j) * a.stride - a.pad + kw * a.dil;
          if (w0 + j < a.Wo && wi >= 0 && wi < a.Wi) {
            float f[VEC];
            Vec<T>::unpack(ldg16(X + (rowbase + wi) * a.ldx + c0), f);
            dw_act<VEC>(f, a.pro_mode, sc, sh);
#pragma unroll
            for (int i = 0; i < VEC; ++i)
              acc[kh * 3 + kw][i] = fmaf(f[i], g[j][i], acc[kh * 3 + kw][i]);
          }
        }
      }
    }
  }
  // block reduction over the strips, three taps at a time (LDS: [spb][cvb][3*VEC])
#pragma unroll
  for (int k3 = 0; k3 < 3; ++k3) {
    float* mine = dw_smem + ((long)sy * cvb + cx) * 3 * VEC;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
      for (int i = 0; i < VEC; ++i) mine[kk * VEC + i] = acc[k3 * 3 + kk][i];
    __syncthreads();
    for (int e = tid; e < cvb * 3 * VEC; e += DW_THREADS) {
      float tot = 0.f;
      for (int r = 0; r < spb; ++r) tot += dw_smem[(long)r * cvb * 3 * VEC + e];
      const int lcx = e / (3 * VEC), k = e % (3 * VEC);
      const int kk = k / VEC, ci = k % VEC;
      const int c = (blockIdx.x * cvb + lcx) * VEC + ci;
      if (c < a.C) a.partial[((long)blockIdx.y * 9 + k3 * 3 + kk) * a.C + c] = tot;
    }
    __syncthreads();
  }
}

// ---- fused backward (stride 1):  one pass over (dy, x) produces
//   g'[p]      = relu_mask(x[p]) * sum_k dy[p - dk] * w[k]          (data gradient wrt act(x))
//   dW[k]     += dy[p - dk] * act(x[p])                              (the SAME shifted dy values)
//   (sum g', sum g'*x)                                               (BatchNorm-backward sums)
// replacing three kernels (dgrad, wgrad, bn_bwd_reduce) that each re-read dy / x.
struct DwBwdArgs {
  const void* dy; const void* x; void* g;
  const float* w;  // [9][C]
  const float* pro_scale; const float* pro_shift;
  float* partial_w;   // [gridDim.y][9][C]
  float* partial_bn;  // [gridDim.y][2][C] or null
  long lddy, ldx, ldg;
  int N, H, W, C, dil, pro_mode, cvb_log2, CV;
  long strips;
};

template <typename T>
__global__ __launch_bounds__(DW_THREADS) void dwconv_bwd_fused_kernel(const DwBwdArgs a) {
  constexpr int VEC = HVec<T>::N;  // 4 channels per thread (8-byte bf16 vectors): this kernel
                                   // carries 9 tap accumulators per channel
  extern __shared__ __attribute__((aligned(16))) float dw_smem[];
  const int tid = threadIdx.x;
  const int cvb = 1 << a.cvb_log2;
  const int cx = tid & (cvb - 1), sy = tid >> a.cvb_log2;
  const int spb = DW_THREADS >> a.cvb_log2;
  const int cv = blockIdx.x * cvb + cx;
  const bool cok = cv < a.CV;
  const int c0 = cok ? cv * VEC : 0;
  const T* __restrict__ DY = reinterpret_cast<const T*>(a.dy);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  T* __restrict__ G = reinterpret_cast<T*>(a.g);
  const int WQ = (a.W + DW_TW - 1) / DW_TW;
  const float inv_wq = 1.0f / (float)WQ, inv_h = 1.0f / (float)a.H;
  const int d = a.dil;

  float sc[VEC], sh[VEC];
  if (a.pro_mode & PRO_AFFINE) {
    load_params<VEC>(a.pro_scale, c0, sc);
    load_params<VEC>(a.pro_shift, c0, sh);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
  }
  float accw[9][VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) accw[k][i] = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;

  const int nstrips = (int)a.strips;
  for (int s = blockIdx.y * spb + sy; s < nstrips; s += gridDim.y * spb) {
    if (!cok) continue;
    const int t = fast_div(s, WQ, inv_wq);
    const int wq = s - t * WQ;
    const int n = fast_div(t, a.H, inv_h);
    const int h = t - n * a.H;
    const int w0 = wq * DW_TW;
    const long prow = ((long)n * a.H + h) * a.W;
    // centre pixels: raw x (for mask / BN sums) and activated x (for the weight gradient)
    float xr[DW_TW][VEC], xa[DW_TW][VEC], g[DW_TW][VEC];
#pragma unroll
    for (int j = 0; j < DW_TW; ++j) {
      const int wc = min(w0 + j, a.W - 1);
      HVec<T>::load(X + (prow + wc) * a.ldx + c0, xr[j]);
      const float keep = (w0 + j < a.W) ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        xa[j][i] = xr[j][i];
        g[j][i] = 0.f;
      }
      dw_act<VEC>(xa[j], a.pro_mode, sc, sh);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        xa[j][i] *= keep;
        xr[j][i] *= keep;
      }
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int r = h + (1 - kh) * d;
      if (r < 0 || r >= a.H) continue;
      const long rowbase = ((long)n * a.H + r) * a.W;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        float wv[VEC];
        load_params<VEC>(a.w + (kh * 3 + kw) * a.C, c0, wv);
#pragma unroll
        for (int j = 0; j < DW_TW; ++j) {
          const int c = w0 + j + (1 - kw) * d;
          if (w0 + j < a.W && c >= 0 && c < a.W) {
            float dyv[VEC];
            HVec<T>::load(DY + (rowbase + c) * a.lddy + c0, dyv);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              g[j][i] = fmaf(dyv[i], wv[i], g[j][i]);
              accw[kh * 3 + kw][i] = fmaf(dyv[i], xa[j][i], accw[kh * 3 + kw][i]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < DW_TW; ++j) {
      if (w0 + j < a.W) {
        if (a.pro_mode & PRO_RELU) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const bool on = xa[j][i] > 0.f && (!(a.pro_mode & PRO_CLAMP6) || xa[j][i] < 6.f);
            g[j][i] = on ? g[j][i] : 0.f;
          }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          s1[i] += g[j][i];
          s2[i] = fmaf(g[j][i], xr[j][i], s2[i]);
        }
        HVec<T>::store(G + (prow + w0 + j) * a.ldg + c0, g[j]);
      }
    }
  }
  // block reductions (LDS: [spb][cvb][3*VEC] reused): 9 taps in three rounds, then BN sums
#pragma unroll
  for (int k3 = 0; k3 < 4; ++k3) {
    if (k3 == 3 && a.partial_bn == nullptr) break;
    float* mine = dw_smem + ((long)sy * cvb + cx) * 3 * VEC;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        mine[kk * VEC + i] = k3 < 3 ? accw[(k3 < 3 ? k3 : 0) * 3 + kk][i]
                                    : (kk == 0 ? s1[i] : (kk == 1 ? s2[i] : 0.f));
    __syncthreads();
    const int ncol = (k3 < 3 ? 3 : 2) * VEC;
    for (int e = tid; e < cvb * 3 * VEC; e += DW_THREADS) {
      const int lcx = e / (3 * VEC), k = e % (3 * VEC);
      if (k >= ncol) continue;
      float tot = 0.f;
      for (int r = 0; r < spb; ++r) tot += dw_smem[(long)r * cvb * 3 * VEC + e];
      const int kk = k / VEC, ci = k % VEC;
      const int c = (blockIdx.x * cvb + lcx) * VEC + ci;
      if (c < a.C) {
        if (k3 < 3) a.partial_w[((long)blockIdx.y * 9 + k3 * 3 + kk) * a.C + c] = tot;
        else a.partial_bn[((long)blockIdx.y * 2 + kk) * a.C + c] = tot;
      }
    }
    __syncthreads();
  }
}

static int pick_cvb_log2(int CV) {
  // largest utilisation among 32/16/8 channel vectors per block; ties -> wider
  int best = 5;
  double bu = 0;
  for (int l = 5; l >= 3; --l) {
    const int b = 1 << l;
    const double u = (double)CV / (double)(((CV + b - 1) / b) * b);
    if (u > bu + 1e-9) { bu = u; best = l; }
  }
  return best;
}

}  // namespace seg

// Number of partial rows / persistent blocks per channel block for one depthwise launch.
// kind: 0 forward / data gradient, 1 fused backward, 2 weight gradient (the LDS-tiled kernels size
// them differently; the strip kernels — stride 2, dilation > 2 — share one geometry).
extern "C" int seg_dwconv_grid_y(int dtype, int C, int N, int Ho, int Wo, int stride, int dil,
                                 int kind) {
  using namespace seg;
  const int tiled_stride = kind == 0 ? stride : 1;  // kinds 1/2 describe a stride-1 layer's backward
  // r06: stride 1 / dilation 1 forward and fused backward on the register-sliding kernels
  if ((kind == 0 || kind == 1) && dw_slide_supported(stride, dil, C))
    return dw_slide_rows(C, N, Ho, Wo);
  if ((kind == 0 || stride == 1) && dw_tiled_supported(tiled_stride, dil))
    return dw_tiled_grid_y(dtype, C, N, Ho, Wo, kind);
  if (kind == 0 && stride == 2 && dil == 1)  // LDS-tiled stride-2 forward
    return dw_tiled_s2_grid_y(dtype, C, N, Ho, Wo);
  if (g_dw_row && (kind == 0 || kind == 1) && dw_row_supported(stride, dil) && C % 4 == 0)
    return dw_row_grid_y(dtype, C, N, Ho, Wo, dil);  // stride 1: Ho x Wo is the input size too
  const int vec = dtype == DT_BF16 ? 8 : 4;
  const int CV = C / vec;
  const int l = pick_cvb_log2(CV);
  const int spb = DW_THREADS >> l;
  const long strips = (long)N * Ho * ((Wo + DW_TW - 1) / DW_TW);
  const int gx = (CV + (1 << l) - 1) >> l;
  long gy = (strips + spb - 1) / spb;
  long cap = 1536 / gx;  // ~6 blocks per CU; bounds the number of partial rows
  if (cap < 1) cap = 1;
  if (gy > cap) gy = cap;
  return (int)gy;
}
extern "C" int seg_dwconv3x3(int dtype, int mode, const void* x, long ldx, int N, int Hi, int Wi,
                             int C, const float* w9c, int w_layout, int stride, int dil, int pro_mode,
                             const float* pro_scale, const float* pro_shift, void* y, long ldy,
                             int Ho, int Wo, float* stat_partial, int grid_y, void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "dwconv3x3: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldx % vec == 0 && ldy % vec == 0,
              "dwconv3x3: C/ldx/ldy must be multiples of %d", vec);
  SEG_REQUIRE(((pro_mode & PRO_AFFINE) == 0) || (pro_scale && pro_shift),
              "dwconv3x3: affine prologue without scale/shift");
  SEG_REQUIRE(mode == MODE_FWD || stat_partial == nullptr, "dwconv3x3: stats only in forward");
  SEG_REQUIRE(grid_y >= 1, "dwconv3x3: grid_y must be >= 1");
  if (mode == MODE_FWD && dw_slide_supported(stride, dil, C)) {
    SEG_REQUIRE(Ho == Hi && Wo == Wi, "dwconv3x3: stride 1 keeps the size");
    return launch_dw_slide_fwd(dtype, x, ldx, N, Hi, Wi, C, w9c, w_layout, pro_mode, pro_scale,
                               pro_shift, y, ldy, stat_partial, grid_y, (hipStream_t)stream);
  }
  if (mode == MODE_FWD && dw_tiled_supported(stride, dil)) {  // incl. stride-1 dgrad (flipped taps)
    SEG_REQUIRE(Ho == Hi && Wo == Wi, "dwconv3x3: stride 1 keeps the size");
    return launch_dw_tiled(dtype, x, ldx, N, Hi, Wi, C, w9c, w_layout, dil, pro_mode, pro_scale,
                           pro_shift, y, ldy, stat_partial, grid_y, (hipStream_t)stream);
  }
  if (mode == MODE_FWD && stride == 2 && dil == 1) {  // LDS-tiled stride-2 forward
    SEG_REQUIRE(Ho == (Hi + 1) / 2 && Wo == (Wi + 1) / 2, "dwconv3x3: stride-2 output size");
    SEG_REQUIRE((long)N * Hi * Wi < (1L << 31), "dwconv3x3: tensor exceeds 32-bit pixel offsets");
    return launch_dw_tiled_s2(dtype, x, ldx, N, Hi, Wi, C, w9c, w_layout, pro_mode, pro_scale,
                              pro_shift, y, ldy, stat_partial, grid_y, (hipStream_t)stream);
  }
  SEG_REQUIRE(w_layout == 0, "dwconv3x3: the strip kernels take tap-major [9][C] weights");
  if (g_dw_row && mode == MODE_FWD && dw_row_supported(stride, dil) && C % 4 == 0) {
    SEG_REQUIRE(Ho == Hi && Wo == Wi, "dwconv3x3: stride 1 keeps the size");
    return launch_dw_row_fwd(dtype, x, ldx, N, Hi, Wi, C, w9c, dil, pro_mode, pro_scale, pro_shift,
                             y, ldy, stat_partial, grid_y, (hipStream_t)stream);
  }
  DwArgs a;
  a.x = x; a.w = w9c; a.y = y; a.pro_scale = pro_scale; a.pro_shift = pro_shift;
  a.stat_partial = stat_partial; a.ldx = ldx; a.ldy = ldy;
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.C = C; a.Ho = Ho; a.Wo = Wo;
  a.stride = stride; a.pad = dil; a.dil = dil; a.pro_mode = pro_mode;
  const int kvec = vec;  // channels per thread
  a.CV = C / kvec; a.cvb_log2 = pick_cvb_log2(a.CV);
  a.strips = (long)N * Ho * ((Wo + DW_TW - 1) / DW_TW);
  const int gx = (a.CV + (1 << a.cvb_log2) - 1) >> a.cvb_log2;
  SEG_REQUIRE(grid_y >= 1, "dwconv3x3: grid_y must be >= 1");
  const dim3 grid(gx, grid_y);
  const size_t lds = stat_partial ? (size_t)DW_THREADS * 2 * kvec * sizeof(float) : 0;
  hipStream_t st = (hipStream_t)stream;
  SEG_REQUIRE(a.strips < (1L << 31), "dwconv3x3: too many strips");
  SEG_REQUIRE((long)N * Hi * Wi * ldx < (1L << 31), "dwconv3x3: tensor exceeds 32-bit offsets");
#define SEG_DW_LAUNCH(TT, MM) \
  hipLaunchKernelGGL((dwconv_kernel<TT, MM>), grid, dim3(DW_THREADS), lds, st, a)
  if (dtype == DT_BF16) {
    if (mode == MODE_FWD) SEG_DW_LAUNCH(bf16_t, MODE_FWD); else SEG_DW_LAUNCH(bf16_t, MODE_DGRAD);
  } else {
    if (mode == MODE_FWD) SEG_DW_LAUNCH(float, MODE_FWD); else SEG_DW_LAUNCH(float, MODE_DGRAD);
  }
#undef SEG_DW_LAUNCH
  return check_launch("dwconv3x3");
}

extern "C" int seg_dwconv3x3_wgrad(int dtype, const void* x, long ldx, int N, int Hi, int Wi,
                                   int C, const void* dy, long lddy, int Ho, int Wo, int stride,
                                   int dil, int pro_mode, const float* pro_scale,
                                   const float* pro_shift, float* partial, int grid_y,
                                   void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "dwconv3x3_wgrad: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldx % vec == 0 && lddy % vec == 0,
              "dwconv3x3_wgrad: C/ldx/lddy must be multiples of %d", vec);
  SEG_REQUIRE(grid_y >= 1, "dwconv3x3_wgrad: grid_y must be >= 1");
  if (dw_tiled_supported(stride, dil)) {
    SEG_REQUIRE(Ho == Hi && Wo == Wi, "dwconv3x3_wgrad: stride 1 keeps the size");
    return launch_dw_wgrad_tiled(dtype, x, ldx, N, Hi, Wi, C, dy, lddy, dil, pro_mode, pro_scale,
                                 pro_shift, partial, grid_y, (hipStream_t)stream);
  }
  DwWgradArgs a;
  a.x = x; a.dy = dy; a.partial = partial; a.pro_scale = pro_scale; a.pro_shift = pro_shift;
  a.ldx = ldx; a.lddy = lddy; a.N = N; a.Hi = Hi; a.Wi = Wi; a.C = C; a.Ho = Ho; a.Wo = Wo;
  a.stride = stride; a.pad = dil; a.dil = dil; a.pro_mode = pro_mode;
  a.CV = C / vec; a.cvb_log2 = pick_cvb_log2(a.CV);
  a.strips = (long)N * Ho * ((Wo + DW_TW - 1) / DW_TW);
  const int gx = (a.CV + (1 << a.cvb_log2) - 1) >> a.cvb_log2;
  SEG_REQUIRE(grid_y >= 1, "dwconv3x3_wgrad: grid_y must be >= 1");
  const dim3 grid(gx, grid_y);
  const size_t lds = (size_t)DW_THREADS * 3 * vec * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  SEG_REQUIRE(a.strips < (1L << 31), "dwconv3x3_wgrad: too many strips");
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((dwconv_wgrad_kernel<bf16_t>), grid, dim3(DW_THREADS), lds, st, a);
  else
    hipLaunchKernelGGL((dwconv_wgrad_kernel<float>), grid, dim3(DW_THREADS), lds, st, a);
  return check_launch("dwconv3x3_wgrad");
}

// Fused stride-1 backward: g = relu_mask(x) * dgrad(dy), partial_w [grid_y][9][C],
// partial_bn [grid_y][2][C] = (sum g, sum g*x_raw) (nullable).  x is the forward input (raw tensor
// + prologue), w9c the forward taps (not reversed).
// The same with a tensor `res` ([N,H,W,C], pitch ldr, element type of g) added to the masked data
// gradient in the store path: g = relu_mask(x) * dgrad + res.  LDS-tiled kernel only (stride 1,
// dilation 1) — the caller falls back to a separate add otherwise (seg_dwconv3x3_bwd_fused_add_ok).
extern "C" int seg_dwconv3x3_bwd_fused_add_ok(int dil) { return dil == 1 ? 1 : 0; }

extern "C" int seg_dwconv3x3_bwd_fused_add(int dtype, const void* dy, long lddy, const void* x,
                                           long ldx, int N, int H, int W, int C, const float* w9c,
                                           int w_layout, int pro_mode, const float* pro_scale,
                                           const float* pro_shift, const void* res, long ldr,
                                           void* g, long ldg, float* partial_w, float* partial_bn,
                                           int grid_y, void* stream) {
  using namespace seg;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "dwconv3x3_bwd_fused_add: bad dtype %d", dtype);
  const int tvec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(C % tvec == 0 && ldx % tvec == 0 && lddy % tvec == 0 && ldg % tvec == 0 &&
                  ldr % 4 == 0 && res != nullptr,
              "dwconv3x3_bwd_fused_add: C/ld must be multiples of %d, res non-null", tvec);
  SEG_REQUIRE(((pro_mode & PRO_AFFINE) == 0) || (pro_scale && pro_shift),
              "dwconv3x3_bwd_fused_add: affine prologue without scale/shift");
  SEG_REQUIRE(grid_y >= 1 && partial_w != nullptr, "dwconv3x3_bwd_fused_add: bad grid/partials");
  if (dw_slide_supported(1, 1, C))
    return launch_dw_slide_bwd(dtype, dy, lddy, x, ldx, N, H, W, C, w9c, w_layout, pro_mode,
                               pro_scale, pro_shift, g, ldg, partial_w, partial_bn, grid_y,
                               (hipStream_t)stream, res, ldr);
  return launch_dw_bwd_tiled(dtype, dy, lddy, x, ldx, N, H, W, C, w9c, w_layout, 1, pro_mode,
                             pro_scale, pro_shift, g, ldg, partial_w, partial_bn, grid_y,
                             (hipStream_t)stream, res, ldr);
}

extern "C" int seg_dwconv3x3_bwd_fused(int dtype, const void* dy, long lddy, const void* x,
                                       long ldx, int N, int H, int W, int C, const float* w9c,
                                       int w_layout, int dil, int pro_mode, const float* pro_scale,
                                       const float* pro_shift, void* g, long ldg,
                                       float* partial_w, float* partial_bn, int grid_y,
                                       void* stream) {
  using namespace seg;
  const int vec = 4;  // HVec: 4 channels per thread in both element types
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "dwconv3x3_bwd_fused: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldx % vec == 0 && lddy % vec == 0 && ldg % vec == 0,
              "dwconv3x3_bwd_fused: C/ld must be multiples of %d", vec);
  SEG_REQUIRE(((pro_mode & PRO_AFFINE) == 0) || (pro_scale && pro_shift),
              "dwconv3x3_bwd_fused: affine prologue without scale/shift");
  SEG_REQUIRE(grid_y >= 1 && partial_w != nullptr, "dwconv3x3_bwd_fused: bad grid/partials");
  if (dw_slide_supported(1, dil, C))
    return launch_dw_slide_bwd(dtype, dy, lddy, x, ldx, N, H, W, C, w9c, w_layout, pro_mode,
                               pro_scale, pro_shift, g, ldg, partial_w, partial_bn, grid_y,
                               (hipStream_t)stream);
  if (dw_tiled_supported(1, dil)) {
    const int tvec = dtype == DT_BF16 ? 8 : 4;
    SEG_REQUIRE(C % tvec == 0 && ldx % tvec == 0 && lddy % tvec == 0 && ldg % tvec == 0,
                "dwconv3x3_bwd_fused: C/ld must be multiples of %d", tvec);
    return launch_dw_bwd_tiled(dtype, dy, lddy, x, ldx, N, H, W, C, w9c, w_layout, dil, pro_mode,
                               pro_scale, pro_shift, g, ldg, partial_w, partial_bn, grid_y,
                               (hipStream_t)stream);
  }
  SEG_REQUIRE(w_layout == 0, "dwconv3x3_bwd_fused: the strip kernel takes tap-major weights");
  if (g_dw_row && dw_row_supported(1, dil))
    return launch_dw_row_bwd(dtype, dy, lddy, x, ldx, N, H, W, C, w9c, dil, pro_mode, pro_scale,
                             pro_shift, g, ldg, partial_w, partial_bn, grid_y, (hipStream_t)stream);
  DwBwdArgs a;
  a.dy = dy; a.x = x; a.g = g; a.w = w9c; a.pro_scale = pro_scale; a.pro_shift = pro_shift;
  a.partial_w = partial_w; a.partial_bn = partial_bn;
  a.lddy = lddy; a.ldx = ldx; a.ldg = ldg; a.N = N; a.H = H; a.W = W; a.C = C; a.dil = dil;
  a.pro_mode = pro_mode; a.CV = C / vec; a.cvb_log2 = pick_cvb_log2(a.CV);
  a.strips = (long)N * H * ((W + DW_TW - 1) / DW_TW);
  SEG_REQUIRE(a.strips < (1L << 31), "dwconv3x3_bwd_fused: too many strips");
  SEG_REQUIRE((long)N * H * W * lddy < (1L << 31), "dwconv3x3_bwd_fused: 32-bit offsets");
  const int gx = (a.CV + (1 << a.cvb_log2) - 1) >> a.cvb_log2;
  const dim3 grid(gx, grid_y);
  const size_t lds = (size_t)DW_THREADS * 3 * vec * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((dwconv_bwd_fused_kernel<bf16_t>), grid, dim3(DW_THREADS), lds, st, a);
  else
    hipLaunchKernelGGL((dwconv_bwd_fused_kernel<float>), grid, dim3(DW_THREADS), lds, st, a);
  return check_launch("dwconv3x3_bwd_fused");
}

// dW [C][9] (= torch [C,1,3,3]) from the weight-gradient partials [R][9][C]
extern "C" int seg_dwconv3x3_wgrad_finalize(const float* partial, int R, int C, float* dw_c9,
                                            void* stream) {
  using namespace seg;
  SEG_REQUIRE(R >= 1 && C >= 1, "dwconv3x3_wgrad_finalize: bad R/C");
  return launch_dw_wgrad_finalize(partial, R, C, dw_c9, (hipStream_t)stream);
}
