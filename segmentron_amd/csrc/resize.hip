// Bilinear resampling (F.interpolate mode='bilinear', align_corners True/False) on NHWC tensors,
// the NHWC(T) -> NCHW(fp32) logits upsample at the model boundary, and the NCHW fp32 image ->
// channel-padded NHWC(T) conversion at the input boundary.
// Reference call sites: segmentron/models/deeplabv3_plus.py:39,44,71, segmentron/modules/module.py:64,96,
// segmentron/models/segbase.py:83; semantics SURVEY.md Appendix B (same float32 source-index
// arithmetic as ATen's upsample_bilinear2d: scale = (in-1)/(out-1), src = scale*dst,
// i0 = floor(src), i1 = min(i0+1, in-1), lambda = src - i0).
// Backward is written as a GATHER over the forward's own tap computation (no atomics, bitwise
// deterministic): an input pixel scans the small window of outputs that can reference it and
// re-derives their (i0, i1, lambda).
#include "common.h"
#include "resize_taps.h"

namespace seg {

constexpr int RS_THREADS = 256;

struct ResizeArgs {
  const void* x; void* y;
  const float* scale; const float* shift; const float* chan_mul;
  long ldx, ldy;
  int N, Hi, Wi, Ho, Wo, C, CV;
  int mode, align;
  float sh, sw;
};

template <typename T>
__global__ __launch_bounds__(RS_THREADS) void bilinear_fwd_kernel(const ResizeArgs a) {
  constexpr int VEC = Vec<T>::N;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const long total = (long)a.N * a.Ho * a.Wo * a.CV;
  for (long i = (long)blockIdx.x * RS_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * RS_THREADS) {
    const int cv = (int)(i % a.CV);
    long p = i / a.CV;
    const int wo = (int)(p % a.Wo); p /= a.Wo;
    const int ho = (int)(p % a.Ho);
    const int n = (int)(p / a.Ho);
    const int c0 = cv * VEC;
    int h0, h1, w0, w1; float lh, lw;
    taps(a.sh, ho, a.Hi, a.align, h0, h1, lh);
    taps(a.sw, wo, a.Wi, a.align, w0, w1, lw);
    const long base = (long)n * a.Hi * a.Wi;
    float f00[VEC], f01[VEC], f10[VEC], f11[VEC];
    Vec<T>::unpack(ldg16(X + (base + (long)h0 * a.Wi + w0) * a.ldx + c0), f00);
    Vec<T>::unpack(ldg16(X + (base + (long)h0 * a.Wi + w1) * a.ldx + c0), f01);
    Vec<T>::unpack(ldg16(X + (base + (long)h1 * a.Wi + w0) * a.ldx + c0), f10);
    Vec<T>::unpack(ldg16(X + (base + (long)h1 * a.Wi + w1) * a.ldx + c0), f11);
    apply_prologue<VEC>(f00, a.mode, a.scale, a.shift, c0);
    apply_prologue<VEC>(f01, a.mode, a.scale, a.shift, c0);
    apply_prologue<VEC>(f10, a.mode, a.scale, a.shift, c0);
    apply_prologue<VEC>(f11, a.mode, a.scale, a.shift, c0);
    float o[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float top = (1.f - lw) * f00[k] + lw * f01[k];
      const float bot = (1.f - lw) * f10[k] + lw * f11[k];
      o[k] = (1.f - lh) * top + lh * bot;
    }
    if (a.chan_mul) {
      const float* m = a.chan_mul + (long)n * a.C + c0;
#pragma unroll
      for (int k = 0; k < VEC; ++k) o[k] *= m[k];
    }
    stg16(Y + (((long)n * a.Ho + ho) * a.Wo + wo) * a.ldy + c0, Vec<T>::pack(o));
  }
}

// gx[n,hi,wi,:] = sum over outputs of weight * gy  (x side: Hi x Wi, y side: Ho x Wo)
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void bilinear_bwd_kernel(const ResizeArgs a) {
  constexpr int VEC = Vec<T>::N;
  const T* __restrict__ GY = reinterpret_cast<const T*>(a.y);
  T* __restrict__ GX = reinterpret_cast<T*>(const_cast<void*>(a.x));
  const long total = (long)a.N * a.Hi * a.Wi * a.CV;
  for (long i = (long)blockIdx.x * RS_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * RS_THREADS) {
    const int cv = (int)(i % a.CV);
    long p = i / a.CV;
    const int wi = (int)(p % a.Wi); p /= a.Wi;
    const int hi = (int)(p % a.Hi);
    const int n = (int)(p / a.Hi);
    const int c0 = cv * VEC;
    int hlo, hhi, wlo, whi;
    cand_range(a.sh, hi, a.Ho, a.align, hlo, hhi);
    cand_range(a.sw, wi, a.Wo, a.align, wlo, whi);
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int ho = hlo; ho <= hhi; ++ho) {
      const float wh = tap_weight(a.sh, ho, a.Hi, a.align, hi);
      if (wh == 0.f) continue;
      for (int wo = wlo; wo <= whi; ++wo) {
        const float ww = tap_weight(a.sw, wo, a.Wi, a.align, wi);
        if (ww == 0.f) continue;
        float g[VEC];
        Vec<T>::unpack(ldg16(GY + (((long)n * a.Ho + ho) * a.Wo + wo) * a.ldy + c0), g);
        const float w = wh * ww;
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(w, g[k], acc[k]);
      }
    }
    stg16(GX + (((long)n * a.Hi + hi) * a.Wi + wi) * a.ldx + c0, Vec<T>::pack(acc));
  }
}

// Same, for LARGE magnifications (PSP pyramid pooling: 1x1 .. 6x6 bins -> 129x257, every source
// pixel collects from up to Ho*Wo / (Hi*Wi) outputs): with one thread per source vector the 1x1
// bin ran 128 threads over 33 k outputs each — 17 ms, 30 % of the PSPNet train step.  Here a
// BLOCK owns (source pixel, 8 channel vectors): 32 pixel lanes stride over the footprint, the
// partial sums meet in LDS in a fixed order (deterministic).
constexpr int RSW_CVB = 8, RSW_PL = RS_THREADS / RSW_CVB;
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void bilinear_bwd_wide_kernel(const ResizeArgs a) {
  constexpr int VEC = Vec<T>::N;
  __shared__ float red[RSW_PL][RSW_CVB][VEC];
  const T* __restrict__ GY = reinterpret_cast<const T*>(a.y);
  T* __restrict__ GX = reinterpret_cast<T*>(const_cast<void*>(a.x));
  const int cvb = (a.CV + RSW_CVB - 1) / RSW_CVB;
  const int cx = threadIdx.x & (RSW_CVB - 1), pl = threadIdx.x / RSW_CVB;
  long p = blockIdx.x / cvb;
  const int cv = (int)(blockIdx.x - p * cvb) * RSW_CVB + cx;
  const int wi = (int)(p % a.Wi); p /= a.Wi;
  const int hi = (int)(p % a.Hi);
  const int n = (int)(p / a.Hi);
  const bool cok = cv < a.CV;
  const int c0 = (cok ? cv : 0) * VEC;
  int hlo, hhi, wlo, whi;
  cand_range(a.sh, hi, a.Ho, a.align, hlo, hhi);
  cand_range(a.sw, wi, a.Wo, a.align, wlo, whi);
  const int nw = whi - wlo + 1, nf = (hhi - hlo + 1) * nw;
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  for (int f = pl; f < nf; f += RSW_PL) {
    const int r = f / nw;
    const int ho = hlo + r, wo = wlo + (f - r * nw);
    const float w = tap_weight(a.sh, ho, a.Hi, a.align, hi) * tap_weight(a.sw, wo, a.Wi, a.align, wi);
    if (w == 0.f || !cok) continue;
    float g[VEC];
    Vec<T>::unpack(ldg16(GY + (((long)n * a.Ho + ho) * a.Wo + wo) * a.ldy + c0), g);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(w, g[k], acc[k]);
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) red[pl][cx][k] = acc[k];
  __syncthreads();
  if (pl == 0 && cok) {
    float tot[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) tot[k] = 0.f;
    for (int q = 0; q < RSW_PL; ++q)
#pragma unroll
      for (int k = 0; k < VEC; ++k) tot[k] += red[q][cx][k];
    stg16(GX + (((long)n * a.Hi + hi) * a.Wi + wi) * a.ldx + c0, Vec<T>::pack(tot));
  }
}

// ---- logits: NHWC (T, C valid channels, row pitch ldx) -> NCHW fp32 at (Ho, Wo)
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void upsample_to_nchw_kernel(const ResizeArgs a,
                                                                      float* __restrict__ out) {
  constexpr int VEC = Vec<T>::N;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const long total = (long)a.N * a.Ho * a.Wo;
  const long plane = (long)a.Ho * a.Wo;
  for (long i = (long)blockIdx.x * RS_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * RS_THREADS) {
    long p = i;
    const int wo = (int)(p % a.Wo); p /= a.Wo;
    const int ho = (int)(p % a.Ho);
    const int n = (int)(p / a.Ho);
    int h0, h1, w0, w1; float lh, lw;
    taps(a.sh, ho, a.Hi, a.align, h0, h1, lh);
    taps(a.sw, wo, a.Wi, a.align, w0, w1, lw);
    const long base = (long)n * a.Hi * a.Wi;
    float* o = out + (long)n * a.C * plane + (long)ho * a.Wo + wo;
    for (int cv = 0; cv < a.CV; ++cv) {
      const int c0 = cv * VEC;
      float f00[VEC], f01[VEC], f10[VEC], f11[VEC];
      Vec<T>::unpack(ldg16(X + (base + (long)h0 * a.Wi + w0) * a.ldx + c0), f00);
      Vec<T>::unpack(ldg16(X + (base + (long)h0 * a.Wi + w1) * a.ldx + c0), f01);
      Vec<T>::unpack(ldg16(X + (base + (long)h1 * a.Wi + w0) * a.ldx + c0), f10);
      Vec<T>::unpack(ldg16(X + (base + (long)h1 * a.Wi + w1) * a.ldx + c0), f11);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if (c0 + k < a.C) {
          const float top = (1.f - lw) * f00[k] + lw * f01[k];
          const float bot = (1.f - lw) * f10[k] + lw * f11[k];
          o[(long)(c0 + k) * plane] = (1.f - lh) * top + lh * bot;
        }
      }
    }
  }
}

// ... four consecutive output columns per thread -> 16-byte plane stores (r05: the inference
// configs write 0.15 - 2.5 GB of float32 logits through this kernel; dword stores reached
// 3.2 TB/s).  Same per-output arithmetic as above, bit for bit.  Wo % 4 == 0, out 16-byte aligned.
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void upsample_to_nchw_v4_kernel(const ResizeArgs a,
                                                                         float* __restrict__ out) {
  constexpr int VEC = Vec<T>::N;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const int wq = a.Wo >> 2;
  const long total = (long)a.N * a.Ho * wq;
  const long plane = (long)a.Ho * a.Wo;
  for (long i = (long)blockIdx.x * RS_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * RS_THREADS) {
    long p = i;
    const int wo0 = (int)(p % wq) * 4; p /= wq;
    const int ho = (int)(p % a.Ho);
    const int n = (int)(p / a.Ho);
    int h0, h1; float lh;
    taps(a.sh, ho, a.Hi, a.align, h0, h1, lh);
    int w0[4], w1[4]; float lw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) taps(a.sw, wo0 + j, a.Wi, a.align, w0[j], w1[j], lw[j]);
    const long base = (long)n * a.Hi * a.Wi;
    const T* r0 = X + (base + (long)h0 * a.Wi) * a.ldx;
    const T* r1 = X + (base + (long)h1 * a.Wi) * a.ldx;
    float* o = out + (long)n * a.C * plane + (long)ho * a.Wo + wo0;
    for (int cv = 0; cv < a.CV; ++cv) {
      const int c0 = cv * VEC;
      float res[4][VEC];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float f00[VEC], f01[VEC], f10[VEC], f11[VEC];
        Vec<T>::unpack(ldg16(r0 + (long)w0[j] * a.ldx + c0), f00);
        Vec<T>::unpack(ldg16(r0 + (long)w1[j] * a.ldx + c0), f01);
        Vec<T>::unpack(ldg16(r1 + (long)w0[j] * a.ldx + c0), f10);
        Vec<T>::unpack(ldg16(r1 + (long)w1[j] * a.ldx + c0), f11);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float top = (1.f - lw[j]) * f00[k] + lw[j] * f01[k];
          const float bot = (1.f - lw[j]) * f10[k] + lw[j] * f11[k];
          res[j][k] = (1.f - lh) * top + lh * bot;
        }
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if (c0 + k < a.C)
          *reinterpret_cast<float4*>(o + (long)(c0 + k) * plane) =
              make_float4(res[0][k], res[1][k], res[2][k], res[3][k]);
      }
    }
  }
}

// gx NHWC (T, padded channels written as 0) <- gy NCHW fp32
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void upsample_to_nchw_bwd_kernel(
    const ResizeArgs a, const float* __restrict__ gy) {
  constexpr int VEC = Vec<T>::N;
  T* __restrict__ GX = reinterpret_cast<T*>(const_cast<void*>(a.x));
  const long total = (long)a.N * a.Hi * a.Wi;
  const long plane = (long)a.Ho * a.Wo;
  for (long i = (long)blockIdx.x * RS_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * RS_THREADS) {
    long p = i;
    const int wi = (int)(p % a.Wi); p /= a.Wi;
    const int hi = (int)(p % a.Hi);
    const int n = (int)(p / a.Hi);
    int hlo, hhi, wlo, whi;
    cand_range(a.sh, hi, a.Ho, a.align, hlo, hhi);
    cand_range(a.sw, wi, a.Wo, a.align, wlo, whi);
    for (int cv = 0; cv < a.CV; ++cv) {
      const int c0 = cv * VEC;
      float acc[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
      for (int ho = hlo; ho <= hhi; ++ho) {
        const float wh = tap_weight(a.sh, ho, a.Hi, a.align, hi);
        if (wh == 0.f) continue;
        for (int wo = wlo; wo <= whi; ++wo) {
          const float ww = tap_weight(a.sw, wo, a.Wi, a.align, wi);
          if (ww == 0.f) continue;
          const float w = wh * ww;
          const float* g = gy + (long)n * a.C * plane + (long)ho * a.Wo + wo;
#pragma unroll
          for (int k = 0; k < VEC; ++k)
            if (c0 + k < a.C) acc[k] = fmaf(w, g[(long)(c0 + k) * plane], acc[k]);
        }
      }
      stg16(GX + (((long)n * a.Hi + hi) * a.Wi + wi) * a.ldx + c0, Vec<T>::pack(acc));
    }
  }
}

// ---- image boundary: NCHW fp32 [N,Cin,H,W] -> NHWC T [N,H,W,VEC] (channels >= Cin zero)
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void nchw_to_nhwc_pad_kernel(const float* __restrict__ x,
                                                                      T* __restrict__ y, int N,
                                                                      int Cin, long HW) {
  constexpr int VEC = Vec<T>::N;
  const long total = (long)N * HW;
  for (long i = (long)blockIdx.x * RS_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * RS_THREADS) {
    const long n = i / HW, p = i - n * HW;
    float f[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) f[k] = (k < Cin) ? x[(n * Cin + k) * HW + p] : 0.f;
    stg16(y + i * VEC, Vec<T>::pack(f));
  }
}

// ---- HRNet cross-resolution fuse (segmentron/models/backbones/hrnet.py:167-229):
//   y[n,h,w,:] = post_relu?( act_x(x[n,h,w,:]) + act_r(r[n, h>>s, w>>s, :]) )
// i.e. nn.Upsample(scale_factor=2^s, mode='nearest') of a deferred (conv1x1+BN) tensor fused with
// the running sum; backward of the upsampled operand = f x f block sums.
struct AddUpArgs {
  const void* x; const void* r; void* y;
  const float* sx; const float* tx; const float* sr; const float* tr;
  long ldx, ldr, ldy;
  int N, H, W, C, CV, mode_x, mode_r, shift, post_relu;
};

__device__ __forceinline__ int rs_fast_div(int s, int d, float inv) {
  int q = (int)((float)s * inv);
  if (q * d > s) --q;
  if ((q + 1) * d <= s) ++q;
  return q;
}

template <typename T>
__global__ __launch_bounds__(RS_THREADS) void nearest_add_kernel(const AddUpArgs a) {
  constexpr int VEC = Vec<T>::N;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ R = reinterpret_cast<const T*>(a.r);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const int total = a.N * a.H * a.W * a.CV;
  const int Hr = a.H >> a.shift, Wr = a.W >> a.shift;
  const float inv_cv = 1.f / (float)a.CV, inv_w = 1.f / (float)a.W, inv_h = 1.f / (float)a.H;
  for (int i = blockIdx.x * RS_THREADS + threadIdx.x; i < total; i += gridDim.x * RS_THREADS) {
    const int p = rs_fast_div(i, a.CV, inv_cv);
    const int c0 = (i - p * a.CV) * VEC;
    const int t = rs_fast_div(p, a.W, inv_w);
    const int w = p - t * a.W;
    const int n = rs_fast_div(t, a.H, inv_h);
    const int h = t - n * a.H;
    float f[VEC], g[VEC];
    Vec<T>::unpack(ldg16(X + (long)p * a.ldx + c0), f);
    apply_prologue<VEC>(f, a.mode_x, a.sx, a.tx, c0);
    const long pr = ((long)n * Hr + (h >> a.shift)) * Wr + (w >> a.shift);
    Vec<T>::unpack(ldg16(R + pr * a.ldr + c0), g);
    apply_prologue<VEC>(g, a.mode_r, a.sr, a.tr, c0);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      f[k] += g[k];
      if (a.post_relu) f[k] = fmaxf(f[k], 0.f);
    }
    stg16(Y + (long)p * a.ldy + c0, Vec<T>::pack(f));
  }
}

// gr[n,hr,wr,:] = sum_{dh,dw < 2^s} g[n, (hr<<s)+dh, (wr<<s)+dw, :]      (H, W: size of g)
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void nearest_sum_bwd_kernel(const AddUpArgs a) {
  constexpr int VEC = Vec<T>::N;
  const T* __restrict__ G = reinterpret_cast<const T*>(a.x);
  T* __restrict__ GR = reinterpret_cast<T*>(a.y);
  const int Hr = a.H >> a.shift, Wr = a.W >> a.shift, f = 1 << a.shift;
  const int total = a.N * Hr * Wr * a.CV;
  const float inv_cv = 1.f / (float)a.CV, inv_w = 1.f / (float)Wr, inv_h = 1.f / (float)Hr;
  for (int i = blockIdx.x * RS_THREADS + threadIdx.x; i < total; i += gridDim.x * RS_THREADS) {
    const int p = rs_fast_div(i, a.CV, inv_cv);
    const int c0 = (i - p * a.CV) * VEC;
    const int t = rs_fast_div(p, Wr, inv_w);
    const int wr = p - t * Wr;
    const int n = rs_fast_div(t, Hr, inv_h);
    const int hr = t - n * Hr;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int dh = 0; dh < f; ++dh)
      for (int dw = 0; dw < f; ++dw) {
        float g[VEC];
        Vec<T>::unpack(ldg16(G + (((long)n * a.H + (hr << a.shift) + dh) * a.W + (wr << a.shift) + dw) * a.ldx + c0), g);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += g[k];
      }
    stg16(GR + (long)p * a.ldy + c0, Vec<T>::pack(acc));
  }
}

static int rs_grid(long total) {
  long g = (total + RS_THREADS - 1) / RS_THREADS;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

static int fill_args(ResizeArgs& a, int dtype, const void* x, long ldx, int N, int Hi, int Wi,
                     int C, void* y, long ldy, int Ho, int Wo, int align, bool need_vec_c) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "resize: bad dtype %d", dtype);
  SEG_REQUIRE(ldx % vec == 0 && (!need_vec_c || (C % vec == 0 && ldy % vec == 0)),
              "resize: C/ld must be multiples of %d", vec);
  SEG_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "resize: empty");
  a.x = x; a.y = y; a.scale = nullptr; a.shift = nullptr; a.chan_mul = nullptr;
  a.ldx = ldx; a.ldy = ldy; a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.C = C;
  a.CV = (C + vec - 1) / vec; a.mode = PRO_NONE; a.align = align;
  a.sh = host_scale(Hi, Ho, align); a.sw = host_scale(Wi, Wo, align);
  return 0;
}

}  // namespace seg

extern "C" int seg_bilinear_fwd(int dtype, const void* x, long ldx, int N, int Hi, int Wi, int C,
                                int pro_mode, const float* pro_scale, const float* pro_shift,
                                const float* chan_mul, void* y, long ldy, int Ho, int Wo,
                                int align_corners, void* stream) {
  using namespace seg;
  ResizeArgs a;
  if (fill_args(a, dtype, x, ldx, N, Hi, Wi, C, y, ldy, Ho, Wo, align_corners, true)) return 1;
  SEG_REQUIRE(((pro_mode & PRO_AFFINE) == 0) || (pro_scale && pro_shift),
              "bilinear_fwd: missing scale/shift");
  a.mode = pro_mode; a.scale = pro_scale; a.shift = pro_shift; a.chan_mul = chan_mul;
  const int grid = rs_grid((long)N * Ho * Wo * a.CV);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((bilinear_fwd_kernel<bf16_t>), dim3(grid), dim3(RS_THREADS), 0,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((bilinear_fwd_kernel<float>), dim3(grid), dim3(RS_THREADS), 0,
                       (hipStream_t)stream, a);
  return check_launch("bilinear_fwd");
}

// gx [N,Hi,Wi,C] (written) <- gy [N,Ho,Wo,C]; (Hi,Wi) is the forward INPUT size
extern "C" int seg_bilinear_bwd(int dtype, void* gx, long ldgx, int N, int Hi, int Wi, int C,
                                const void* gy, long ldgy, int Ho, int Wo, int align_corners,
                                void* stream) {
  using namespace seg;
  ResizeArgs a;
  if (fill_args(a, dtype, gx, ldgx, N, Hi, Wi, C, const_cast<void*>(gy), ldgy, Ho, Wo,
                align_corners, true))
    return 1;
  // large magnification (>= 64 outputs per source pixel): one block per source pixel
  if ((long)Ho * Wo >= 64L * Hi * Wi) {
    const long nblk = (long)N * Hi * Wi * ((a.CV + RSW_CVB - 1) / RSW_CVB);
    SEG_REQUIRE(nblk < (1L << 31), "bilinear_bwd: too many blocks");
    if (dtype == DT_BF16)
      hipLaunchKernelGGL((bilinear_bwd_wide_kernel<bf16_t>), dim3((unsigned)nblk),
                         dim3(RS_THREADS), 0, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((bilinear_bwd_wide_kernel<float>), dim3((unsigned)nblk), dim3(RS_THREADS),
                         0, (hipStream_t)stream, a);
    return check_launch("bilinear_bwd (wide)");
  }
  const int grid = rs_grid((long)N * Hi * Wi * a.CV);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((bilinear_bwd_kernel<bf16_t>), dim3(grid), dim3(RS_THREADS), 0,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((bilinear_bwd_kernel<float>), dim3(grid), dim3(RS_THREADS), 0,
                       (hipStream_t)stream, a);
  return check_launch("bilinear_bwd");
}

// x: NHWC T with C valid channels and row pitch ldx >= roundup(C, VEC); out: NCHW fp32
extern "C" int seg_upsample_to_nchw(int dtype, const void* x, long ldx, int N, int Hi, int Wi,
                                    int C, float* out, int Ho, int Wo, int align_corners,
                                    void* stream) {
  using namespace seg;
  ResizeArgs a;
  if (fill_args(a, dtype, x, ldx, N, Hi, Wi, C, nullptr, 0, Ho, Wo, align_corners, false)) return 1;
  SEG_REQUIRE(ldx >= (long)a.CV * (dtype == DT_BF16 ? 8 : 4), "upsample_to_nchw: ldx too small");
  if (Wo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int grid4 = rs_grid((long)N * Ho * (Wo / 4));
    if (dtype == DT_BF16)
      hipLaunchKernelGGL((upsample_to_nchw_v4_kernel<bf16_t>), dim3(grid4), dim3(RS_THREADS), 0,
                         (hipStream_t)stream, a, out);
    else
      hipLaunchKernelGGL((upsample_to_nchw_v4_kernel<float>), dim3(grid4), dim3(RS_THREADS), 0,
                         (hipStream_t)stream, a, out);
    return check_launch("upsample_to_nchw");
  }
  const int grid = rs_grid((long)N * Ho * Wo);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((upsample_to_nchw_kernel<bf16_t>), dim3(grid), dim3(RS_THREADS), 0,
                       (hipStream_t)stream, a, out);
  else
    hipLaunchKernelGGL((upsample_to_nchw_kernel<float>), dim3(grid), dim3(RS_THREADS), 0,
                       (hipStream_t)stream, a, out);
  return check_launch("upsample_to_nchw");
}

extern "C" int seg_upsample_to_nchw_bwd(int dtype, void* gx, long ldgx, int N, int Hi, int Wi,
                                        int C, const float* gy, int Ho, int Wo, int align_corners,
                                        void* stream) {
  using namespace seg;
  ResizeArgs a;
  if (fill_args(a, dtype, gx, ldgx, N, Hi, Wi, C, nullptr, 0, Ho, Wo, align_corners, false))
    return 1;
  SEG_REQUIRE(ldgx >= (long)a.CV * (dtype == DT_BF16 ? 8 : 4), "upsample_to_nchw_bwd: ld small");
  a.CV = (int)(ldgx / (dtype == DT_BF16 ? 8 : 4));  // zero-fill every padded channel of gx
  const int grid = rs_grid((long)N * Hi * Wi);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((upsample_to_nchw_bwd_kernel<bf16_t>), dim3(grid), dim3(RS_THREADS), 0,
                       (hipStream_t)stream, a, gy);
  else
    hipLaunchKernelGGL((upsample_to_nchw_bwd_kernel<float>), dim3(grid), dim3(RS_THREADS), 0,
                       (hipStream_t)stream, a, gy);
  return check_launch("upsample_to_nchw_bwd");
}

extern "C" int seg_nchw_to_nhwc_pad(int dtype, const float* x, int N, int Cin, int H, int W,
                                    void* y, void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "nchw_to_nhwc_pad: bad dtype %d", dtype);
  SEG_REQUIRE(Cin >= 1 && Cin <= vec, "nchw_to_nhwc_pad: Cin=%d must be <= %d", Cin, vec);
  const long HW = (long)H * W;
  const int grid = rs_grid((long)N * HW);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<bf16_t>), dim3(grid), dim3(RS_THREADS), 0,
                       (hipStream_t)stream, x, reinterpret_cast<bf16_t*>(y), N, Cin, HW);
  else
    hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<float>), dim3(grid), dim3(RS_THREADS), 0,
                       (hipStream_t)stream, x, reinterpret_cast<float*>(y), N, Cin, HW);
  return check_launch("nchw_to_nhwc_pad");
}

// y = post_relu?( act_x(x) + act_r(nearest_up_{2^shift}(r)) ); x,y: [N,H,W,C], r: [N,H>>shift,W>>shift,C]
extern "C" int seg_nearest_add(int dtype, const void* x, long ldx, int mode_x, const float* sx,
                               const float* tx, const void* r, long ldr, int mode_r,
                               const float* sr, const float* tr, int shift, int post_relu, void* y,
                               long ldy, int N, int H, int W, int C, void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "nearest_add: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldx % vec == 0 && ldr % vec == 0 && ldy % vec == 0,
              "nearest_add: C/ld must be multiples of %d", vec);
  SEG_REQUIRE(shift >= 0 && shift < 8 && (H % (1 << shift)) == 0 && (W % (1 << shift)) == 0,
              "nearest_add: H=%d W=%d not divisible by 2^%d", H, W, shift);
  SEG_REQUIRE((long)N * H * W * (C / vec) < (1L << 31), "nearest_add: too large");
  AddUpArgs a;
  a.x = x; a.r = r; a.y = y; a.sx = sx; a.tx = tx; a.sr = sr; a.tr = tr;
  a.ldx = ldx; a.ldr = ldr; a.ldy = ldy; a.N = N; a.H = H; a.W = W; a.C = C; a.CV = C / vec;
  a.mode_x = mode_x; a.mode_r = mode_r; a.shift = shift; a.post_relu = post_relu;
  const int grid = rs_grid((long)N * H * W * a.CV);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((nearest_add_kernel<bf16_t>), dim3(grid), dim3(RS_THREADS), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((nearest_add_kernel<float>), dim3(grid), dim3(RS_THREADS), 0, (hipStream_t)stream, a);
  return check_launch("nearest_add");
}

// gr [N,H>>shift,W>>shift,C] = 2^shift x 2^shift block sums of g [N,H,W,C]
extern "C" int seg_nearest_sum_bwd(int dtype, const void* g, long ldg, int N, int H, int W, int C,
                                   int shift, void* gr, long ldgr, void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "nearest_sum_bwd: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldg % vec == 0 && ldgr % vec == 0, "nearest_sum_bwd: C/ld");
  SEG_REQUIRE(shift >= 0 && shift < 8 && (H % (1 << shift)) == 0 && (W % (1 << shift)) == 0,
              "nearest_sum_bwd: size not divisible");
  AddUpArgs a;
  a.x = g; a.r = nullptr; a.y = gr; a.sx = a.tx = a.sr = a.tr = nullptr;
  a.ldx = ldg; a.ldr = 0; a.ldy = ldgr; a.N = N; a.H = H; a.W = W; a.C = C; a.CV = C / vec;
  a.mode_x = 0; a.mode_r = 0; a.shift = shift; a.post_relu = 0;
  const int grid = rs_grid((long)N * (H >> shift) * (W >> shift) * a.CV);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((nearest_sum_bwd_kernel<bf16_t>), dim3(grid), dim3(RS_THREADS), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((nearest_sum_bwd_kernel<float>), dim3(grid), dim3(RS_THREADS), 0, (hipStream_t)stream, a);
  return check_launch("nearest_sum_bwd");
}
