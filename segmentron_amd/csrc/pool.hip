// Pooling on NHWC tensors.
//   nn.MaxPool2d(3, 2, 1)          segmentron/models/backbones/resnet.py:119   (fwd + bwd)
//   nn.AdaptiveAvgPool2d(o)        segmentron/modules/module.py:52,89          (fwd + bwd)
// Max-pool consumes a deferred activation (the stem's BN+ReLU rides in its load path), pads with
// -inf and records the winning tap (first maximum in (kh,kw) scan order, like ATen) as one byte per
// output element; backward is a gather over those indices (deterministic, no atomics).
// Adaptive average pooling uses ATen's bins [floor(i*H/o), ceil((i+1)*H/o)) (overlapping when
// H % o != 0): forward emits fp32 partial sums per (bin, pixel chunk) that seg_colsum reduces;
// backward gathers from the at most 2x2 bins that contain a pixel.
#include "common.h"
#include <float.h>

namespace seg {

constexpr int PL_THREADS = 256;

__device__ __forceinline__ int pl_fast_div(int s, int d, float inv) {
  int q = (int)((float)s * inv);
  if (q * d > s) --q;
  if ((q + 1) * d <= s) ++q;
  return q;
}

struct MaxPoolArgs {
  const void* x; void* y; unsigned char* idx;
  const float* scale; const float* shift;
  long ldx, ldy;
  int N, Hi, Wi, Ho, Wo, C, CV;
  int k, stride, pad, mode;
};

template <typename T>
__global__ __launch_bounds__(PL_THREADS) void maxpool_fwd_kernel(const MaxPoolArgs a) {
  constexpr int VEC = Vec<T>::N;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const int total = a.N * a.Ho * a.Wo * a.CV;
  const float inv_cv = 1.f / (float)a.CV, inv_wo = 1.f / (float)a.Wo, inv_ho = 1.f / (float)a.Ho;
  for (int i = blockIdx.x * PL_THREADS + threadIdx.x; i < total; i += gridDim.x * PL_THREADS) {
    int p = pl_fast_div(i, a.CV, inv_cv);
    const int cv = i - p * a.CV;
    int t = pl_fast_div(p, a.Wo, inv_wo);
    const int wo = p - t * a.Wo;
    const int n = pl_fast_div(t, a.Ho, inv_ho);
    const int ho = t - n * a.Ho;
    const int c0 = cv * VEC;
    float best[VEC];
    int bi[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { best[e] = -FLT_MAX; bi[e] = 0; }
    bool first = true;
    for (int kh = 0; kh < a.k; ++kh) {
      const int hi = ho * a.stride - a.pad + kh;
      if (hi < 0 || hi >= a.Hi) continue;
      for (int kw = 0; kw < a.k; ++kw) {
        const int wi = wo * a.stride - a.pad + kw;
        if (wi < 0 || wi >= a.Wi) continue;
        float f[VEC];
        Vec<T>::unpack(ldg16(X + (((long)n * a.Hi + hi) * a.Wi + wi) * a.ldx + c0), f);
        apply_prologue<VEC>(f, a.mode, a.scale, a.shift, c0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if (first || f[e] > best[e]) { best[e] = f[e]; bi[e] = kh * a.k + kw; }
        }
        first = false;
      }
    }
    const long o = (((long)n * a.Ho + ho) * a.Wo + wo);
    stg16(Y + o * a.ldy + c0, Vec<T>::pack(best));
#pragma unroll
    for (int e = 0; e < VEC; ++e) a.idx[o * a.C + c0 + e] = (unsigned char)bi[e];
  }
}

// gx[n,h,w,c] = sum over windows containing (h,w) whose recorded winner is this pixel
template <typename T>
__global__ __launch_bounds__(PL_THREADS) void maxpool_bwd_kernel(const MaxPoolArgs a) {
  constexpr int VEC = Vec<T>::N;
  const T* __restrict__ GY = reinterpret_cast<const T*>(a.y);
  T* __restrict__ GX = reinterpret_cast<T*>(const_cast<void*>(a.x));
  const int total = a.N * a.Hi * a.Wi * a.CV;
  const float inv_cv = 1.f / (float)a.CV, inv_wi = 1.f / (float)a.Wi, inv_hi = 1.f / (float)a.Hi;
  for (int i = blockIdx.x * PL_THREADS + threadIdx.x; i < total; i += gridDim.x * PL_THREADS) {
    int p = pl_fast_div(i, a.CV, inv_cv);
    const int cv = i - p * a.CV;
    int t = pl_fast_div(p, a.Wi, inv_wi);
    const int w = p - t * a.Wi;
    const int n = pl_fast_div(t, a.Hi, inv_hi);
    const int h = t - n * a.Hi;
    const int c0 = cv * VEC;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int kh = 0; kh < a.k; ++kh) {
      const int hn = h + a.pad - kh;
      if (hn < 0 || (hn % a.stride) != 0) continue;
      const int ho = hn / a.stride;
      if (ho >= a.Ho) continue;
      for (int kw = 0; kw < a.k; ++kw) {
        const int wn = w + a.pad - kw;
        if (wn < 0 || (wn % a.stride) != 0) continue;
        const int wo = wn / a.stride;
        if (wo >= a.Wo) continue;
        const long o = (((long)n * a.Ho + ho) * a.Wo + wo);
        float g[VEC];
        Vec<T>::unpack(ldg16(GY + o * a.ldy + c0), g);
        const int tap = kh * a.k + kw;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if ((int)a.idx[o * a.C + c0 + e] == tap) acc[e] += g[e];
      }
    }
    stg16(GX + (((long)n * a.Hi + h) * a.Wi + w) * a.ldx + c0, Vec<T>::pack(acc));
  }
}

// ------------------------------------------------------------------ adaptive average pooling
struct AvgPoolArgs {
  const void* x; void* gx; const void* gy;
  float* partial;  // [chunks][N*o*o][C]
  long ldx, ldg, ldgy;
  int N, H, W, C, CV, o, chunks, cvb_log2;
};

__device__ __forceinline__ int bin_start(int i, int H, int o) { return (i * H) / o; }
__device__ __forceinline__ int bin_end(int i, int H, int o) { return ((i + 1) * H + o - 1) / o; }

// grid: x = channel-vector blocks, y = chunk, z = n*o*o + bin
template <typename T>
__global__ __launch_bounds__(PL_THREADS) void adaptive_avgpool_partial_kernel(const AvgPoolArgs a) {
  constexpr int VEC = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) float pl_smem[];
  const int tid = threadIdx.x;
  const int cvb = 1 << a.cvb_log2;
  const int cx = tid & (cvb - 1), sy = tid >> a.cvb_log2;
  const int spb = PL_THREADS >> a.cvb_log2;
  const int cv = blockIdx.x * cvb + cx;
  const int bin = blockIdx.z;
  const int n = bin / (a.o * a.o), b = bin - n * a.o * a.o;
  const int bi = b / a.o, bj = b - bi * a.o;
  const int h0 = bin_start(bi, a.H, a.o), h1 = bin_end(bi, a.H, a.o);
  const int w0 = bin_start(bj, a.W, a.o), w1 = bin_end(bj, a.W, a.o);
  const int bw = w1 - w0, npix = (h1 - h0) * bw;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (cv < a.CV) {
    const float inv_bw = 1.f / (float)bw;
    for (int q = blockIdx.y * spb + sy; q < npix; q += a.chunks * spb) {
      const int r = pl_fast_div(q, bw, inv_bw);
      const int h = h0 + r, w = w0 + (q - r * bw);
      float f[VEC];
      Vec<T>::unpack(ldg16(X + (((long)n * a.H + h) * a.W + w) * a.ldx + cv * VEC), f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += f[e];
    }
  }
  float* mine = pl_smem + ((long)sy * cvb + cx) * VEC;
#pragma unroll
  for (int e = 0; e < VEC; ++e) mine[e] = acc[e];
  __syncthreads();
  for (int e = tid; e < cvb * VEC; e += PL_THREADS) {
    float tot = 0.f;
    for (int r = 0; r < spb; ++r) tot += pl_smem[(long)r * cvb * VEC + e];
    const int c = blockIdx.x * cvb * VEC + e;
    if (c < a.C) a.partial[((long)blockIdx.y * gridDim.z + bin) * a.C + c] = tot;
  }
}

// gx[n,h,w,:] = sum_{bins containing (h,w)} gy[n,bi,bj,:] / area(bi,bj)
template <typename T>
__global__ __launch_bounds__(PL_THREADS) void adaptive_avgpool_bwd_kernel(const AvgPoolArgs a) {
  constexpr int VEC = Vec<T>::N;
  const T* __restrict__ GY = reinterpret_cast<const T*>(a.gy);
  T* __restrict__ GX = reinterpret_cast<T*>(a.gx);
  const int total = a.N * a.H * a.W * a.CV;
  const float inv_cv = 1.f / (float)a.CV, inv_w = 1.f / (float)a.W, inv_h = 1.f / (float)a.H;
  for (int i = blockIdx.x * PL_THREADS + threadIdx.x; i < total; i += gridDim.x * PL_THREADS) {
    int p = pl_fast_div(i, a.CV, inv_cv);
    const int cv = i - p * a.CV;
    int t = pl_fast_div(p, a.W, inv_w);
    const int w = p - t * a.W;
    const int n = pl_fast_div(t, a.H, inv_h);
    const int h = t - n * a.H;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int bi = 0; bi < a.o; ++bi) {
      const int h0 = bin_start(bi, a.H, a.o), h1 = bin_end(bi, a.H, a.o);
      if (h < h0 || h >= h1) continue;
      for (int bj = 0; bj < a.o; ++bj) {
        const int w0 = bin_start(bj, a.W, a.o), w1 = bin_end(bj, a.W, a.o);
        if (w < w0 || w >= w1) continue;
        const float inv_area = 1.f / (float)((h1 - h0) * (w1 - w0));
        float g[VEC];
        Vec<T>::unpack(ldg16(GY + (((long)n * a.o + bi) * a.o + bj) * a.ldgy + cv * VEC), g);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = fmaf(g[e], inv_area, acc[e]);
      }
    }
    stg16(GX + (((long)n * a.H + h) * a.W + w) * a.ldg + cv * VEC, Vec<T>::pack(acc));
  }
}

static int pl_grid(long total) {
  long g = (total + PL_THREADS - 1) / PL_THREADS;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

static int pl_pick_cvb_log2(int CV) {
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

// y = maxpool_{k,stride,pad}(act(x)) with -inf padding; idx: one byte per output element (winning
// tap kh*k+kw).  mode/scale/shift: the producer's pending BatchNorm(+ReLU).
extern "C" int seg_maxpool_fwd(int dtype, const void* x, long ldx, int N, int Hi, int Wi, int C,
                               int k, int stride, int pad, int pro_mode, const float* pro_scale,
                               const float* pro_shift, void* y, long ldy, int Ho, int Wo,
                               void* idx, void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "maxpool_fwd: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldx % vec == 0 && ldy % vec == 0, "maxpool_fwd: C/ld multiples of %d", vec);
  SEG_REQUIRE(k >= 1 && k <= 15 && idx != nullptr, "maxpool_fwd: bad kernel size / idx");
  SEG_REQUIRE((long)N * Ho * Wo * (C / vec) < (1L << 31), "maxpool_fwd: too large");
  MaxPoolArgs a;
  a.x = x; a.y = y; a.idx = (unsigned char*)idx; a.scale = pro_scale; a.shift = pro_shift;
  a.ldx = ldx; a.ldy = ldy; a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.C = C;
  a.CV = C / vec; a.k = k; a.stride = stride; a.pad = pad; a.mode = pro_mode;
  const int grid = pl_grid((long)N * Ho * Wo * a.CV);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t>), dim3(grid), dim3(PL_THREADS), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(grid), dim3(PL_THREADS), 0, (hipStream_t)stream, a);
  return check_launch("maxpool_fwd");
}

// gx [N,Hi,Wi,C] (written) <- gy [N,Ho,Wo,C] through the recorded indices
extern "C" int seg_maxpool_bwd(int dtype, void* gx, long ldgx, int N, int Hi, int Wi, int C, int k,
                               int stride, int pad, const void* gy, long ldgy, int Ho, int Wo,
                               const void* idx, void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "maxpool_bwd: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldgx % vec == 0 && ldgy % vec == 0, "maxpool_bwd: C/ld multiples of %d", vec);
  SEG_REQUIRE((long)N * Hi * Wi * (C / vec) < (1L << 31), "maxpool_bwd: too large");
  MaxPoolArgs a;
  a.x = gx; a.y = const_cast<void*>(gy); a.idx = (unsigned char*)const_cast<void*>(idx);
  a.scale = nullptr; a.shift = nullptr; a.ldx = ldgx; a.ldy = ldgy;
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.C = C; a.CV = C / vec;
  a.k = k; a.stride = stride; a.pad = pad; a.mode = 0;
  const int grid = pl_grid((long)N * Hi * Wi * a.CV);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t>), dim3(grid), dim3(PL_THREADS), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(grid), dim3(PL_THREADS), 0, (hipStream_t)stream, a);
  return check_launch("maxpool_bwd");
}

extern "C" int seg_adaptive_avgpool_chunks(int H, int W, int o) {
  const long npix = ((long)H / o + 1) * ((long)W / o + 1);
  long c = (npix + 2047) / 2048;  // ~2048 pixels per chunk
  if (c > 64) c = 64;
  if (c < 1) c = 1;
  return (int)c;
}

// partial [chunks][N*o*o][C] fp32 bin sums; seg_colsum over chunks, then divide by the bin areas.
extern "C" int seg_adaptive_avgpool_partial(int dtype, const void* x, long ldx, int N, int H, int W,
                                            int C, int o, float* partial, int chunks,
                                            void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "adaptive_avgpool: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldx % vec == 0, "adaptive_avgpool: C/ld multiples of %d", vec);
  SEG_REQUIRE(o >= 1 && o <= H && o <= W && chunks >= 1, "adaptive_avgpool: bad bins/chunks");
  AvgPoolArgs a;
  a.x = x; a.gx = nullptr; a.gy = nullptr; a.partial = partial; a.ldx = ldx; a.ldg = 0; a.ldgy = 0;
  a.N = N; a.H = H; a.W = W; a.C = C; a.CV = C / vec; a.o = o; a.chunks = chunks;
  a.cvb_log2 = pl_pick_cvb_log2(a.CV);
  const int gx = (a.CV + (1 << a.cvb_log2) - 1) >> a.cvb_log2;
  const dim3 grid(gx, chunks, N * o * o);
  const size_t lds = (size_t)PL_THREADS * vec * sizeof(float);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((adaptive_avgpool_partial_kernel<bf16_t>), grid, dim3(PL_THREADS), lds, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((adaptive_avgpool_partial_kernel<float>), grid, dim3(PL_THREADS), lds, (hipStream_t)stream, a);
  return check_launch("adaptive_avgpool_partial");
}

extern "C" int seg_adaptive_avgpool_bwd(int dtype, void* gx, long ldgx, int N, int H, int W, int C,
                                        int o, const void* gy, long ldgy, void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "adaptive_avgpool_bwd: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldgx % vec == 0 && ldgy % vec == 0, "adaptive_avgpool_bwd: C/ld");
  SEG_REQUIRE((long)N * H * W * (C / vec) < (1L << 31), "adaptive_avgpool_bwd: too large");
  AvgPoolArgs a;
  a.x = nullptr; a.gx = gx; a.gy = gy; a.partial = nullptr; a.ldx = 0; a.ldg = ldgx; a.ldgy = ldgy;
  a.N = N; a.H = H; a.W = W; a.C = C; a.CV = C / vec; a.o = o; a.chunks = 1; a.cvb_log2 = 0;
  const int grid = pl_grid((long)N * H * W * a.CV);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((adaptive_avgpool_bwd_kernel<bf16_t>), dim3(grid), dim3(PL_THREADS), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((adaptive_avgpool_bwd_kernel<float>), dim3(grid), dim3(PL_THREADS), 0, (hipStream_t)stream, a);
  return check_launch("adaptive_avgpool_bwd");
}
