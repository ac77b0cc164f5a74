// pixAcc / mIoU counters of segmentron/utils/score.py:83-113 in ONE pass over the logits:
//   batch_pix_accuracy       : predict = argmax(logits.long()) + 1 (the reference really truncates
//                              the logits to integers first, score.py:86), labelled = target+1 > 0
//   batch_intersection_union : predict = (argmax(logits) + 1) * labelled, three torch.histc over
//                              [1, nclass] (values 0 fall outside and are dropped)
// The reference materialises argmax maps, moves three float maps to the CPU for histc and comes
// back (score.py:108-110).  Here every pixel updates block-local integer counters in LDS, one
// 64-bit atomic per counter and block goes to HBM (integers: the result does not depend on the
// order).  counters (int64, ACCUMULATED): [correct, labelled, inter[nclass], pred[nclass],
// lab[nclass]].  The fused variant takes the network's low-resolution NHWC logits and applies the
// final bilinear upsample on the fly with exactly the arithmetic of seg_upsample_to_nchw
// (resize.hip), so its arg-maxes are those of the materialised tensor.
#include "common.h"
#include "resize_taps.h"

namespace seg {

constexpr int MT_THREADS = 256;

struct MetricAcc {
  int best_f, best_l;
  float vmax;
  long lmax;
  __device__ __forceinline__ void init() { best_f = best_l = -1; vmax = 0.f; lmax = 0; }
  __device__ __forceinline__ void feed(int c, float v) {
    const long lv = (long)v;  // Tensor.long(): truncation toward zero
    if (best_f < 0 || v > vmax) { vmax = v; best_f = c; }     // first maximum wins (torch.argmax)
    if (best_l < 0 || lv > lmax) { lmax = lv; best_l = c; }
  }
};

__device__ __forceinline__ void metric_count(unsigned* cnt, int nclass, const MetricAcc& m,
                                             long t) {
  const long tl = t + 1;
  const bool labelled = tl > 0;
  if (labelled) {
    atomicAdd(&cnt[1], 1u);
    if ((long)m.best_l + 1 == tl) atomicAdd(&cnt[0], 1u);
    const long pred = (long)m.best_f + 1;  // in [1, C]
    if (pred >= 1 && pred <= nclass) atomicAdd(&cnt[2 + nclass + (int)pred - 1], 1u);
    if (pred == tl && pred >= 1 && pred <= nclass) atomicAdd(&cnt[2 + (int)pred - 1], 1u);
  }
  if (tl >= 1 && tl <= nclass) atomicAdd(&cnt[2 + 2 * nclass + (int)tl - 1], 1u);
}

__device__ __forceinline__ void metric_flush(unsigned* cnt, int ncnt, long* out) {
  __syncthreads();
  for (int i = threadIdx.x; i < ncnt; i += MT_THREADS)
    if (cnt[i]) atomicAdd(reinterpret_cast<unsigned long long*>(out) + i, (unsigned long long)cnt[i]);
}

__global__ __launch_bounds__(MT_THREADS) void metric_nchw_kernel(
    const float* __restrict__ x, const long* __restrict__ target, int N, int C, long plane,
    int nclass, long* __restrict__ out) {
  extern __shared__ unsigned mt_cnt[];
  const int ncnt = 2 + 3 * nclass;
  for (int i = threadIdx.x; i < ncnt; i += MT_THREADS) mt_cnt[i] = 0;
  __syncthreads();
  const long total = (long)N * plane;
  for (long i = (long)blockIdx.x * MT_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * MT_THREADS) {
    const long n = i / plane, p = i - n * plane;
    const float* px = x + n * C * plane + p;
    MetricAcc m;
    m.init();
    for (int c = 0; c < C; ++c) m.feed(c, px[(long)c * plane]);
    metric_count(mt_cnt, nclass, m, target[i]);
  }
  metric_flush(mt_cnt, ncnt, out);
}

template <typename T>
__global__ __launch_bounds__(MT_THREADS) void metric_upsample_kernel(
    const T* __restrict__ X, long ldx, int N, int Hi, int Wi, int C, const long* __restrict__ target,
    int Ho, int Wo, float sh, float sw, int align, int nclass, long* __restrict__ out) {
  constexpr int VEC = Vec<T>::N;
  extern __shared__ unsigned mt_cnt[];
  const int ncnt = 2 + 3 * nclass;
  for (int i = threadIdx.x; i < ncnt; i += MT_THREADS) mt_cnt[i] = 0;
  __syncthreads();
  const long total = (long)N * Ho * Wo;
  const int CV = (C + VEC - 1) / VEC;
  for (long i = (long)blockIdx.x * MT_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * MT_THREADS) {
    long p = i;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    int h0, h1, w0, w1; float lh, lw;
    taps(sh, ho, Hi, align, h0, h1, lh);
    taps(sw, wo, Wi, align, w0, w1, lw);
    const long base = (long)n * Hi * Wi;
    MetricAcc m;
    m.init();
    for (int cv = 0; cv < CV; ++cv) {
      const int c0 = cv * VEC;
      float f00[VEC], f01[VEC], f10[VEC], f11[VEC];
      Vec<T>::unpack(ldg16(X + (base + (long)h0 * Wi + w0) * ldx + c0), f00);
      Vec<T>::unpack(ldg16(X + (base + (long)h0 * Wi + w1) * ldx + c0), f01);
      Vec<T>::unpack(ldg16(X + (base + (long)h1 * Wi + w0) * ldx + c0), f10);
      Vec<T>::unpack(ldg16(X + (base + (long)h1 * Wi + w1) * ldx + c0), f11);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if (c0 + k < C) {  // (same expression as upsample_to_nchw_kernel)
          const float top = (1.f - lw) * f00[k] + lw * f01[k];
          const float bot = (1.f - lw) * f10[k] + lw * f11[k];
          m.feed(c0 + k, (1.f - lh) * top + lh * bot);
        }
      }
    }
    metric_count(mt_cnt, nclass, m, target[i]);
  }
  metric_flush(mt_cnt, ncnt, out);
}

static int metric_grid(long total) {
  long g = (total + MT_THREADS - 1) / MT_THREADS;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace seg

// logits: fp32 NCHW [N, C, H, W] (what SegBaseModel.evaluate returns); target int64 [N, H, W]
extern "C" int seg_metric_update_nchw(const float* logits, int N, int C, int H, int W,
                                      const long* target, int nclass, long* counters,
                                      void* stream) {
  using namespace seg;
  SEG_REQUIRE(logits && target && counters && N >= 1 && C >= 1 && H >= 1 && W >= 1,
              "metric_update_nchw: bad arguments");
  SEG_REQUIRE(nclass >= 1 && nclass <= 1024, "metric_update_nchw: nclass=%d", nclass);
  const long plane = (long)H * W;
  hipLaunchKernelGGL(metric_nchw_kernel, dim3(metric_grid((long)N * plane)), dim3(MT_THREADS),
                     (2 + 3 * nclass) * sizeof(unsigned), (hipStream_t)stream, logits, target, N, C,
                     plane, nclass, counters);
  return check_launch("metric_update_nchw");
}

// lo: the network's NHWC logits [N, Hi, Wi, C] (row pitch ld); the metric is taken on their
// bilinear upsample to [H, W] without materialising it
extern "C" int seg_metric_update_upsample(int dtype, const void* lo, long ld, int N, int Hi, int Wi,
                                          int C, const long* target, int H, int W,
                                          int align_corners, int nclass, long* counters,
                                          void* stream) {
  using namespace seg;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "metric_update_upsample: bad dtype %d", dtype);
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(lo && target && counters && N >= 1 && C >= 1 && Hi >= 1 && Wi >= 1 && H >= 1 && W >= 1,
              "metric_update_upsample: bad arguments");
  SEG_REQUIRE(ld % vec == 0 && ld >= (long)((C + vec - 1) / vec) * vec,
              "metric_update_upsample: ld=%ld must cover C=%d in whole %d-vectors", ld, C, vec);
  SEG_REQUIRE(nclass >= 1 && nclass <= 1024, "metric_update_upsample: nclass=%d", nclass);
  const float sh = host_scale(Hi, H, align_corners), sw = host_scale(Wi, W, align_corners);
  const dim3 grid(metric_grid((long)N * H * W));
  const size_t lds = (2 + 3 * nclass) * sizeof(unsigned);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((metric_upsample_kernel<bf16_t>), grid, dim3(MT_THREADS), lds,
                       (hipStream_t)stream, (const bf16_t*)lo, ld, N, Hi, Wi, C, target, H, W, sh,
                       sw, align_corners, nclass, counters);
  else
    hipLaunchKernelGGL((metric_upsample_kernel<float>), grid, dim3(MT_THREADS), lds,
                       (hipStream_t)stream, (const float*)lo, ld, N, Hi, Wi, C, target, H, W, sh, sw,
                       align_corners, nclass, counters);
  return check_launch("metric_update_upsample");
}
