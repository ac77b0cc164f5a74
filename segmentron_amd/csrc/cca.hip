// Criss-cross attention (CCNet) on NHWC tensors — the MI355X counterpart of the reference's only
// native code, segmentron/modules/csrc/criss_cross_attention/ca_cuda.cu:8-177 (ca_forward /
// ca_backward / ca_map_forward / ca_map_backward) + the softmax between them
// (segmentron/modules/cc_attention.py:57-72).
//
// For a pixel p = (y, x) of an H x W map the attended set has L = W + H - 1 entries:
//     z <  W : pixel (y, z)                              (its row, itself included)
//     z >= W : pixel (j, x), i = z - W, j = i < y ? i : i + 1   (its column without itself)
// (ca_cuda.cu:21-31).  The reference keeps t / f / g as NCHW and the weights as [N, L, H, W] and
// gives every (pixel, z) or (pixel, channel) its own thread looping over the other index with
// plane-strided loads.  Here everything is NHWC: ONE WAVE PER PIXEL,
//   * "dot" kernels  (energy, and dA in backward): lane = z, the pixel's own channel vector is
//     staged once in LDS, every lane streams its partner pixel's contiguous channel row;
//     the softmax over L (forward) and its backward are wave reductions in the same kernel, so
//     neither the energies nor dA ever reach HBM;
//   * "map" kernels  (aggregation, and dq / dk / dv in backward): lane = 8 channels (bf16) /
//     4 (fp32), loop over z with the weight broadcast; the TRANSPOSED form gathers from the
//     pixels that attend TO p (row: (y, i) with z = x; column: (i, x), i != y, with
//     z = W + (y < i ? y : y - 1); ca_cuda.cu:72-92, 160-177) — gathers, no atomics.
// Accumulation fp32; attention weights fp32 [N, H, W, L].
#include "common.h"

namespace seg {

constexpr int CCA_WAVES = 4, CCA_THREADS = 64 * CCA_WAVES;

struct CcaGeom {
  int N, H, W, L;
};

// partner pixel (row-major index inside the image) of entry z of pixel (y, x)
__device__ __forceinline__ int cca_src(int y, int x, int z, int W) {
  if (z < W) return y * W + z;
  const int i = z - W;
  const int j = i < y ? i : i + 1;
  return j * W + x;
}

// ---- dot kernels: out[p][z] = sum_c a[p][c] * b[src(p, z)][c]
// MODE 0: softmax over z -> attention.      (a = query, b = key)
// MODE 1: softmax backward: dA = scale * dot, dE = att * (dA - sum_z att * dA).  (a = dout, b = value)
template <typename T, int MODE>
__global__ __launch_bounds__(CCA_THREADS) void cca_dot_kernel(
    const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb, int C, CcaGeom g,
    const float* __restrict__ att, const float* __restrict__ scale, float* __restrict__ out) {
  constexpr int VEC = Vec<T>::N;
  extern __shared__ float cca_smem[];  // [CCA_WAVES][C] : the pixel's own channel vector
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long npix = (long)g.N * g.H * g.W;
  const long p = (long)blockIdx.x * CCA_WAVES + wave;
  float* mine = cca_smem + wave * C;
  if (p < npix) {
    for (int c = lane * VEC; c < C; c += 64 * VEC) {
      float f[VEC];
      Vec<T>::unpack(ldg16(a + p * lda + c), f);
#pragma unroll
      for (int k = 0; k < VEC; ++k) mine[c + k] = f[k];
    }
  }
  __syncthreads();
  if (p >= npix) return;
  const int hw = g.H * g.W;
  const int n = (int)(p / hw), r = (int)(p - (long)n * hw);
  const int y = r / g.W, x = r - y * g.W;
  const T* bn = b + (long)n * hw * ldb;
  const float sc = (MODE == 1 && scale) ? *scale : 1.f;
  // every lane owns entries z = lane + 64*i, i < 8 (L <= 512); fully unrolled so e[] stays in
  // registers — entries past L read the partner of entry 0 and are ignored
  constexpr int ZI = 8;
  float e[ZI];
#pragma unroll
  for (int i = 0; i < ZI; ++i) {
    const int z = lane + 64 * i;
    e[i] = 0.f;
    if (64 * i >= g.L) continue;  // (uniform)
    const T* row = bn + (long)cca_src(y, x, z < g.L ? z : 0, g.W) * ldb;
    float acc = 0.f;
    for (int c = 0; c < C; c += VEC) {
      float f[VEC];
      Vec<T>::unpack(ldg16(row + c), f);
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc = fmaf(mine[c + k], f[k], acc);
    }
    e[i] = acc * sc;
  }
  float* op = out + p * g.L;
  if (MODE == 0) {
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < ZI; ++i)
      if (lane + 64 * i < g.L) m = fmaxf(m, e[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ZI; ++i) {
      e[i] = (lane + 64 * i < g.L) ? __expf(e[i] - m) : 0.f;
      s += e[i];
    }
    s = wave_sum(s);
    const float inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < ZI; ++i)
      if (lane + 64 * i < g.L) op[lane + 64 * i] = e[i] * inv;
  } else {
    const float* ap = att + p * g.L;
    float av[ZI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ZI; ++i) {
      av[i] = (lane + 64 * i < g.L) ? ap[lane + 64 * i] : 0.f;
      s = fmaf(av[i], e[i], s);
    }
    s = wave_sum(s);
#pragma unroll
    for (int i = 0; i < ZI; ++i)
      if (lane + 64 * i < g.L) op[lane + 64 * i] = av[i] * (e[i] - s);
  }
}

// ---- map kernels: out[p][c] = sum_z w(p, z) * b[partner(p, z)][c]
// TRANSPOSED = false : w = wt[p][z], partner = src(p, z)              (aggregation, dq)
// TRANSPOSED = true  : the pixels q that attend to p, w = wt[q][z'] (dv, dk)
// Epilogue: out = (gamma ? *gamma : 1) * acc (+ res[p][c]); `raw` (nullable) receives acc itself.
template <typename T, bool TRANSPOSED>
__global__ __launch_bounds__(CCA_THREADS) void cca_map_kernel(
    const float* __restrict__ wt, const T* __restrict__ b, long ldb, int C, CcaGeom g,
    const float* __restrict__ gamma, const T* __restrict__ res, long ldres, T* __restrict__ out,
    long ldo, T* __restrict__ raw, long ldraw) {
  constexpr int VEC = Vec<T>::N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long npix = (long)g.N * g.H * g.W;
  const long p = (long)blockIdx.x * CCA_WAVES + wave;
  if (p >= npix) return;
  const int hw = g.H * g.W;
  const int n = (int)(p / hw), r = (int)(p - (long)n * hw);
  const int y = r / g.W, x = r - y * g.W;
  const T* bn = b + (long)n * hw * ldb;
  const float* wn = wt + (long)n * hw * g.L;
  const float gm = gamma ? *gamma : 1.f;
  for (int c = lane * VEC; c < C; c += 64 * VEC) {
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    if (!TRANSPOSED) {
      const float* wp = wn + (long)r * g.L;
      for (int z = 0; z < g.L; ++z) {
        const float w = wp[z];
        float f[VEC];
        Vec<T>::unpack(ldg16(bn + (long)cca_src(y, x, z, g.W) * ldb + c), f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(w, f[k], acc[k]);
      }
    } else {
      for (int i = 0; i < g.W; ++i) {  // row: pixel (y, i) looks at p through its entry z = x
        const int q = y * g.W + i;
        const float w = wn[(long)q * g.L + x];
        float f[VEC];
        Vec<T>::unpack(ldg16(bn + (long)q * ldb + c), f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(w, f[k], acc[k]);
      }
      for (int i = 0; i < g.H; ++i) {  // column: pixel (i, x), i != y, entry W + (y < i ? y : y-1)
        if (i == y) continue;
        const int q = i * g.W + x;
        const float w = wn[(long)q * g.L + g.W + (y < i ? y : y - 1)];
        float f[VEC];
        Vec<T>::unpack(ldg16(bn + (long)q * ldb + c), f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(w, f[k], acc[k]);
      }
    }
    if (raw) Vec<T>::store(raw + p * ldraw + c, acc);
    float o[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = gm * acc[k];
    if (res) {
      float f[VEC];
      Vec<T>::unpack(ldg16(res + p * ldres + c), f);
#pragma unroll
      for (int k = 0; k < VEC; ++k) o[k] += f[k];
    }
    Vec<T>::store(out + p * ldo + c, o);
  }
}

static int cca_check(const char* what, int dtype, int N, int H, int W, int C, long ld) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "%s: bad dtype %d", what, dtype);
  SEG_REQUIRE(N >= 1 && H >= 1 && W >= 1 && C >= 1, "%s: empty problem", what);
  SEG_REQUIRE(C % vec == 0 && ld % vec == 0 && ld >= C, "%s: C=%d / ld=%ld must be multiples of %d",
              what, C, ld, vec);
  SEG_REQUIRE(H + W - 1 <= 512, "%s: H + W - 1 = %d exceeds 512 attended entries", what, H + W - 1);
  SEG_REQUIRE((long)N * H * W < (1L << 31), "%s: too many pixels", what);
  return 0;
}

}  // namespace seg

// attention[N,H,W,L] (fp32) = softmax_z( q[p] . k[src(p, z)] ), L = W + H - 1
// (= F.softmax(ca_weight(q, k), 1) of cc_attention.py:65-66 in [N, H, W, L] layout)
extern "C" int seg_cca_attention(int dtype, const void* q, long ldq, const void* k, long ldk, int N,
                                 int H, int W, int C, float* att, void* stream) {
  using namespace seg;
  if (cca_check("cca_attention", dtype, N, H, W, C, ldq) || cca_check("cca_attention", dtype, N, H, W, C, ldk))
    return 1;
  const CcaGeom g = {N, H, W, H + W - 1};
  const long npix = (long)N * H * W;
  const dim3 grid((unsigned)((npix + CCA_WAVES - 1) / CCA_WAVES));
  const size_t lds = (size_t)CCA_WAVES * C * sizeof(float);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((cca_dot_kernel<bf16_t, 0>), grid, dim3(CCA_THREADS), lds, (hipStream_t)stream,
                       (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, C, g, nullptr, nullptr, att);
  else
    hipLaunchKernelGGL((cca_dot_kernel<float, 0>), grid, dim3(CCA_THREADS), lds, (hipStream_t)stream,
                       (const float*)q, ldq, (const float*)k, ldk, C, g, nullptr, nullptr, att);
  return check_launch("cca_attention");
}

// dE[N,H,W,L] (fp32): softmax backward of dA[p][z] = scale * dout[p] . v[src(p, z)]
extern "C" int seg_cca_attention_bwd(int dtype, const void* dout, long lddo, const void* v, long ldv,
                                     int N, int H, int W, int C, const float* att,
                                     const float* scale, float* de, void* stream) {
  using namespace seg;
  if (cca_check("cca_attention_bwd", dtype, N, H, W, C, lddo) ||
      cca_check("cca_attention_bwd", dtype, N, H, W, C, ldv))
    return 1;
  SEG_REQUIRE(att && de, "cca_attention_bwd: null attention / output");
  const CcaGeom g = {N, H, W, H + W - 1};
  const long npix = (long)N * H * W;
  const dim3 grid((unsigned)((npix + CCA_WAVES - 1) / CCA_WAVES));
  const size_t lds = (size_t)CCA_WAVES * C * sizeof(float);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((cca_dot_kernel<bf16_t, 1>), grid, dim3(CCA_THREADS), lds, (hipStream_t)stream,
                       (const bf16_t*)dout, lddo, (const bf16_t*)v, ldv, C, g, att, scale, de);
  else
    hipLaunchKernelGGL((cca_dot_kernel<float, 1>), grid, dim3(CCA_THREADS), lds, (hipStream_t)stream,
                       (const float*)dout, lddo, (const float*)v, ldv, C, g, att, scale, de);
  return check_launch("cca_attention_bwd");
}

// out[p][c] = gamma * sum_z w(p, z) b[partner][c] (+ res[p][c]);  transposed = 0: w = wt[p][z]
// (ca_map_forward, dq); 1: the pixels attending TO p (ca_map_backward_g / ca_backward_kernel_f).
// gamma (device scalar), res and raw (receives the un-scaled sum) may be null.
extern "C" int seg_cca_map(int dtype, const float* wt, const void* b, long ldb, int N, int H, int W,
                           int C, int transposed, const float* gamma, const void* res, long ldres,
                           void* out, long ldo, void* raw, long ldraw, void* stream) {
  using namespace seg;
  if (cca_check("cca_map", dtype, N, H, W, C, ldb) || cca_check("cca_map", dtype, N, H, W, C, ldo))
    return 1;
  SEG_REQUIRE(wt && b && out, "cca_map: null tensor");
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(!res || (ldres % vec == 0 && ldres >= C), "cca_map: bad residual pitch %ld", ldres);
  SEG_REQUIRE(!raw || (ldraw % vec == 0 && ldraw >= C), "cca_map: bad raw pitch %ld", ldraw);
  const CcaGeom g = {N, H, W, H + W - 1};
  const long npix = (long)N * H * W;
  const dim3 grid((unsigned)((npix + CCA_WAVES - 1) / CCA_WAVES));
  hipStream_t st = (hipStream_t)stream;
#define SEG_CCA(T, TR)                                                                          \
  hipLaunchKernelGGL((cca_map_kernel<T, TR>), grid, dim3(CCA_THREADS), 0, st, wt, (const T*)b, ldb, \
                     C, g, gamma, (const T*)res, ldres, (T*)out, ldo, (T*)raw, ldraw)
  if (dtype == DT_BF16) { if (transposed) SEG_CCA(bf16_t, true); else SEG_CCA(bf16_t, false); }
  else { if (transposed) SEG_CCA(float, true); else SEG_CCA(float, false); }
#undef SEG_CCA
  return check_launch("cca_map");
}
