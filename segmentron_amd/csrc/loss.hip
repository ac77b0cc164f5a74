// Fused  bilinear upsample (align_corners) -> log-softmax -> NLL(ignore_index), forward and
// backward, on the LOW-resolution NHWC logits.  Replaces, for the train step of
// tools/train.py:135-146, the chain
//     F.interpolate(head(x), size, mode='bilinear', align_corners=True)    deeplabv3_plus.py:44
//     F.cross_entropy(pred, target, ignore_index=-1)                       solver/loss.py:16-46
// whose intermediate [N, nclass, H, W] float32 logits (319 MB at 2 x 19 x 1025 x 2049) are
// written once and read three times by five ATen kernels.  Here they are never
// materialised: the forward interpolates, soft-maxes and reduces per output pixel; the backward
// recomputes the soft-max and gathers the gradient straight into the low-resolution tensor.
//
// Numerics follow ATen: interpolation in float32 with the tap arithmetic of resize_taps.h and
// the same association  h0l*(w0l*x00 + w1l*x01) + h1l*(w0l*x10 + w1l*x11); log-softmax as
// z - max - log(sum exp(z - max)); loss = sum over valid pixels / number of valid pixels
// (reduction='mean').  Sums are taken in float64 in a fixed order: bitwise deterministic.
#include "common.h"
#include "resize_taps.h"

namespace seg {

constexpr int CE_THREADS = 256;

struct CeArgs {
  const void* lo;        // [N, Hi, Wi, ld] logits, element type T
  const long* target;    // [N, H, W] int64
  long ld;
  int N, Hi, Wi, H, W, C;
  long ignore;
  float sh, sw;
  int align;
};

// NC channels of one pixel; vectors beyond the row pitch (narrow class counts: the buffer is
// padded to a vector multiple of C, not to NC) are not touched
template <typename T, int NC>
__device__ __forceinline__ void ce_load_pixel(const T* __restrict__ p, long ld, float (&f)[NC]) {
  constexpr int VEC = Vec<T>::N;
#pragma unroll
  for (int v = 0; v < NC / VEC; ++v) {
    if (v * VEC < ld) {
      Vec<T>::unpack(ldg16(p + v * VEC), &f[v * VEC]);
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) f[v * VEC + k] = 0.f;
    }
  }
}

// z[c] of output pixel (n, h, w): NC >= C channels are loaded (the buffer is channel-padded)
template <typename T, int NC>
__device__ __forceinline__ void ce_logits(const CeArgs& a, int n, int h, int w, float (&z)[NC]) {
  const T* __restrict__ X = reinterpret_cast<const T*>(a.lo);
  int h0, h1, w0, w1; float lh, lw;
  taps(a.sh, h, a.Hi, a.align, h0, h1, lh);
  taps(a.sw, w, a.Wi, a.align, w0, w1, lw);
  const long base = (long)n * a.Hi * a.Wi;
  // the four taps stay PACKED until they are combined, one channel vector at a time: 4 * NC
  // unpacked floats alive at once (96 registers for 24 classes) cost the backward kernel its
  // occupancy; all loads are still issued before the first use
  constexpr int VEC = Vec<T>::N, NV = NC / VEC;
  const T* __restrict__ p00 = X + (base + (long)h0 * a.Wi + w0) * a.ld;
  const T* __restrict__ p01 = X + (base + (long)h0 * a.Wi + w1) * a.ld;
  const T* __restrict__ p10 = X + (base + (long)h1 * a.Wi + w0) * a.ld;
  const T* __restrict__ p11 = X + (base + (long)h1 * a.Wi + w1) * a.ld;
  const float h0l = 1.f - lh, w0l = 1.f - lw;
  // one vector ahead: the next vector's four loads are in flight while this one is combined
  uint4 q00 = ldg16(p00), q01 = ldg16(p01), q10 = ldg16(p10), q11 = ldg16(p11);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const uint4 c00 = q00, c01 = q01, c10 = q10, c11 = q11;
    if (v + 1 < NV) {
      const int off = ((v + 1) * VEC < a.ld) ? (v + 1) * VEC : 0;  // beyond the pitch: masked below
      q00 = ldg16(p00 + off); q01 = ldg16(p01 + off);
      q10 = ldg16(p10 + off); q11 = ldg16(p11 + off);
    }
    const bool in = v * VEC < a.ld;
    if constexpr (sizeof(T) == 2) {
      // bf16: two channels per dword, combined pair by pair (8 temporaries instead of 32)
      const unsigned d00[4] = {c00.x, c00.y, c00.z, c00.w}, d01[4] = {c01.x, c01.y, c01.z, c01.w};
      const unsigned d10[4] = {c10.x, c10.y, c10.z, c10.w}, d11[4] = {c11.x, c11.y, c11.z, c11.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a0 = __uint_as_float(d00[k] << 16), a1 = __uint_as_float(d00[k] & 0xFFFF0000u);
        const float b0 = __uint_as_float(d01[k] << 16), b1 = __uint_as_float(d01[k] & 0xFFFF0000u);
        const float e0 = __uint_as_float(d10[k] << 16), e1 = __uint_as_float(d10[k] & 0xFFFF0000u);
        const float g0 = __uint_as_float(d11[k] << 16), g1 = __uint_as_float(d11[k] & 0xFFFF0000u);
        const float t0 = h0l * (w0l * a0 + lw * b0) + lh * (w0l * e0 + lw * g0);
        const float t1 = h0l * (w0l * a1 + lw * b1) + lh * (w0l * e1 + lw * g1);
        z[v * VEC + 2 * k] = in ? t0 : 0.f;
        z[v * VEC + 2 * k + 1] = in ? t1 : 0.f;
      }
    } else {
      float f00[VEC], f01[VEC], f10[VEC], f11[VEC];
      Vec<T>::unpack(c00, f00); Vec<T>::unpack(c01, f01);
      Vec<T>::unpack(c10, f10); Vec<T>::unpack(c11, f11);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float t = h0l * (w0l * f00[k] + lw * f01[k]) + lh * (w0l * f10[k] + lw * f11[k]);
        z[v * VEC + k] = in ? t : 0.f;
      }
    }
  }
}

// forward: partial[block] = (sum of -log p_target over the block's valid pixels, valid count)
template <typename T, int NC>
__global__ __launch_bounds__(CE_THREADS) void ce_fwd_kernel(const CeArgs a, double* partial) {
  __shared__ double red[2][CE_THREADS / 64];
  const long total = (long)a.N * a.H * a.W;
  double lsum = 0.0, lcnt = 0.0;
  for (long i = (long)blockIdx.x * CE_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * CE_THREADS) {
    const long t = a.target[i];
    // targets outside [0, C) that are not ignore_index (torch raises a device assert for them)
    // are left out of the sum AND the count, like ignored pixels — never counted with z_t = 0.
    // (No branch on the target: its load and the logit taps go out together.)
    const bool valid = !(t == a.ignore || t < 0 || t >= a.C);
    const int w = (int)(i % a.W);
    const long q = i / a.W;
    const int h = (int)(q % a.H), n = (int)(q / a.H);
    float z[NC];
    ce_logits<T, NC>(a, n, h, w, z);
    float m = z[0];
#pragma unroll
    for (int c = 1; c < NC; ++c) m = (c < a.C) ? fmaxf(m, z[c]) : m;
    float s = 0.f, zt = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c < a.C) s += expf(z[c] - m);
      if (c == (int)t) zt = z[c];
    }
    if (valid) {
      lsum += (double)(logf(s) + m - zt);
      lcnt += 1.0;
    }
  }
  // block reduction in a fixed order (wave butterfly, then the waves in index order)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lsum += __shfl_xor(lsum, o, 64);
    lcnt += __shfl_xor(lcnt, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = lsum; red[1][wave] = lcnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0, c = 0.0;
    for (int k = 0; k < CE_THREADS / 64; ++k) { s += red[0][k]; c += red[1][k]; }
    partial[2 * blockIdx.x] = s;
    partial[2 * blockIdx.x + 1] = c;
  }
}

// out[0] = loss (mean over valid pixels), out[1] = 1 / valid count (0 if none), both float32
__global__ void ce_finalize_kernel(const double* partial, int nblocks, float* out) {
  __shared__ double red[2][256];
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) { s += partial[2 * i]; c += partial[2 * i + 1]; }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double cnt = red[1][0];
    out[0] = cnt > 0.0 ? (float)(red[0][0] / cnt) : nanf("");  // torch: mean over nothing = nan
    out[1] = cnt > 0.0 ? (float)(1.0 / cnt) : 0.f;
  }
}

// backward: one block per LOW-resolution tile of LT x LT pixels.  The output rows that touch the
// tile are processed in CHUNKS of CH rows.  Per chunk — phase 1: dz = (softmax - onehot) of the
// chunk's output pixels (redundantly with the neighbouring tiles at the borders) into LDS;
// phase 2a: reduced along w with the tile columns' interpolation weights; phase 2b: every
// (pixel, channel) of the tile adds its weighted rows of the chunk to a register accumulator, in
// ascending row order — the same fixed summation order as a whole-footprint gather.
// LT = low-res tile edge, HT_MAX = most output rows / columns that can touch LT low-res rows at
// up to 4.1x upsampling: (LT + 1.5) * 4.1 + 5.  LDS: NC * CH * (HT_MAX + LT) floats — the first
// version staged the whole NC x HT_MAX x HT_MAX footprint (137 KiB: ONE 512-thread block per CU,
// 7396 blocks in 29 rounds of 17 us = 0.5 ms per C3 step, the largest single kernel of the step);
// with 12-row chunks three blocks share a CU and overlap each other's load / LDS phases.
constexpr int CE_BWD_THREADS = 512;
template <typename T, int NC, int LT, int HT_MAX, int CH>
__global__ __launch_bounds__(CE_BWD_THREADS, 4) void ce_bwd_kernel(const CeArgs a, const float* gscale,
                                                                   const float* gout, void* dlo,
                                                                   long lddlo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_ce[];
  const int tiles_w = (a.Wi + LT - 1) / LT, tiles_h = (a.Hi + LT - 1) / LT;
  const int tw = blockIdx.x % tiles_w, th = (blockIdx.x / tiles_w) % tiles_h;
  const int n = blockIdx.x / (tiles_w * tiles_h);
  const int i0 = th * LT, j0 = tw * LT;
  const int i1 = min(a.Hi, i0 + LT) - 1, j1 = min(a.Wi, j0 + LT) - 1;
  // output range touching rows [i0, i1] / columns [j0, j1]
  int hlo, hhi, wlo, whi, t0, t1;
  cand_range(a.sh, i0, a.H, a.align, hlo, t1);
  cand_range(a.sh, i1, a.H, a.align, t0, hhi);
  cand_range(a.sw, j0, a.W, a.align, wlo, t1);
  cand_range(a.sw, j1, a.W, a.align, t0, whi);
  const int nh = hhi - hlo + 1, nw = whi - wlo + 1;  // <= HT_MAX (checked on the host)
  float* dz = reinterpret_cast<float*>(smem_ce);                  // [NC][CH][HT_MAX]
  float* tmp = dz + (long)NC * CH * HT_MAX;                       // [NC][CH][LT]
  float* wth = tmp + (long)NC * CH * LT;                          // [HT_MAX][LT] row weights
  float* wtw = wth + HT_MAX * LT;                                 // [HT_MAX][LT] column weights
  int* rng = reinterpret_cast<int*>(wtw + HT_MAX * LT);           // [2][LT][2] candidate ranges
  const float g = gout[0] * gscale[1];  // dLoss * (1 / valid count)
  // ---- interpolation weights of every (output row, tile row) / (output column, tile column)
  for (int p = threadIdx.x; p < 2 * HT_MAX * LT; p += CE_BWD_THREADS) {
    const int which = p / (HT_MAX * LT), q = p - which * HT_MAX * LT;
    const int oo = q / LT, ii = q - oo * LT;
    if (which == 0) wth[q] = oo < nh ? tap_weight(a.sh, hlo + oo, a.Hi, a.align, i0 + ii) : 0.f;
    else wtw[q] = oo < nw ? tap_weight(a.sw, wlo + oo, a.Wi, a.align, j0 + ii) : 0.f;
  }
  if (threadIdx.x < 2 * LT) {
    const int which = threadIdx.x / LT, ii = threadIdx.x - which * LT;
    int clo, chi;
    if (which == 0) {
      cand_range(a.sh, i0 + ii, a.H, a.align, clo, chi);
      clo = max(clo, hlo) - hlo; chi = min(chi, hhi) - hlo;
    } else {
      cand_range(a.sw, j0 + ii, a.W, a.align, clo, chi);
      clo = max(clo, wlo) - wlo; chi = min(chi, whi) - wlo;
    }
    rng[(which * LT + ii) * 2] = clo;
    rng[(which * LT + ii) * 2 + 1] = chi;
  }
  const int lw_n = j1 - j0 + 1, lh_n = i1 - i0 + 1;
  // this thread's (tile pixel, channel) items of phase 2b: p = (ii * LT + jj) * lddlo + c
  constexpr int MAXI = (LT * LT * 32 + CE_BWD_THREADS - 1) / CE_BWD_THREADS;
  const int ld = (int)lddlo, nitems = LT * LT * ld;
  float acc[MAXI];
#pragma unroll
  for (int k = 0; k < MAXI; ++k) acc[k] = 0.f;

  for (int r0 = 0; r0 < nh; r0 += CH) {
    const int rows = min(CH, nh - r0);
    // ---- phase 1: the chunk's output pixels (at most one per thread: rows * nw <= 432)
#pragma unroll 1
    for (int p = threadIdx.x; p < rows * nw; p += CE_BWD_THREADS) {
      const int hh = p / nw, ww = p - hh * nw;
      const int h = hlo + r0 + hh, w = wlo + ww;
      // the target and the four logit taps are requested together (the softmax of an ignored
      // pixel — 5 % of Cityscapes-like labels — is computed and discarded: a branch on the
      // target would put a second, dependent memory round trip behind the first)
      const long t = a.target[((long)n * a.H + h) * a.W + w];
      float z[NC];
      ce_logits<T, NC>(a, n, h, w, z);
      const bool valid = t != a.ignore && t >= 0 && t < a.C;
      float m = z[0];
#pragma unroll
      for (int c = 1; c < NC; ++c) m = (c < a.C) ? fmaxf(m, z[c]) : m;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        z[c] = (c < a.C) ? expf(z[c] - m) : 0.f;
        s += z[c];
      }
      const float inv = valid ? g / s : 0.f, hot = valid ? g : 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) z[c] = z[c] * inv - (c == (int)t ? hot : 0.f);
#pragma unroll
      for (int c = 0; c < NC; ++c) dz[((long)c * CH + hh) * HT_MAX + ww] = z[c];
    }
    __syncthreads();  // (also orders the weight tables / the previous chunk's phase 2b)
    // ---- phase 2a: along w.  tmp[c][hh][jj] = sum_ww wtw[ww][jj] * dz[c][hh][ww]
    for (int p = threadIdx.x; p < a.C * rows * LT; p += CE_BWD_THREADS) {
      const int jj = p % LT, hh = (p / LT) % rows, c = p / (LT * rows);
      float v = 0.f;
      if (jj < lw_n) {
        const int clo = rng[(LT + jj) * 2], chi = rng[(LT + jj) * 2 + 1];
        const float* row = dz + ((long)c * CH + hh) * HT_MAX;
        for (int ww = clo; ww <= chi; ++ww) v = fmaf(wtw[ww * LT + jj], row[ww], v);
      }
      tmp[((long)c * CH + hh) * LT + jj] = v;
    }
    __syncthreads();
    // ---- phase 2b: along h, this chunk's rows (ascending: the whole sum runs in row order)
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
      const int p = threadIdx.x + k * CE_BWD_THREADS;
      if (p < nitems) {
        const int c = p % ld, jj = (p / ld) % LT, ii = p / (ld * LT);
        if (c < a.C && ii < lh_n && jj < lw_n) {
          const int clo = max(rng[ii * 2], r0), chi = min(rng[ii * 2 + 1], r0 + rows - 1);
          float v = acc[k];
          for (int hh = clo; hh <= chi; ++hh)
            v = fmaf(wth[hh * LT + ii], tmp[((long)c * CH + (hh - r0)) * LT + jj], v);
          acc[k] = v;
        }
      }
    }
  }
  // ---- store (channels >= C are written as zero padding)
  T* __restrict__ D = reinterpret_cast<T*>(dlo);
#pragma unroll
  for (int k = 0; k < MAXI; ++k) {
    const int p = threadIdx.x + k * CE_BWD_THREADS;
    if (p < nitems) {
      const int c = p % ld, jj = (p / ld) % LT, ii = p / (ld * LT);
      if (ii < lh_n && jj < lw_n)
        Vec<T>::store1(D + (((long)n * a.Hi + i0 + ii) * a.Wi + j0 + jj) * lddlo + c, acc[k]);
    }
  }
}

template <int NC, int LT, int HT_MAX, int CH> constexpr size_t ce_bwd_lds() {
  return ((size_t)NC * CH * HT_MAX + (size_t)NC * CH * LT + 2 * (size_t)HT_MAX * LT) *
             sizeof(float) + 4 * (size_t)LT * sizeof(int);
}
// <= 24 classes: 6 x 6 tiles, 12-row chunks (49 KiB of LDS); <= 32 classes: 4 x 4 tiles, 10-row
// chunks (41 KiB): three blocks per CU
constexpr int CE_LT24 = 6, CE_HT24 = 36, CE_CH24 = 12, CE_LT32 = 4, CE_HT32 = 28, CE_CH32 = 10;
static_assert(ce_bwd_lds<24, CE_LT24, CE_HT24, CE_CH24>() <= 53 * 1024, "LDS budget (3 blocks/CU)");
static_assert(ce_bwd_lds<32, CE_LT32, CE_HT32, CE_CH32>() <= 53 * 1024, "LDS budget (3 blocks/CU)");
// r05: up to 8.1x (output-stride-8 heads: PSPNet / DANet / CCNet logits at 129x257 for a 1025x2049
// image; the PSPNet train step paid 6.5 ms of 72 for materialised logits + torch's log-softmax /
// NLL kernels): the same kernel on 3 x 3 (2 x 2) low-resolution tiles, (LT + 1.5) * 8.1 + 5 rows
constexpr int CE8_LT24 = 3, CE8_HT24 = 42, CE8_LT32 = 2, CE8_HT32 = 34;
static_assert(ce_bwd_lds<24, CE8_LT24, CE8_HT24, CE_CH24>() <= 53 * 1024, "LDS budget (3 blocks/CU)");
static_assert(ce_bwd_lds<32, CE8_LT32, CE8_HT32, CE_CH32>() <= 53 * 1024, "LDS budget (3 blocks/CU)");
constexpr float CE_MAX_SCALE = 4.1f, CE8_MAX_SCALE = 8.1f;
static_assert((CE_LT24 + 1.5f) * CE_MAX_SCALE + 5 <= CE_HT24 + 1 && (CE_LT32 + 1.5f) * CE_MAX_SCALE + 5 <= CE_HT32 + 1, "");
static_assert((CE8_LT24 + 1.5f) * CE8_MAX_SCALE + 5 <= CE8_HT24 && (CE8_LT32 + 1.5f) * CE8_MAX_SCALE + 5 <= CE8_HT32, "");

template <typename T, int NC, int LT, int HT, int CH>
static int launch_ce_bwd(int blocks, hipStream_t st, const CeArgs& a, const float* loss_out,
                         const float* grad_out, void* dlo, long lddlo) {
  constexpr size_t lds = ce_bwd_lds<NC, LT, HT, CH>();
  static const int once = (int)hipFuncSetAttribute(
      reinterpret_cast<const void*>(&ce_bwd_kernel<T, NC, LT, HT, CH>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  SEG_REQUIRE(once == 0, "upsample_ce_bwd: cannot reserve %d bytes of LDS", (int)lds);
  hipLaunchKernelGGL((ce_bwd_kernel<T, NC, LT, HT, CH>), dim3(blocks), dim3(CE_BWD_THREADS), lds, st,
                     a, loss_out, grad_out, dlo, lddlo);
  return 0;
}

}  // namespace seg

// loss_out: float32[2] = (mean loss, 1 / valid count); ws: >= 2 * seg_upsample_ce_blocks doubles
extern "C" int seg_upsample_ce_blocks(int N, int H, int W) {
  const long total = (long)N * H * W;
  long b = (total + seg::CE_THREADS - 1) / seg::CE_THREADS;
  return (int)(b > 4096 ? 4096 : b);
}

extern "C" int seg_upsample_ce_fwd(int dtype, const void* lo, long ld, int N, int Hi, int Wi, int C,
                                   const long* target, int H, int W, long ignore_index,
                                   int align_corners, double* ws, float* loss_out, void* stream) {
  using namespace seg;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "upsample_ce_fwd: bad dtype %d", dtype);
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(C >= 1 && C <= 32 && ld % vec == 0 && ld >= (C + vec - 1) / vec * vec,
              "upsample_ce_fwd: C=%d must be <= 32 and the pitch %ld a padded multiple of %d", C,
              ld, vec);
  SEG_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && H > 0 && W > 0, "upsample_ce_fwd: empty problem");
  CeArgs a;
  a.lo = lo; a.target = target; a.ld = ld; a.N = N; a.Hi = Hi; a.Wi = Wi; a.H = H; a.W = W;
  a.C = C; a.ignore = ignore_index; a.align = align_corners;
  a.sh = host_scale(Hi, H, align_corners); a.sw = host_scale(Wi, W, align_corners);
  const int blocks = seg_upsample_ce_blocks(N, H, W);
  hipStream_t st = (hipStream_t)stream;
  const int nc = C <= 24 ? 24 : 32;
  if (dtype == DT_BF16) {
    if (nc == 24) hipLaunchKernelGGL((ce_fwd_kernel<bf16_t, 24>), dim3(blocks), dim3(CE_THREADS), 0, st, a, ws);
    else hipLaunchKernelGGL((ce_fwd_kernel<bf16_t, 32>), dim3(blocks), dim3(CE_THREADS), 0, st, a, ws);
  } else {
    if (nc == 24) hipLaunchKernelGGL((ce_fwd_kernel<float, 24>), dim3(blocks), dim3(CE_THREADS), 0, st, a, ws);
    else hipLaunchKernelGGL((ce_fwd_kernel<float, 32>), dim3(blocks), dim3(CE_THREADS), 0, st, a, ws);
  }
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, ws, blocks, loss_out);
  return check_launch("upsample_ce_fwd");
}

extern "C" int seg_upsample_ce_bwd(int dtype, const void* lo, long ld, int N, int Hi, int Wi, int C,
                                   const long* target, int H, int W, long ignore_index,
                                   int align_corners, const float* loss_out, const float* grad_out,
                                   void* dlo, long lddlo, void* stream) {
  using namespace seg;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "upsample_ce_bwd: bad dtype %d", dtype);
  const int vec = dtype == DT_BF16 ? 8 : 4;
  // the per-thread accumulators of ce_bwd_kernel cover LT*LT*32 items: a wider pitch would leave
  // part of the gradient tile unwritten
  SEG_REQUIRE(C >= 1 && C <= 32 && ld % vec == 0 && lddlo >= C && lddlo <= 32,
              "upsample_ce_bwd: bad C / pitch (1 <= C <= lddlo <= 32)");
  SEG_REQUIRE(H >= Hi && W >= Wi, "upsample_ce_bwd: the fused loss is for UP-sampling heads");
  // a low-res tile row is touched by at most HT_MAX output rows / columns up to 4.1x (8.1x)
  const float sh = host_scale(Hi, H, align_corners), sw = host_scale(Wi, W, align_corners);
  const float smin = fminf(sh > 0.f ? sh : 1.f, sw > 0.f ? sw : 1.f);
  SEG_REQUIRE(1.f / smin <= CE8_MAX_SCALE,
              "upsample_ce_bwd: scale factor %.2f too large for the fused backward", 1.f / smin);
  const bool wide = 1.f / smin > CE_MAX_SCALE;  // 4.1x .. 8.1x: the small-tile instances
  CeArgs a;
  a.lo = lo; a.target = target; a.ld = ld; a.N = N; a.Hi = Hi; a.Wi = Wi; a.H = H; a.W = W;
  a.C = C; a.ignore = ignore_index; a.align = align_corners; a.sh = sh; a.sw = sw;
  hipStream_t st = (hipStream_t)stream;
  const int nc = C <= 24 ? 24 : 32;
  const int lt = wide ? (nc == 24 ? CE8_LT24 : CE8_LT32) : (nc == 24 ? CE_LT24 : CE_LT32);
  const int blocks = N * ((Hi + lt - 1) / lt) * ((Wi + lt - 1) / lt);
  int rc;
  if (wide && dtype == DT_BF16) {
    rc = nc == 24 ? launch_ce_bwd<bf16_t, 24, CE8_LT24, CE8_HT24, CE_CH24>(blocks, st, a, loss_out, grad_out, dlo, lddlo)
                  : launch_ce_bwd<bf16_t, 32, CE8_LT32, CE8_HT32, CE_CH32>(blocks, st, a, loss_out, grad_out, dlo, lddlo);
  } else if (wide) {
    rc = nc == 24 ? launch_ce_bwd<float, 24, CE8_LT24, CE8_HT24, CE_CH24>(blocks, st, a, loss_out, grad_out, dlo, lddlo)
                  : launch_ce_bwd<float, 32, CE8_LT32, CE8_HT32, CE_CH32>(blocks, st, a, loss_out, grad_out, dlo, lddlo);
  } else if (dtype == DT_BF16) {
    rc = nc == 24 ? launch_ce_bwd<bf16_t, 24, CE_LT24, CE_HT24, CE_CH24>(blocks, st, a, loss_out, grad_out, dlo, lddlo)
                  : launch_ce_bwd<bf16_t, 32, CE_LT32, CE_HT32, CE_CH32>(blocks, st, a, loss_out, grad_out, dlo, lddlo);
  } else {
    rc = nc == 24 ? launch_ce_bwd<float, 24, CE_LT24, CE_HT24, CE_CH24>(blocks, st, a, loss_out, grad_out, dlo, lddlo)
                  : launch_ce_bwd<float, 32, CE_LT32, CE_HT32, CE_CH32>(blocks, st, a, loss_out, grad_out, dlo, lddlo);
  }
  if (rc) return rc;
  return check_launch("upsample_ce_bwd");
}
