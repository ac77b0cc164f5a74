// Depthwise 3x3, stride 1, WIDE dilation (ASPP rates 6 / 12 / 18, or 12 / 24 / 36 at output
// stride 8; segmentron/modules/module.py:39-47): forward (+ BatchNorm statistics) and the fused
// one-pass backward (masked data gradient + weight-gradient partials + BatchNorm-backward sums).
//
// The strip kernels fetch nine taps per output from L2 (each input vector nine times): 0.9-1.4
// TB/s on the 68 MB ASPP tensors.  A square LDS tile would need a halo of `dil` pixels on every
// side (dil 18: 4.5x the tile).  Here a block walks a CHAIN of rows ph, ph + dil, ph + 2 dil, ...
// (one phase ph < dil of one image, a segment of TW <= 160 pixels, 32 channels) with a ring of
// THREE rows (TW + 2*dil pixels each, storage dtype) in LDS: output row h needs rows h - dil, h,
// h + dil, and the next output row of the chain, h + dil, re-uses two of them — every input row
// is staged ONCE per chain, the column taps +-dil come out of LDS: L2 reads drop from nine per
// output to (TW + 2 dil) / TW.  Thread = (4 channels, 1 of 32 pixel lanes); a wave reads 8
// pixels x 64 contiguous bytes per tap — conflict-free.  Persistent blocks over chains
// (XCD-local, channel block fastest), accumulators in registers, one deterministic block
// reduction.
//   forward : tile = act(x) (BatchNorm/ReLU prologue applied while staging, zero outside the
//             image as the reference's zero padding), y = sum_taps tile * w, statistics of y
//   backward: tile = dy (raw); g[h][w] = mask * sum dy[h-(kh-1)d][w-(kw-1)d] w[kh][kw];
//             accw[kh][kw] += dy_tap * act(x[h][w]); sums (g, g * x_raw)
#include "common.h"
#include "dwconv_tiled.h"

namespace seg {

constexpr int RW_THREADS = 256, RW_CQ = 8, RW_PL = RW_THREADS / RW_CQ, RW_CH = RW_CQ * 4;  // 32 channels / block

struct DwRowArgs {
  const void* src;     // tile source: forward input x / backward dy
  const void* xc;      // backward: the forward input x (raw, + prologue); forward: unused
  void* out;           // forward y / backward g
  const float* w;      // [9][C] tap-major fp32
  const float* sc; const float* sh;
  float* partial_a;    // forward: [gy][2][C] statistics (nullable); backward: [gy][9][C]
  float* partial_bn;   // backward: [gy][2][C] (nullable)
  long ldsrc, ldxc, ldout;
  int N, H, W, C, dil, pro_mode, TW, ntw, ntiles;
};

template <typename T, bool BWD>
__global__ __launch_bounds__(RW_THREADS, BWD ? 2 : 3) void dwconv_row_kernel(const DwRowArgs a) {
  using V = Vec<T>;     // 16-byte staging vectors
  using H4 = HVec<T>;   // 4-channel compute vectors
  constexpr int VN = V::N, VPP = RW_CH / VN;  // staging vectors per pixel (8 bf16 / 16 fp32)
  constexpr int NACC = BWD ? 9 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char rw_smem[];
  T* tile = reinterpret_cast<T*>(rw_smem);  // [3][TWH][64]
  const int tid = threadIdx.x;
  const int flat = blockIdx.x + gridDim.x * blockIdx.y;
  const int L = xcd_remap(flat, gridDim.x * gridDim.y);
  const int by = L / (int)gridDim.x, bx = L - by * (int)gridDim.x;
  const int cq = tid & (RW_CQ - 1), pl = tid / RW_CQ;
  const int cbase = bx * RW_CH;
  const int c0 = cbase + cq * 4;
  const bool cok = c0 < a.C;
  const int c0s = cok ? c0 : 0;
  const int d = a.dil, TWH = a.TW + 2 * d;
  const T* __restrict__ SRC = reinterpret_cast<const T*>(a.src);
  const T* __restrict__ XC = reinterpret_cast<const T*>(a.xc);
  T* __restrict__ OUT = reinterpret_cast<T*>(a.out);

  // staging: this thread's vector slot inside a pixel never changes (256 % VPP == 0)
  const int sv = tid % VPP;
  const int sc0 = cbase + sv * VN;
  const bool sok = sc0 < a.C;
  float ps[VN], pt[VN];
#pragma unroll
  for (int i = 0; i < VN; ++i) { ps[i] = 1.f; pt[i] = 0.f; }
  float sc[4], sh[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
  if (a.pro_mode & PRO_AFFINE) {
    if (!BWD) {
      load_params<VN>(a.sc, sok ? sc0 : 0, ps);
      load_params<VN>(a.sh, sok ? sc0 : 0, pt);
    } else {
      load_params<4>(a.sc, c0s, sc);
      load_params<4>(a.sh, c0s, sh);
    }
  }
  float wv[9][4];
#pragma unroll
  for (int k = 0; k < 9; ++k) load_params<4>(a.w + (long)k * a.C, c0s, wv[k]);
  float acc[NACC][4], s1[4], s2[4];
#pragma unroll
  for (int k = 0; k < NACC; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[k][i] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s1[i] = s2[i] = 0.f;

  const int nph = d < a.H ? d : a.H;  // phases that own at least one row
  const int nvec_row = TWH * VPP;
  for (int t = by; t < a.ntiles; t += gridDim.y) {
    const int tw = t % a.ntw, tq = t / a.ntw;
    const int ph = tq % nph, n = tq / nph;
    const int w0 = tw * a.TW;
    // stage image row hh (zeros outside the image) into ring slot `slot`
    auto stage_row = [&](int slot, int hh) {
      T* dst = tile + (long)slot * TWH * RW_CH;
      const bool rok = sok && hh >= 0 && hh < a.H;
      for (int idx = tid; idx < nvec_row; idx += RW_THREADS) {
        const int px = idx / VPP;
        const int ww = w0 - d + px;
        const bool ok = rok && ww >= 0 && ww < a.W;
        uint4 v = ldg16(SRC + (ok ? (((long)n * a.H + hh) * a.W + ww) * a.ldsrc + sc0 : 0));
        if (!BWD && a.pro_mode != PRO_NONE) {
          float f[VN];
          V::unpack(v, f);
          apply_prologue_regs<VN>(f, a.pro_mode, ps, pt);
          v = V::pack(f);
        }
        v = mask_u4(v, ok);
        *reinterpret_cast<uint4*>(dst + (long)px * RW_CH + sv * VN) = v;
      }
    };
    __syncthreads();  // the previous chain's readers are done
    stage_row(0, ph - d);
    stage_row(1, ph);
    int k = 0;
    for (int h = ph; h < a.H; h += d, ++k) {
      // rows h - d, h, h + d live in slots k % 3, (k+1) % 3, (k+2) % 3
      const long orow = ((long)n * a.H + h) * a.W;
      // backward: this thread's centre pixels of x, requested BEFORE the row is staged so that
      // their latency hides behind it (TW <= 160 -> at most MAXP pixels per lane)
      constexpr int MAXP = (160 + RW_PL - 1) / RW_PL;
      typename H4::raw_t xraw[MAXP];
      if (BWD) {
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
          const int p = pl + q * RW_PL;
          const bool ok = cok && p < a.TW && w0 + p < a.W;
          xraw[q] = H4::load_raw(XC + (ok ? (orow + w0 + p) * a.ldxc + c0 : 0));
        }
      }
      stage_row((k + 2) % 3, h + d);
      __syncthreads();
      const T* slot_of[3] = {tile + (long)(k % 3) * TWH * RW_CH,
                             tile + (long)((k + 1) % 3) * TWH * RW_CH,
                             tile + (long)((k + 2) % 3) * TWH * RW_CH};
#pragma unroll
      for (int q = 0; q < MAXP; ++q) {
        const int p = pl + q * RW_PL;
        if (!(cok && p < a.TW && w0 + p < a.W)) continue;
        const long poff = (long)(p + d) * RW_CH + cq * 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (!BWD) {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              float f[4];
              H4::load(slot_of[kh] + poff + (long)(kw - 1) * d * RW_CH, f);
#pragma unroll
              for (int i = 0; i < 4; ++i) o[i] = fmaf(f[i], wv[kh * 3 + kw][i], o[i]);
            }
          H4::store(OUT + (orow + w0 + p) * a.ldout + c0, o);
#pragma unroll
          for (int i = 0; i < 4; ++i) {  // statistics from the fp32 accumulators (as the strip kernel)
            acc[0][i] += o[i];
            acc[1][i] = fmaf(o[i], o[i], acc[1][i]);
          }
        } else {
          float xr[4], xa[4];
          H4::unpack_raw(xraw[q], xr);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = xr[i];
            if (a.pro_mode & PRO_AFFINE) v = fmaf(v, sc[i], sh[i]);
            if (a.pro_mode & PRO_RELU) v = fmaxf(v, 0.f);
            if (a.pro_mode & PRO_CLAMP6) v = fminf(v, 6.f);
            xa[i] = v;
          }
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              float f[4];  // dy[h - (kh-1) d][w - (kw-1) d]
              H4::load(slot_of[2 - kh] + poff - (long)(kw - 1) * d * RW_CH, f);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                o[i] = fmaf(f[i], wv[kh * 3 + kw][i], o[i]);
                acc[kh * 3 + kw][i] = fmaf(f[i], xa[i], acc[kh * 3 + kw][i]);
              }
            }
          if (a.pro_mode & PRO_RELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool on = xa[i] > 0.f && (!(a.pro_mode & PRO_CLAMP6) || xa[i] < 6.f);
              o[i] = on ? o[i] : 0.f;
            }
          }
          H4::store(OUT + (orow + w0 + p) * a.ldout + c0, o);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s1[i] += o[i];
            s2[i] = fmaf(o[i], xr[i], s2[i]);
          }
        }
      }
      __syncthreads();  // slot k % 3 (row h - d) is free for row h + 2 d
    }
  }

  // ---- block reduction over the pixel lanes: lanes of a wave = cq (low bits) x 64 / RW_CQ pixel
  // lanes; the four waves through LDS (fixed order)
  constexpr int NR = BWD ? 11 : 2;  // rows per channel quad: taps (+ the two sums) / statistics
  float r[NR][4];
#pragma unroll
  for (int k = 0; k < NACC; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) r[k][i] = acc[k][i];
  if (BWD) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[NR - 2][i] = s1[i]; r[NR - 1][i] = s2[i]; }
  }
#pragma unroll
  for (int k = 0; k < NR; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int m = RW_CQ; m < 64; m <<= 1) r[k][i] += __shfl_xor(r[k][i], m, 64);
    }
  __syncthreads();
  float* red = reinterpret_cast<float*>(rw_smem);  // [4 waves][16 cq][NR][4]
  const int wave = tid >> 6;
  if ((tid & 63) < RW_CQ) {
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[((wave * RW_CQ + cq) * NR + k) * 4 + i] = r[k][i];
  }
  __syncthreads();
  for (int e = tid; e < RW_CQ * NR * 4; e += RW_THREADS) {
    float tot = 0.f;
#pragma unroll
    for (int wv4 = 0; wv4 < 4; ++wv4) tot += red[wv4 * RW_CQ * NR * 4 + e];
    const int lq = e / (NR * 4), k = (e / 4) % NR, i = e & 3;
    const int c = cbase + lq * 4 + i;
    if (c >= a.C) continue;
    if (!BWD) {
      if (a.partial_a != nullptr) a.partial_a[((long)by * 2 + k) * a.C + c] = tot;
    } else if (k < 9) {
      a.partial_a[((long)by * 9 + k) * a.C + c] = tot;
    } else if (a.partial_bn != nullptr) {
      a.partial_bn[((long)by * 2 + (k - 9)) * a.C + c] = tot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
bool dw_row_supported(int stride, int dil) { return stride == 1 && dil > 2 && dil <= 64; }

constexpr size_t RW_LDS_MAX = 120 * 1024;

// segment width: <= 160 pixels, equal segments, and a 3-row ring that fits RW_LDS_MAX
static void row_geom(int dtype, int W, int dil, int& TW, int& ntw) {
  const size_t px_bytes = (size_t)3 * RW_CH * (dtype == DT_BF16 ? 2 : 4);
  ntw = (W + 159) / 160;
  for (;;) {
    TW = (W + ntw - 1) / ntw;
    if ((size_t)(TW + 2 * dil) * px_bytes <= RW_LDS_MAX || TW == 1) break;
    ++ntw;
  }
}

int dw_row_grid_y(int dtype, int C, int N, int H, int W, int dil) {
  int TW, ntw;
  row_geom(dtype, W, dil, TW, ntw);
  const long ntiles = (long)N * (dil < H ? dil : H) * ntw;  // chains
  const int gx = (C + RW_CH - 1) / RW_CH;
  long cap = 768 / gx;  // one resident set (3 blocks per CU); bounds the partial rows
  if (cap < 1) cap = 1;
  return (int)(ntiles < cap ? ntiles : cap);
}

template <typename T, bool BWD>
static int launch_row(DwRowArgs a, int dtype, int grid_y, hipStream_t st) {
  row_geom(dtype, a.W, a.dil, a.TW, a.ntw);
  a.ntiles = a.N * (a.dil < a.H ? a.dil : a.H) * a.ntw;  // chains: (image, phase, segment)
  const size_t tile_b = (size_t)3 * (a.TW + 2 * a.dil) * RW_CH * sizeof(T);
  const size_t red_b = (size_t)4 * RW_CQ * (BWD ? 11 : 2) * 4 * sizeof(float);
  const size_t lds = tile_b > red_b ? tile_b : red_b;
  SEG_REQUIRE(lds <= RW_LDS_MAX, "dwconv3x3 (row): %zu bytes of LDS for dil=%d", lds, a.dil);
  static const int once = (int)hipFuncSetAttribute(
      reinterpret_cast<const void*>(&dwconv_row_kernel<T, BWD>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)RW_LDS_MAX);
  if (once != 0) {
    set_error("dwconv3x3 (row): cannot reserve %zu bytes of LDS", RW_LDS_MAX);
    return 2;
  }
  const dim3 grid((a.C + RW_CH - 1) / RW_CH, grid_y);
  hipLaunchKernelGGL((dwconv_row_kernel<T, BWD>), grid, dim3(RW_THREADS), lds, st, a);
  return check_launch(BWD ? "dwconv3x3_bwd_fused (row)" : "dwconv3x3 (row)");
}

int launch_dw_row_fwd(int dtype, const void* x, long ldx, int N, int H, int W, int C,
                      const float* w9c, int dil, int pro_mode, const float* sc, const float* sh,
                      void* y, long ldy, float* stat_partial, int grid_y, hipStream_t st) {
  SEG_REQUIRE(grid_y == dw_row_grid_y(dtype, C, N, H, W, dil), "dwconv3x3 (row): grid_y %d != %d",
              grid_y, dw_row_grid_y(dtype, C, N, H, W, dil));
  DwRowArgs a = {};
  a.src = x; a.out = y; a.w = w9c; a.sc = sc; a.sh = sh; a.partial_a = stat_partial;
  a.ldsrc = ldx; a.ldout = ldy; a.N = N; a.H = H; a.W = W; a.C = C; a.dil = dil;
  a.pro_mode = pro_mode;
  if (dtype == DT_BF16) return launch_row<bf16_t, false>(a, dtype, grid_y, st);
  return launch_row<float, false>(a, dtype, grid_y, st);
}

int launch_dw_row_bwd(int dtype, const void* dy, long lddy, const void* x, long ldx, int N, int H,
                      int W, int C, const float* w9c, int dil, int pro_mode, const float* sc,
                      const float* sh, void* g, long ldg, float* partial_w, float* partial_bn,
                      int grid_y, hipStream_t st) {
  SEG_REQUIRE(grid_y == dw_row_grid_y(dtype, C, N, H, W, dil),
              "dwconv3x3_bwd_fused (row): grid_y %d != %d", grid_y,
              dw_row_grid_y(dtype, C, N, H, W, dil));
  DwRowArgs a = {};
  a.src = dy; a.xc = x; a.out = g; a.w = w9c; a.sc = sc; a.sh = sh;
  a.partial_a = partial_w; a.partial_bn = partial_bn;
  a.ldsrc = lddy; a.ldxc = ldx; a.ldout = ldg; a.N = N; a.H = H; a.W = W; a.C = C; a.dil = dil;
  a.pro_mode = pro_mode;
  if (dtype == DT_BF16) return launch_row<bf16_t, true>(a, dtype, grid_y, st);
  return launch_row<float, true>(a, dtype, grid_y, st);
}

}  // namespace seg
