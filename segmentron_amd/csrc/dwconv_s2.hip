// Fused backward of the STRIDE-2 depthwise 3x3 convolution (padding 1, dilation 1: the last
// separable conv of the Xception entry-flow blocks, xception.py:33-35, and MobileNetV2's strided
// inverted residuals): ONE pass over dy and x produces
//   * g   = mask(x) * conv_transpose(dy, w)      (masked by the prologue's ReLU / ReLU6)
//   * the weight-gradient partials                [grid_y][9][C]
//   * the BatchNorm-backward sums of the input    [grid_y][2][C]  (sum g, sum g*x_raw)
// — what the strip kernels did in three launches (dgrad 176 us + wgrad 195 us + bn_bwd_reduce
// 103 us on the 268 MB block1 tensor, each re-reading x or g).
//
// Geometry: y[ho][wo] = sum x[2ho-1+kh][2wo-1+kw] w[kh][kw], so input pixel (h, w) meets tap
// (kh, kw) only where h+1-kh and w+1-kw are even: an even row sees kh = 1 (ho = h/2), an odd row
// kh = 0 (ho = (h+1)/2) and kh = 2 (ho = (h-1)/2); same for columns.  A block owns an 8 x 16
// INPUT tile (h0 % 8 == 0, w0 % 16 == 0) of 32 channels: the 5 x 9 dy pixels it needs go to LDS
// as fp32; thread = (4 channels, row, strip of 4 columns).  Row parity is a property of the
// THREAD (tiles start on even rows), column parity of the unrolled strip position, so the taps a
// thread touches are fixed for the whole kernel: two kh "slots" x three kw weight vectors and tap
// accumulators live in registers; the slot an even row does not have carries zero weights and a
// zero dy multiplier (uniform control flow, no divergence).  Persistent blocks over tiles; one
// deterministic block reduction at the end (rows of equal parity by lane exchange, strips in LDS).
#include "common.h"

namespace seg {

constexpr int S2_TH = 8, S2_TW = 16, S2_CVB = 8, S2_THREADS = 256;
constexpr int S2_DH = S2_TH / 2 + 1, S2_DW = S2_TW / 2 + 1;  // dy tile: 5 x 9 pixels

struct DwS2Args {
  const void* x; const void* dy; void* g;
  const float* w;  // torch's [C][9]
  const float* sc; const float* sh;
  float* partial_w;   // [grid_y][9][C]
  float* partial_bn;  // [grid_y][2][C] or null
  long ldx, lddy, ldg;
  int N, H, W, Ho, Wo, C, CV, pro_mode, tiles_h, tiles_w, ntiles;
};

template <typename T>
__global__ __launch_bounds__(S2_THREADS, 3) void dwconv_bwd_s2_kernel(const DwS2Args a) {
  using V = HVec<T>;
  constexpr int VEC = 4;
  __shared__ float4 dyt[S2_DH * S2_DW * S2_CVB];         // [pixel][cx]
  __shared__ float red[4 * S2_CVB * 11 * VEC];           // [strip][cx][9 taps + 2 sums][4]
  const int tid = threadIdx.x;
  // XCD-local logical block ids, channel block fastest (the blocks sharing cache lines of a
  // pixel sit on one XCD)
  const int flat = blockIdx.x + gridDim.x * blockIdx.y;
  const int L = xcd_remap(flat, gridDim.x * gridDim.y);
  const int by = L / (int)gridDim.x, bx = L - by * (int)gridDim.x;
  const int cx = tid & (S2_CVB - 1), row = (tid >> 3) & (S2_TH - 1), strip = tid >> 6;
  const int cv = bx * S2_CVB + cx;
  const bool cok = cv < a.CV;
  const int c0 = (cok ? cv : 0) * VEC;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ DY = reinterpret_cast<const T*>(a.dy);
  T* __restrict__ G = reinterpret_cast<T*>(a.g);

  // ---- this thread's taps: slot s -> kernel row, dy row (relative to the tile's first dy row)
  const int ph = row & 1;
  const int kh_s[2] = {ph ? 0 : 1, 2};
  const int ry_s[2] = {ph ? (row + 1) >> 1 : row >> 1, ph ? (row - 1) >> 1 : 0};
  const float vs[2] = {1.f, ph ? 1.f : 0.f};
  float wv[2][3][VEC];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        wv[s][kw][i] = (cok && vs[s] != 0.f) ? a.w[(long)(c0 + i) * 9 + kh_s[s] * 3 + kw] : 0.f;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
  if (a.pro_mode & PRO_AFFINE) {
    load_params<VEC>(a.sc, c0, sc);
    load_params<VEC>(a.sh, c0, sh);
  }
  float accw[2][3][VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int i = 0; i < VEC; ++i) accw[s][kw][i] = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;

  for (int t = by; t < a.ntiles; t += gridDim.y) {
    const int tw = t % a.tiles_w, tq = t / a.tiles_w;
    const int th = tq % a.tiles_h, n = tq / a.tiles_h;
    const int h0 = th * S2_TH, w0 = tw * S2_TW;
    const int ho0 = h0 >> 1, wo0 = w0 >> 1;
    // ---- x of this thread's four pixels (issued first: consumed after the dy tile is staged)
    const int h = h0 + row;
    typename V::raw_t xraw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int w = w0 + strip * 4 + j;
      xraw[j] = V::load_raw(X + (((long)n * a.H + min(h, a.H - 1)) * a.W + min(w, a.W - 1)) * a.ldx + c0);
    }
    // ---- dy tile -> LDS (fp32), zero outside the output image
    __syncthreads();  // the previous tile's readers are done
    for (int idx = tid; idx < S2_DH * S2_DW * S2_CVB; idx += S2_THREADS) {
      const int lcx = idx & (S2_CVB - 1), pix = idx >> 3;
      const int r = pix / S2_DW, c = pix - r * S2_DW;
      const int ho = ho0 + r, wo = wo0 + c;
      const int ccv = bx * S2_CVB + lcx;
      const bool ok = ho < a.Ho && wo < a.Wo && ccv < a.CV;
      float f[VEC];
      V::load(DY + (ok ? (((long)n * a.Ho + ho) * a.Wo + wo) * a.lddy + (long)ccv * VEC : 0), f);
      dyt[idx] = ok ? make_float4(f[0], f[1], f[2], f[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const bool rok = cok && h < a.H;
    float xr[4][VEC], xa[4][VEC], g[4][VEC];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      V::unpack_raw(xraw[j], xr[j]);
      const bool ok = rok && (w0 + strip * 4 + j) < a.W;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float v = xr[j][i];
        if (a.pro_mode & PRO_AFFINE) v = fmaf(v, sc[i], sh[i]);
        if (a.pro_mode & PRO_RELU) v = fmaxf(v, 0.f);
        if (a.pro_mode & PRO_CLAMP6) v = fminf(v, 6.f);
        xa[j][i] = ok ? v : 0.f;
        xr[j][i] = ok ? xr[j][i] : 0.f;
        g[j][i] = 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float4* drow = dyt + (ry_s[s] * S2_DW + strip * 2) * S2_CVB + cx;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // column j of the strip: even -> kw = 1 at dy column j/2; odd -> kw = 0 at (j+1)/2 and
        // kw = 2 at (j-1)/2   (strip * 4 is even, so the parity of j is the pixel's)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          if (((j + 1 - kw) & 1) != 0) continue;
          const float4 d4 = drow[((j + 1 - kw) >> 1) * S2_CVB];
          const float d[VEC] = {d4.x * vs[s], d4.y * vs[s], d4.z * vs[s], d4.w * vs[s]};
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            g[j][i] = fmaf(d[i], wv[s][kw][i], g[j][i]);
            accw[s][kw][i] = fmaf(d[i], xa[j][i], accw[s][kw][i]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int w = w0 + strip * 4 + j;
      if (rok && w < a.W) {
        if (a.pro_mode & PRO_RELU) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const bool on = xa[j][i] > 0.f && (!(a.pro_mode & PRO_CLAMP6) || xa[j][i] < 6.f);
            g[j][i] = on ? g[j][i] : 0.f;
          }
        }
        V::store(G + (((long)n * a.H + h) * a.W + w) * a.ldg + c0, g[j]);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          s1[i] += g[j][i];
          s2[i] = fmaf(g[j][i], xr[j][i], s2[i]);
        }
      }
    }
  }

  // ---- block reduction.  Lanes of a wave: cx = bits 0-2, row = bits 3-5; rows of equal parity
  // (same taps) differ in bits 4-5; the statistics sum over all rows.
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float v = accw[s][kw][i];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        accw[s][kw][i] = v;
      }
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    s1[i] += __shfl_xor(s1[i], 8, 64);  s1[i] += __shfl_xor(s1[i], 16, 64);
    s1[i] += __shfl_xor(s1[i], 32, 64);
    s2[i] += __shfl_xor(s2[i], 8, 64);  s2[i] += __shfl_xor(s2[i], 16, 64);
    s2[i] += __shfl_xor(s2[i], 32, 64);
  }
  __syncthreads();
  float* mine = red + (strip * S2_CVB + cx) * 11 * VEC;
  if (row == 0) {  // even rows: kernel row 1; also the statistics
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int i = 0; i < VEC; ++i) mine[(3 + kw) * VEC + i] = accw[0][kw][i];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      mine[9 * VEC + i] = s1[i];
      mine[10 * VEC + i] = s2[i];
    }
  } else if (row == 1) {  // odd rows: kernel rows 0 (slot 0) and 2 (slot 1)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        mine[kw * VEC + i] = accw[0][kw][i];
        mine[(6 + kw) * VEC + i] = accw[1][kw][i];
      }
  }
  __syncthreads();
  for (int e = tid; e < S2_CVB * 11 * VEC; e += S2_THREADS) {
    float tot = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) tot += red[s * S2_CVB * 11 * VEC + e];
    const int lcx = e / (11 * VEC), k = e - lcx * 11 * VEC;
    const int r = k / VEC, ci = k - r * VEC;
    const int c = (bx * S2_CVB + lcx) * VEC + ci;
    if (c < a.C) {
      if (r < 9) a.partial_w[((long)by * 9 + r) * a.C + c] = tot;
      else if (a.partial_bn != nullptr) a.partial_bn[((long)by * 2 + (r - 9)) * a.C + c] = tot;
    }
  }
}

static int s2_grid_y(int C, int N, int H, int W) {
  const int gx = (C / 4 + S2_CVB - 1) / S2_CVB;
  const long ntiles = (long)N * ((H + S2_TH - 1) / S2_TH) * ((W + S2_TW - 1) / S2_TW);
  long cap = 768 / gx;  // one resident set: 3 blocks per CU
  if (cap < 1) cap = 1;
  return (int)(ntiles < cap ? ntiles : cap);
}

}  // namespace seg

// rows of the partial buffers seg_dwconv3x3_s2_bwd_fused writes (x is [N, H, W, C])
extern "C" int seg_dwconv3x3_s2_grid_y(int C, int N, int H, int W) {
  return seg::s2_grid_y(C, N, H, W);
}

// x [N,H,W,C] (+ prologue), dy [N,Ho,Wo,C] with Ho = (H+1)/2, Wo = (W+1)/2 (stride 2, pad 1,
// dil 1), w: torch's [C,1,3,3] fp32.  g [N,H,W,C] = masked data gradient; partial_w
// [grid_y][9][C]; partial_bn [grid_y][2][C] (nullable) = (sum g, sum g*x_raw).
extern "C" int seg_dwconv3x3_s2_bwd_fused(int dtype, const void* dy, long lddy, const void* x,
                                          long ldx, int N, int H, int W, int C, const float* w_c9,
                                          int pro_mode, const float* pro_scale,
                                          const float* pro_shift, void* g, long ldg,
                                          float* partial_w, float* partial_bn, int grid_y,
                                          void* stream) {
  using namespace seg;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "dwconv3x3_s2_bwd_fused: bad dtype %d", dtype);
  SEG_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && ldg % 4 == 0,
              "dwconv3x3_s2_bwd_fused: C / ld must be multiples of 4");
  SEG_REQUIRE(((pro_mode & PRO_AFFINE) == 0) || (pro_scale && pro_shift),
              "dwconv3x3_s2_bwd_fused: affine prologue without scale/shift");
  SEG_REQUIRE(N >= 1 && H >= 1 && W >= 1 && partial_w && w_c9 && g,
              "dwconv3x3_s2_bwd_fused: bad arguments");
  SEG_REQUIRE(grid_y == s2_grid_y(C, N, H, W), "dwconv3x3_s2_bwd_fused: grid_y %d != %d", grid_y,
              s2_grid_y(C, N, H, W));
  DwS2Args a;
  a.x = x; a.dy = dy; a.g = g; a.w = w_c9; a.sc = pro_scale; a.sh = pro_shift;
  a.partial_w = partial_w; a.partial_bn = partial_bn;
  a.ldx = ldx; a.lddy = lddy; a.ldg = ldg;
  a.N = N; a.H = H; a.W = W; a.Ho = (H + 1) / 2; a.Wo = (W + 1) / 2; a.C = C; a.CV = C / 4;
  a.pro_mode = pro_mode;
  a.tiles_h = (H + S2_TH - 1) / S2_TH; a.tiles_w = (W + S2_TW - 1) / S2_TW;
  a.ntiles = N * a.tiles_h * a.tiles_w;
  const dim3 grid((a.CV + S2_CVB - 1) / S2_CVB, grid_y);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((dwconv_bwd_s2_kernel<bf16_t>), grid, dim3(S2_THREADS), 0,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((dwconv_bwd_s2_kernel<float>), grid, dim3(S2_THREADS), 0,
                       (hipStream_t)stream, a);
  return check_launch("dwconv3x3_s2_bwd_fused");
}
