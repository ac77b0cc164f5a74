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
#include "p2p.h"

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

// Tiled version (C % 4 == 0, O % 8 == 0): blocks [0, tiles) scale one 64(o) x 64(c) tile each
// and emit it twice — row-major (Wp) straight from the registers, and transposed (WpT) through
// an LDS tile, 16 bytes of consecutive o per store (the wave-per-row kernel above writes WpT as
// 2-byte stores one row apart: 11 us for 728x728).  Blocks [tiles, tiles + ceil(O/4)) compute
// b'[o] = W[o,:] . t, one wave per row.
template <typename T>
__global__ __launch_bounds__(256) void fold_weights_tiled_kernel(
    const float* __restrict__ W, const float* __restrict__ s, const float* __restrict__ t,
    T* __restrict__ Wp, T* __restrict__ WpT, float* __restrict__ bprime, int O, int C,
    int tiles_c, int tiles) {
  constexpr int VEC = Vec<T>::N;
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= tiles) {  // ---- row dot products
    const int lane = tid & 63;
    const int o = ((int)blockIdx.x - tiles) * 4 + (tid >> 6);
    if (o >= O || bprime == nullptr) return;
    float acc = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const float4 w = *reinterpret_cast<const float4*>(W + (long)o * C + c);
      const float4 tt = *reinterpret_cast<const float4*>(t + c);
      acc = fmaf(w.x, tt.x, fmaf(w.y, tt.y, fmaf(w.z, tt.z, fmaf(w.w, tt.w, acc))));
    }
    acc = wave_sum(acc);
    if (lane == 0) bprime[o] = acc;
    return;
  }
  const int to = blockIdx.x / tiles_c, tc = blockIdx.x - to * tiles_c;
  const int o0 = to * 64, c0 = tc * 64;
  const int cq = tid & 15, r = tid >> 4;
  const int c = c0 + cq * 4;
  float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) sv = *reinterpret_cast<const float4*>(s + c);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ro = r + 16 * i, o = o0 + ro;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o < O && c < C) w = *reinterpret_cast<const float4*>(W + (long)o * C + c);
    const float f[4] = {w.x * sv.x, w.y * sv.y, w.z * sv.z, w.w * sv.w};
    if (o < O && c < C) HVec<T>::store(Wp + (long)o * C + c, f);
    tile[ro][cq * 4 + 0] = f[0];
    tile[ro][cq * 4 + 1] = f[1];
    tile[ro][cq * 4 + 2] = f[2];
    tile[ro][cq * 4 + 3] = f[3];
  }
  if (WpT == nullptr) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    const int og = idx & 7, cl = idx >> 3;  // 8 consecutive o of column cl
    const int o = o0 + og * 8, cc = c0 + cl;
    if (cc < C && o < O) {  // (O % 8 == 0: the 8 rows exist together)
      float f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = tile[og * 8 + k][cl];
      T* dst = WpT + (long)cc * O + o;
#pragma unroll
      for (int k = 0; k < 8; k += VEC) Vec<T>::store(dst + k, f + k);
    }
  }
}

// dWp arrives as the weight-gradient GEMM's split partials [S][O*C] (summed here in a fixed
// order: the separate column-sum pass and its 2x2 MB round trip are gone).  Lanes run along c
// (256 contiguous bytes per row and wave), a block owns 64 columns x one of RS row ranges and
// emits one row of (ds, dt) partials: dsdt[rs][2][C], summed by fold_bwd_finalize.
constexpr int FOLD_RS = 32;
__global__ __launch_bounds__(256) void fold_bwd_reduce_kernel(
    const float* __restrict__ W, const float* __restrict__ dWp, int S, const float* __restrict__ s,
    const float* __restrict__ t, const float* __restrict__ db, float* __restrict__ dW,
    float* __restrict__ dsdt, int O, int C) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int rows_per = (O + gridDim.y - 1) / gridDim.y;
  const int o0 = blockIdx.y * rows_per, o1 = min(O, o0 + rows_per);
  const long OC = (long)O * C;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    const float sc = s[c], tc = t[c];
    for (int o = o0 + wave; o < o1; o += 4) {
      const long idx = (long)o * C + c;
      // the splits are summed in a fixed order, eight independent loads in flight at a time
      float g = 0.f;
      int k = 0;
      for (; k + 8 <= S; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = dWp[(long)(k + u) * OC + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) g += v[u];
      }
      for (; k < S; ++k) g += dWp[(long)k * OC + idx];
      const float w = W[idx];
      const float dbo = db ? db[o] : 0.f;
      dW[idx] = fmaf(g, sc, dbo * tc);
      a0 = fmaf(w, g, a0);
      a1 = fmaf(w, dbo, a1);
    }
  }
  red[0][wave][lane] = a0;
  red[1][wave][lane] = a1;
  __syncthreads();
  if (wave == 0 && c < C) {
    float* dst = dsdt + (long)blockIdx.y * 2 * C;
    dst[c] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    dst[C + c] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
  }
}

// Vectorised version for C % 4 == 0 (every real layer): a lane owns FOUR consecutive columns
// (16-byte loads: 1 KiB per wave instruction instead of 256 B), waves walk the rows of the
// block's row slice.  The first version moved 15 MB in 19 us (0.8 TB/s, dword loads).
constexpr int FOLD_RS4 = 64;
__global__ __launch_bounds__(256) void fold_bwd_reduce4_kernel(
    const float* __restrict__ W, const float* __restrict__ dWp, int S, const float* __restrict__ s,
    const float* __restrict__ t, const float* __restrict__ db, float* __restrict__ dW,
    float* __restrict__ dsdt, int O, int C) {
  __shared__ float4 red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const int rows_per = (O + gridDim.y - 1) / gridDim.y;
  const int o0 = blockIdx.y * rows_per, o1 = min(O, o0 + rows_per);
  const long OC = (long)O * C;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  if (c < C) {
    const float4 sc = *reinterpret_cast<const float4*>(s + c);
    const float4 tc = *reinterpret_cast<const float4*>(t + c);
    for (int o = o0 + wave; o < o1; o += 4) {
      const long idx = (long)o * C + c;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      // fixed order, eight UNCONDITIONAL loads in flight per round (rows past S re-read the
      // last split and are masked: a load behind a per-element branch would make hipcc wait
      // vmcnt(0) after each one)
      for (int k = 0; k < S; k += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = *reinterpret_cast<const float4*>(dWp + (long)min(k + u, S - 1) * OC + idx);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float m = (k + u < S) ? 1.f : 0.f;
          g.x = fmaf(m, v[u].x, g.x); g.y = fmaf(m, v[u].y, g.y);
          g.z = fmaf(m, v[u].z, g.z); g.w = fmaf(m, v[u].w, g.w);
        }
      }
      const float4 w = *reinterpret_cast<const float4*>(W + idx);
      const float dbo = db ? db[o] : 0.f;
      float4 out;
      out.x = fmaf(g.x, sc.x, dbo * tc.x); out.y = fmaf(g.y, sc.y, dbo * tc.y);
      out.z = fmaf(g.z, sc.z, dbo * tc.z); out.w = fmaf(g.w, sc.w, dbo * tc.w);
      *reinterpret_cast<float4*>(dW + idx) = out;
      a0.x = fmaf(w.x, g.x, a0.x); a0.y = fmaf(w.y, g.y, a0.y);
      a0.z = fmaf(w.z, g.z, a0.z); a0.w = fmaf(w.w, g.w, a0.w);
      a1.x = fmaf(w.x, dbo, a1.x); a1.y = fmaf(w.y, dbo, a1.y);
      a1.z = fmaf(w.z, dbo, a1.z); a1.w = fmaf(w.w, dbo, a1.w);
    }
  }
  red[0][wave][lane] = a0;
  red[1][wave][lane] = a1;
  __syncthreads();
  if (wave == 0 && c < C) {
    float* dst = dsdt + (long)blockIdx.y * 2 * C;
    float4 r0 = red[0][0][lane], r1 = red[1][0][lane];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float4 p = red[0][k][lane], q = red[1][k][lane];
      r0.x += p.x; r0.y += p.y; r0.z += p.z; r0.w += p.w;
      r1.x += q.x; r1.y += q.y; r1.z += q.z; r1.w += q.w;
    }
    *reinterpret_cast<float4*>(dst + c) = r0;
    *reinterpret_cast<float4*>(dst + C + c) = r1;
  }
}

// block = 64 channels x 4 row groups; every per-channel parameter is requested before the row
// loop (the kernel is pure latency: was one thread per channel walking all 64 rows, 5.7 us)
__global__ __launch_bounds__(256) void fold_bwd_finalize_kernel(
    const float* __restrict__ dsdt, int R, double count, const double* __restrict__ count_dev,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ scale, float* dgamma, float* dbeta,
    float* c0, float* c1, int C, double grad_scale, const P2PDev p2p) {
  __shared__ double red[2][4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool ok = c < C, fin = ok && rg == 0;
  float muf = 0.f, isf = 1.f, gf = 1.f, scf = 0.f;
  if (fin) {
    muf = mean[c];
    isf = invstd[c];
    if (gamma) gf = gamma[c];
    scf = scale[c];
    if (count_dev) count = *count_dev;
  }
  double ds = 0.0, dt = 0.0;
  if (ok) {
    for (int r0 = rg; r0 < R; r0 += 32) {  // eight independent row loads in flight
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = min(r0 + 4 * u, R - 1);
        a[u] = dsdt[(long)r * 2 * C + c];
        b[u] = dsdt[(long)r * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (r0 + 4 * u < R) { ds += (double)a[u]; dt += (double)b[u]; }
      }
    }
  }
  red[0][rg][cl] = ds;
  red[1][rg][cl] = dt;
  __syncthreads();
  if (rg == 0) {
    ds = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    dt = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  }
  if (p2p.world) {  // SyncBatchNorm: (ds, dt) summed over the ranks inside this kernel (p2p.h)
    double v[2] = {ds, dt};
    p2p_block_exchange<2>(p2p, blockIdx.x, gridDim.x, rg == 0, cl, 64, v);
    ds = v[0]; dt = v[1];
  }
  if (!fin) return;
  const double mu = muf, is = isf;
  const double g = gf;
  const double u = ds - mu * dt;
  const double A = g * u * is * is * is / count;
  if (dgamma) dgamma[c] = (float)(is * u * grad_scale);
  if (dbeta) dbeta[c] = (float)(dt * grad_scale);
  c1[c] = (float)A;
  c0[c] = (float)(dt * (double)scf / count - A * mu);
}

}  // namespace seg

extern "C" int seg_fold_weights(int dtype, const float* W, const float* scale, const float* shift,
                                void* Wp, void* WpT, float* bprime, int O, int C, void* stream) {
  using namespace seg;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "fold_weights: bad dtype %d", dtype);
  SEG_REQUIRE(O >= 1 && C >= 1 && W && scale && shift && Wp, "fold_weights: bad arguments");
  if (C % 4 == 0 && O % 8 == 0) {
    const int tiles_c = (C + 63) / 64, tiles = tiles_c * ((O + 63) / 64);
    const dim3 grid(tiles + (bprime ? (O + 3) / 4 : 0));
    if (dtype == DT_BF16)
      hipLaunchKernelGGL((fold_weights_tiled_kernel<bf16_t>), grid, dim3(256), 0,
                         (hipStream_t)stream, W, scale, shift, (bf16_t*)Wp, (bf16_t*)WpT, bprime,
                         O, C, tiles_c, tiles);
    else
      hipLaunchKernelGGL((fold_weights_tiled_kernel<float>), grid, dim3(256), 0,
                         (hipStream_t)stream, W, scale, shift, (float*)Wp, (float*)WpT, bprime, O,
                         C, tiles_c, tiles);
    return check_launch("fold_weights");
  }
  const dim3 grid((O + 3) / 4);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((fold_weights_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, W,
                       scale, shift, (bf16_t*)Wp, (bf16_t*)WpT, bprime, O, C);
  else
    hipLaunchKernelGGL((fold_weights_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, W,
                       scale, shift, (float*)Wp, (float*)WpT, bprime, O, C);
  return check_launch("fold_weights");
}

extern "C" int seg_fold_bwd_rows(int O) {
  return O < 4 * seg::FOLD_RS ? 1 : seg::FOLD_RS4;
}

extern "C" int seg_fold_bwd_reduce(const float* W, const float* dWp, int splits, const float* scale,
                                   const float* shift, const float* db, float* dW, float* dsdt,
                                   int O, int C, void* stream) {
  using namespace seg;
  SEG_REQUIRE(O >= 1 && C >= 1 && splits >= 1, "fold_bwd_reduce: empty");
  if (C % 4 == 0) {
    hipLaunchKernelGGL(fold_bwd_reduce4_kernel, dim3((C / 4 + 63) / 64, seg_fold_bwd_rows(O)),
                       dim3(256), 0, (hipStream_t)stream, W, dWp, splits, scale, shift, db, dW,
                       dsdt, O, C);
    return check_launch("fold_bwd_reduce");
  }
  hipLaunchKernelGGL(fold_bwd_reduce_kernel, dim3((C + 63) / 64, seg_fold_bwd_rows(O)), dim3(256),
                     0, (hipStream_t)stream, W, dWp, splits, scale, shift, db, dW, dsdt, O, C);
  return check_launch("fold_bwd_reduce");
}

static int fold_bwd_finalize_impl(const float* dsdt, int rows, double count,
                                  const double* count_dev, const float* mean, const float* invstd,
                                  const float* gamma, const float* scale, float* dgamma,
                                  float* dbeta, float* c0, float* c1, int C, double grad_scale,
                                  void* stream, const seg::P2PDev& p2p = seg::p2p_dev_none()) {
  using namespace seg;
  SEG_REQUIRE((count_dev || count >= 1.0) && C >= 1 && rows >= 1,
              "fold_bwd_finalize: bad count/C/rows");
  hipLaunchKernelGGL(fold_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0,
                     (hipStream_t)stream, dsdt, rows, count, count_dev, mean, invstd, gamma, scale,
                     dgamma, dbeta, c0, c1, C, grad_scale, p2p);
  return check_launch("fold_bwd_finalize");
}

// SyncBatchNorm: (ds, dt) exchanged between the ranks inside the kernel (p2p.h) — dsdt holds this
// rank's LOCAL partial rows; `p2p` = handle of seg_p2p_create
extern "C" int seg_fold_bwd_finalize_sync(void* p2p, const float* dsdt, int rows,
                                          const double* count_dev, const float* mean,
                                          const float* invstd, const float* gamma,
                                          const float* scale, float* dgamma, float* dbeta,
                                          float* c0, float* c1, int C, double grad_scale,
                                          void* stream) {
  using namespace seg;
  SEG_REQUIRE(count_dev != nullptr, "fold_bwd_finalize_sync: the global count lives on the device");
  P2PDev d;
  if (!p2p_dev_of(p2p, d)) return 2;
  const int nb = (C + 63) / 64;
  SEG_REQUIRE(nb <= P2P_MAX_BLOCKS && (long)nb * 64 * 2 * 8 <= d.slot_bytes,
              "fold_bwd_finalize_sync: C=%d exceeds the mailbox", C);
  return fold_bwd_finalize_impl(dsdt, rows, 1.0, count_dev, mean, invstd, gamma, scale, dgamma,
                                dbeta, c0, c1, C, grad_scale, stream, d);
}

extern "C" int seg_fold_bwd_finalize(const float* dsdt, int rows, double count,
                                     const double* count_dev, const float* mean,
                                     const float* invstd, const float* gamma, const float* scale,
                                     float* dgamma, float* dbeta, float* c0, float* c1, int C,
                                     void* stream) {
  return fold_bwd_finalize_impl(dsdt, rows, count, count_dev, mean, invstd, gamma, scale, dgamma,
                                dbeta, c0, c1, C, 1.0, stream);
}

// the same with dgamma / dbeta multiplied by grad_scale (SyncBatchNorm: 1 / world size)
extern "C" int seg_fold_bwd_finalize_s(const float* dsdt, int rows, double count,
                                       const double* count_dev, const float* mean,
                                       const float* invstd, const float* gamma,
                                       const float* scale, float* dgamma, float* dbeta, float* c0,
                                       float* c1, int C, double grad_scale, void* stream) {
  return fold_bwd_finalize_impl(dsdt, rows, count, count_dev, mean, invstd, gamma, scale, dgamma,
                                dbeta, c0, c1, C, grad_scale, stream);
}
