// BatchNorm2d (train / eval / sync) as a set of small bandwidth-bound kernels around the fused
// conv kernels.  Replaces nn.BatchNorm2d / nn.SyncBatchNorm forward + backward
// (reference call sites: every `bn*` module of segmentron/modules/basic.py:34-77,
// segmentron/modules/module.py:32-77, segmentron/models/backbones/xception.py:10-165;
// semantics: SURVEY.md Appendix B).
//
// Forward (train):  conv epilogues emit per-tile (sum, sumsq) partials  ->  seg_colsum (fp64)
//                   -> [RCCL all-reduce of 2C doubles when SyncBN]  -> seg_bn_finalize
//                   -> (mean, invstd, scale = gamma*invstd, shift = beta - mean*scale), running
//                   stats updated with the UNBIASED variance, exactly as torch does.
//                   The normalisation itself is applied inside the CONSUMER kernels' prologue.
// Backward:         g' = g * relu_mask ;  seg_bn_bwd_reduce -> (sum g', sum g'*x) partials
//                   -> seg_colsum -> [all-reduce] -> seg_bn_bwd_finalize -> (dgamma, dbeta, c0, c1)
//                   -> seg_bn_bwd_apply:  dx = scale*g' - c0 - c1*x
// Wavefront (64-lane) layout: lanes run along 16-byte channel vectors of NHWC rows.
#include "common.h"
#include "p2p.h"

namespace seg {

constexpr int EW_THREADS = 256;
constexpr int EW_MAX_THREADS = 512;  // the row-tile kernels: rows x channel vectors of a block (ew_geom)
constexpr int EW_CAP = 512;  // blocks per launch (grid-stride beyond): 512 vs 8192 measured 13.0 vs 13.9 us on the 24 MB tensors
constexpr int EW_UN = 4;  // rows per thread per iteration in the row-tile kernels

// ------------------------------------------------------------------ column sums of partials
// in [R][L] fp32 -> out[gridDim.y][L] (fp64 or fp32)
// `tail` (nullable): out[L] = *tail is written by the first block — the SyncBatchNorm message is
// [column sums | local element count] (parallel.allreduce_forward_sums), assembled here instead
// of by a copy and a fill launch.
template <typename TOUT>
__global__ __launch_bounds__(EW_THREADS) void colsum_kernel(const float* __restrict__ in, long R,
                                                            int L, TOUT* __restrict__ out,
                                                            double tail = 0.0, int has_tail = 0) {
  if (has_tail && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) out[L] = (TOUT)tail;
  __shared__ double red[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cx;
  const long rows_per = (R + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * rows_per;
  const long r1 = min(R, r0 + rows_per);
  double acc = 0.0;
  if (col < L) {
    for (long r = r0 + ry; r < r1; r += 8 * 8) {  // (eight loads in flight, the old order)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = in[min(r + 8 * u, r1 - 1) * L + col];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r + 8 * u < r1) acc += (double)v[u];
    }
  }
  red[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && col < L) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][cx];
    out[(long)blockIdx.y * L + col] = (TOUT)t;
  }
}

__global__ __launch_bounds__(EW_THREADS) void colsum_f64_kernel(const double* __restrict__ in,
                                                                int R, int L, double* out_d,
                                                                float* out_f, double tail = 0.0,
                                                                int has_tail = 0) {
  if (has_tail && blockIdx.x == 0 && threadIdx.x == 0 && out_d) out_d[L] = tail;
  const int col = blockIdx.x * EW_THREADS + threadIdx.x;
  if (col >= L) return;
  double t = 0.0;
  for (int r0 = 0; r0 < R; r0 += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = in[(long)min(r0 + u, R - 1) * L + col];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (r0 + u < R) t += v[u];
  }
  if (out_d) out_d[col] = t;
  if (out_f) out_f[col] = (float)t;
}

// ------------------------------------------------------------------ finalize (forward)
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count,
                                   const double* __restrict__ count_dev,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, float* running_mean,
                                   float* running_var, float* mean_o, float* invstd_o,
                                   float* scale_o, float* shift_o, int C,
                                   const float* __restrict__ mean_offset) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (count_dev) count = *count_dev;  // SyncBN: the all-reduced element count
  const double mean = sums[c] / count;
  double var = sums[C + c] / count - mean * mean;  // biased
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean_o[c] = (float)mean;
  invstd_o[c] = (float)invstd;
  const float sc = (float)((double)g * invstd);
  scale_o[c] = sc;
  shift_o[c] = (float)((double)b - mean * (double)g * invstd);
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    const double off = mean_offset ? (double)mean_offset[c] : 0.0;
    running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * (mean + off));
    running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
  }
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma,
                                      const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv,
                                      float eps, float* scale_o, float* shift_o, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float invstd = 1.0f / sqrtf(rv[c] + eps);
  scale_o[c] = g * invstd;
  shift_o[c] = b - rm[c] * g * invstd;
}

// The same for up to 48 BatchNorms in ONE launch (r05): an evaluation-mode forward asked for
// one 3 us launch per BatchNorm — 52 of the 132 kernels of a DeepLabv3+/MobileNetV2 inference
// step and 12 % of its 1.39 ms (`profiles/r05_bench_c2.json`).  Block b serves job b / 8,
// channels (b % 8) * 256 .. (C <= 2048 per job; wider layers go through the single-job entry).
constexpr int EVAL_MULTI_MAX = 48, EVAL_MULTI_BPJ = 8;
struct EvalMultiArgs {
  const float* gamma[EVAL_MULTI_MAX];
  const float* beta[EVAL_MULTI_MAX];
  const float* rm[EVAL_MULTI_MAX];
  const float* rv[EVAL_MULTI_MAX];
  float* out[EVAL_MULTI_MAX];  // [2][C]: scale row, shift row
  float eps[EVAL_MULTI_MAX];
  int C[EVAL_MULTI_MAX];
};
__global__ void bn_eval_affine_multi_kernel(const EvalMultiArgs a) {
  const int j = blockIdx.x / EVAL_MULTI_BPJ;
  const int c = (blockIdx.x % EVAL_MULTI_BPJ) * 256 + threadIdx.x;
  const int C = a.C[j];
  if (c >= C) return;
  const float g = a.gamma[j] ? a.gamma[j][c] : 1.f, b = a.beta[j] ? a.beta[j][c] : 0.f;
  const float invstd = 1.0f / sqrtf(a.rv[j][c] + a.eps[j]);
  a.out[j][c] = g * invstd;
  a.out[j][C + c] = b - a.rm[j][c] * g * invstd;
}

// ------------------------------------------------------------------ apply (+ residual)
struct ApplyArgs {
  const void* x; const void* r; void* y;
  const float* sx; const float* tx; const float* sr; const float* tr;
  const float* chan_mul;  // optional [N][C] multiplier (Dropout2d mask * 1/(1-p)); rows_per_n rows each
  const void* elem_mul;   // optional per-element multiplier of type T, pitch ldm (nn.Dropout mask/(1-p))
  long ldm;
  long ldx, ldr, ldy;
  long M; int C, CV;
  int mode_x, mode_r, post_relu;
  long rows_per_n;
  int lpr, rpb;  // lanes (channel vectors) per row of a block, rows per block (ew_geom)
};

// Row-tile mapping shared by the element-wise kernels: a thread owns ONE 16-byte channel vector
// (so its BatchNorm parameters live in registers for the whole kernel) and walks rows with a
// grid stride; lanes run along the channel vectors of a row (coalesced NHWC), no divisions.
template <int VEC>
__device__ __forceinline__ void load_affine(int mode, const float* __restrict__ s,
                                            const float* __restrict__ t, int c0, float* sc,
                                            float* sh) {
  if (mode & PRO_AFFINE) {
    load_params<VEC>(s, c0, sc);
    load_params<VEC>(t, c0, sh);
  } else {
#pragma unroll
    for (int k = 0; k < VEC; ++k) { sc[k] = 1.f; sh[k] = 0.f; }
  }
}
template <int VEC>
__device__ __forceinline__ void act_regs(float* f, int mode, const float* sc, const float* sh) {
  if (mode & PRO_AFFINE) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) f[k] = fmaf(f[k], sc[k], sh[k]);
  }
  if (mode & PRO_RELU) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) f[k] = fmaxf(f[k], 0.f);
  }
  if (mode & PRO_CLAMP6) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) f[k] = fminf(f[k], 6.f);
  }
}

template <typename T>
__global__ __launch_bounds__(EW_MAX_THREADS) void bn_apply_kernel(const ApplyArgs a) {
  constexpr int VEC = Vec<T>::N;
  const int cvb = a.lpr, rpb = a.rpb;
  const int sy = threadIdx.x / cvb, cx = threadIdx.x - sy * cvb;
  const int cv = blockIdx.x * cvb + cx;
  if (cv >= a.CV || sy >= rpb) return;
  const int c0 = cv * VEC;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ R = reinterpret_cast<const T*>(a.r);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  float sx[VEC], tx[VEC], sr[VEC], tr[VEC];
  load_affine<VEC>(a.mode_x, a.sx, a.tx, c0, sx, tx);
  load_affine<VEC>(R ? a.mode_r : 0, a.sr, a.tr, c0, sr, tr);
  // EW_UN rows per thread and iteration, every load of the batch issued before the first use
  // (one 16-byte load in flight per thread left these kernels latency-bound at 2.5 TB/s on the
  // 24 MB middle-flow tensors; a plain copy of the same tensor runs at 6 TB/s out of MALL)
  const int M = (int)a.M, tile = rpb * EW_UN, step = gridDim.y * tile;
  const T* __restrict__ EM = reinterpret_cast<const T*>(a.elem_mul);
  for (int base = blockIdx.y * tile + sy; base < M; base += step) {
    uint4 rx[EW_UN], rr[EW_UN], rm[EW_UN];
#pragma unroll
    for (int u = 0; u < EW_UN; ++u) {
      const int row = min(base + u * rpb, M - 1);
      rx[u] = ldg16(X + (long)row * a.ldx + c0);
      if (R) rr[u] = ldg16(R + (long)row * a.ldr + c0);
      if (EM) rm[u] = ldg16(EM + (long)row * a.ldm + c0);
    }
#pragma unroll
    for (int u = 0; u < EW_UN; ++u) {
      const int row = base + u * rpb;
      if (row >= M) break;
      float f[VEC];
      Vec<T>::unpack(rx[u], f);
      act_regs<VEC>(f, a.mode_x, sx, tx);
      if (a.chan_mul) {
        float m[VEC];
        load_params<VEC>(a.chan_mul + (long)(row / (int)a.rows_per_n) * a.C, c0, m);
#pragma unroll
        for (int k = 0; k < VEC; ++k) f[k] *= m[k];
      }
      if (EM) {
        float m[VEC];
        Vec<T>::unpack(rm[u], m);
#pragma unroll
        for (int k = 0; k < VEC; ++k) f[k] *= m[k];
      }
      if (R) {
        float g[VEC];
        Vec<T>::unpack(rr[u], g);
        act_regs<VEC>(g, a.mode_r, sr, tr);
#pragma unroll
        for (int k = 0; k < VEC; ++k) f[k] += g[k];
      }
      if (a.post_relu) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) f[k] = fmaxf(f[k], 0.f);
      }
      stg16(Y + (long)row * a.ldy + c0, Vec<T>::pack(f));
    }
  }
}

// ------------------------------------------------------------------ backward reduce
struct BwdArgs {
  const void* g; const void* x; void* dx;
  const float* scale; const float* shift; const float* c0; const float* c1;
  const float* chan_mul; long rows_per_n;
  const void* elem_mul; long ldm;
  float* partial;  // [gridDim.y][2][C]
  long ldg, ldx, lddx;
  long M; int C, CV;
  int mode;  // PRO_* of the forward prologue this is the backward of
  int lpr, rpb;  // lanes (channel vectors) per row of a block, rows per block (ew_geom)
};

template <typename T>
struct BwdRaw { uint4 g, x, m; };

template <typename T>
__device__ __forceinline__ void load_bwd_raw(const BwdArgs& a, const T* G, const T* X, int row,
                                             int c0, BwdRaw<T>& r) {
  r.g = ldg16(G + (long)row * a.ldg + c0);
  r.x = ldg16(X + (long)row * a.ldx + c0);
  if (a.elem_mul) r.m = ldg16(reinterpret_cast<const T*>(a.elem_mul) + (long)row * a.ldm + c0);
}

template <typename T>
__device__ __forceinline__ void masked_grad(const BwdArgs& a, const BwdRaw<T>& r, int row, int c0,
                                            const float* sc, const float* sh, float* g, float* x) {
  constexpr int VEC = Vec<T>::N;
  Vec<T>::unpack(r.g, g);
  Vec<T>::unpack(r.x, x);
  if (a.chan_mul) {
    float m[VEC];
    load_params<VEC>(a.chan_mul + (long)(row / (int)a.rows_per_n) * a.C, c0, m);
#pragma unroll
    for (int k = 0; k < VEC; ++k) g[k] *= m[k];
  }
  if (a.elem_mul) {
    float m[VEC];
    Vec<T>::unpack(r.m, m);
#pragma unroll
    for (int k = 0; k < VEC; ++k) g[k] *= m[k];
  }
  if (a.mode & PRO_RELU) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float y = (a.mode & PRO_AFFINE) ? fmaf(x[k], sc[k], sh[k]) : x[k];
      const bool on = y > 0.f && (!(a.mode & PRO_CLAMP6) || y < 6.f);  // ReLU / ReLU6 derivative
      g[k] = on ? g[k] : 0.f;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(EW_MAX_THREADS) void bn_bwd_reduce_kernel(const BwdArgs a) {
  constexpr int VEC = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) float bn_smem[];
  const int tid = threadIdx.x;
  const int cvb = a.lpr, spb = a.rpb;
  const int sy = tid / cvb, cx = tid - sy * cvb;
  const int cv = blockIdx.x * cvb + cx;
  const int c0 = cv * VEC;
  const T* __restrict__ G = reinterpret_cast<const T*>(a.g);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) s1[k] = s2[k] = 0.f;
  if (cv < a.CV && sy < spb) {
    float sc[VEC], sh[VEC];
    load_affine<VEC>(a.mode, a.scale, a.shift, c0, sc, sh);
    const int M = (int)a.M, tile = spb * EW_UN, step = gridDim.y * tile;
    for (int base = blockIdx.y * tile + sy; base < M; base += step) {
      BwdRaw<T> raw[EW_UN];
#pragma unroll
      for (int u = 0; u < EW_UN; ++u)
        load_bwd_raw<T>(a, G, X, min(base + u * spb, M - 1), c0, raw[u]);
#pragma unroll
      for (int u = 0; u < EW_UN; ++u) {
        const int row = base + u * spb;
        if (row >= M) break;
        float g[VEC], x[VEC];
        masked_grad<T>(a, raw[u], row, c0, sc, sh, g, x);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          s1[k] += g[k];
          s2[k] = fmaf(g[k], x[k], s2[k]);
        }
      }
    }
  }
  float* mine = bn_smem + ((long)sy * cvb + cx) * 2 * VEC;
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    mine[k] = s1[k];
    mine[VEC + k] = s2[k];
  }
  __syncthreads();
  for (int e = tid; e < cvb * 2 * VEC; e += (int)blockDim.x) {
    float tot = 0.f;
    for (int r = 0; r < spb; ++r) tot += bn_smem[(long)r * cvb * 2 * VEC + e];
    const int lcx = e / (2 * VEC), k = e % (2 * VEC);
    const int which = k / VEC, ci = k % VEC;
    const int c = (blockIdx.x * cvb + lcx) * VEC + ci;
    if (c < a.C) a.partial[((long)blockIdx.y * 2 + which) * a.C + c] = tot;
  }
}

// grad_scale: factor on the PARAMETER gradients only (SyncBatchNorm: the sums are global, and the
// data-parallel gradient averaging divides by the world size once more — 1 / world here keeps the
// averaged value equal to the reference's, parallel.local_param_grads)
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sums, double count,
                                       const double* __restrict__ count_dev,
                                       const float* __restrict__ mean,
                                       const float* __restrict__ invstd,
                                       const float* __restrict__ gamma, float* dgamma,
                                       float* dbeta, float* c0_o, float* c1_o, int C,
                                       double grad_scale = 1.0) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (count_dev) count = *count_dev;
  const double sg = sums[c], sgx = sums[C + c];
  const double mu = mean[c], is = invstd[c];
  const double dg = (sgx - mu * sg) * is;  // sum g' * xhat
  const double g = gamma ? (double)gamma[c] : 1.0;
  const double s = g * is;
  const double m1 = sg / count, m2 = dg / count;
  const double c1 = s * m2 * is;
  if (dgamma) dgamma[c] = (float)(dg * grad_scale);
  if (dbeta) dbeta[c] = (float)(sg * grad_scale);
  c1_o[c] = (float)c1;
  c0_o[c] = (float)(s * m1 - c1 * mu);
}

template <typename T>
__global__ __launch_bounds__(EW_MAX_THREADS) void bn_bwd_apply_kernel(const BwdArgs a) {
  constexpr int VEC = Vec<T>::N;
  const int cvb = a.lpr, rpb = a.rpb;
  const int sy = threadIdx.x / cvb, cx = threadIdx.x - sy * cvb;
  const int cv = blockIdx.x * cvb + cx;
  if (cv >= a.CV || sy >= rpb) return;
  const int c0 = cv * VEC;
  const T* __restrict__ G = reinterpret_cast<const T*>(a.g);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  T* __restrict__ DX = reinterpret_cast<T*>(a.dx);
  float sc[VEC], sh[VEC], k0[VEC], k1[VEC];
  load_affine<VEC>(a.mode, a.scale, a.shift, c0, sc, sh);
  if (a.c0) {
    load_params<VEC>(a.c0, c0, k0);
    load_params<VEC>(a.c1, c0, k1);
  } else {
#pragma unroll
    for (int k = 0; k < VEC; ++k) k0[k] = k1[k] = 0.f;
  }
  const int M = (int)a.M, tile = rpb * EW_UN, step = gridDim.y * tile;
  for (int base = blockIdx.y * tile + sy; base < M; base += step) {
    BwdRaw<T> raw[EW_UN];
#pragma unroll
    for (int u = 0; u < EW_UN; ++u)
      load_bwd_raw<T>(a, G, X, min(base + u * rpb, M - 1), c0, raw[u]);
#pragma unroll
    for (int u = 0; u < EW_UN; ++u) {
      const int row = base + u * rpb;
      if (row >= M) break;
      float g[VEC], x[VEC];
      masked_grad<T>(a, raw[u], row, c0, sc, sh, g, x);
      if (a.mode & PRO_AFFINE) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) g[k] = g[k] * sc[k] - k0[k] - k1[k] * x[k];
      }
      stg16(DX + (long)row * a.lddx + c0, Vec<T>::pack(g));
    }
  }
}

// ------------------------------------------------------------------ fused partials -> finalize
// One launch instead of colsum + colsum_f64 + finalize (non-sync BatchNorm): a block owns 32
// channels, its 8 row groups reduce the [R][2][C] partial rows in fp64 through LDS.
// block = 8 channels x 32 row groups: short serial loops (R/32) and C/8 blocks in flight
template <typename TIN>
__device__ __forceinline__ void reduce_two_columns(const TIN* __restrict__ part, int R, int C,
                                                   int c, int ry, int cx, double (&red)[2][32][9],
                                                   double& s0, double& s1) {
  double a0 = 0.0, a1 = 0.0;
  if (c < C) {
    // r06: eight rows (sixteen loads) in flight per round, added in the old order.  One row pair
    // per iteration made the loop a chain of memory round trips (hipcc waits vmcnt(0) before each
    // add): 5 trips for the 138 rows of a depthwise forward, 3 for a GEMM's 66.
    for (int r0 = ry; r0 < R; r0 += 32 * 8) {
      TIN v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long r = min(r0 + 32 * u, R - 1);
        v0[u] = part[r * 2 * C + c];
        v1[u] = part[r * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (r0 + 32 * u < R) {
          a0 += (double)v0[u];
          a1 += (double)v1[u];
        }
      }
    }
  }
  red[0][ry][cx] = a0;
  red[1][ry][cx] = a1;
  __syncthreads();
  s0 = s1 = 0.0;
  if (ry == 0) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      s0 += red[0][k][cx];
      s1 += red[1][k][cx];
    }
  }
}

// (All per-channel parameters are requested BEFORE the partial-row reduction: these kernels are
// pure latency — one dependent memory round trip instead of two or three: 5-6 us -> ~3 us.)
template <typename TIN>
__global__ __launch_bounds__(EW_THREADS) void bn_finalize_p_kernel(
    const TIN* __restrict__ part, int R, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* running_mean,
    float* running_var, float* mean_o, float* invstd_o, float* scale_o, float* shift_o, int C,
    const float* __restrict__ mean_offset, const P2PDev p2p, double* count_out) {
  __shared__ double red[2][32][9];
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cx;
  const bool fin = ry == 0 && c < C;
  float g = 1.f, b = 0.f, rm = 0.f, rv = 0.f, moff = 0.f;
  if (fin) {
    if (gamma) g = gamma[c];
    if (beta) b = beta[c];
    if (running_mean) { rm = running_mean[c]; rv = running_var[c]; }
    if (mean_offset) moff = mean_offset[c];
  }
  double sx, sxx;
  reduce_two_columns<TIN>(part, R, C, c, ry, cx, red, sx, sxx);
  if (p2p.world) {  // SyncBatchNorm: (sum, sum of squares, element count) summed over the ranks
    double v[3] = {sx, sxx, count};
    p2p_block_exchange<3>(p2p, blockIdx.x, gridDim.x, ry == 0, cx, 8, v);
    sx = v[0]; sxx = v[1]; count = v[2];
    if (count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = count;
  }
  if (!fin) return;
  const double mean = sx / count;
  double var = sxx / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  mean_o[c] = (float)mean;
  invstd_o[c] = (float)invstd;
  scale_o[c] = (float)((double)g * invstd);
  shift_o[c] = (float)((double)b - mean * (double)g * invstd);
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * (double)rm + momentum * (mean + (double)moff));
    running_var[c] = (float)((1.0 - momentum) * (double)rv + momentum * unbiased);
  }
}

// ---- BatchNorm over a SMALL number of samples (ASPP image pooling: N values per channel,
// module.py:52-64; PSP's pyramid bins: N*o*o = 2..72, module.py:89-97): statistics taken TWO-PASS
// from the stored tensor itself — mean first, then sum (x - mean)^2 — like ATen's CPU kernel.
// The single-pass form var = E[x^2] - mean^2 on fp32 partial sums loses 6e-8 * mean^2 / var:
// with two samples a, b per channel that is catastrophic as soon as |a - b| << |a| (measured
// r04 on the conditioned fixture: the one 2-sample BatchNorm of DeepLabv3+ put 2e-4 into the head's
// activations and, through ReLU-mask flips, 5e-2 into every encoder gradient of the fp32 path).
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void bn_finalize_small_kernel(
    const T* __restrict__ y, long ldy, int M, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* running_mean,
    float* running_var, float* mean_o, float* invstd_o, float* scale_o, float* shift_o, int C,
    const float* __restrict__ mean_offset, const P2PDev p2p, double* count_out,
    double* moments_out) {
  __shared__ double red[32][9];
  __shared__ double s_mean[8];
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cx;
  const bool fin = ry == 0 && c < C;
  float g = 1.f, b = 0.f, rm = 0.f, rv = 0.f, moff = 0.f;
  if (fin) {
    if (gamma) g = gamma[c];
    if (beta) b = beta[c];
    if (running_mean) { rm = running_mean[c]; rv = running_var[c]; }
    if (mean_offset) moff = mean_offset[c];
  }
  double a = 0.0;
  if (c < C)
    for (int m = ry; m < M; m += 32) a += (double)Vec<T>::load1(y + (long)m * ldy + c);
  red[ry][cx] = a;
  __syncthreads();
  if (ry == 0) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) s += red[k][cx];
    s_mean[cx] = s / count;
  }
  __syncthreads();
  const double mean = s_mean[cx];
  a = 0.0;
  if (c < C)
    for (int m = ry; m < M; m += 32) {
      const double d = (double)Vec<T>::load1(y + (long)m * ldy + c) - mean;
      a += d * d;
    }
  red[ry][cx] = a;
  __syncthreads();
  double ss = 0.0;
  if (ry == 0) {
#pragma unroll
    for (int k = 0; k < 32; ++k) ss += red[k][cx];
  }
  double mean_g = mean, var = ss / count;
  if (p2p.world || moments_out) {
    // SyncBatchNorm over a small tensor: this rank's TWO-PASS moments (mean_r, M2_r) are turned
    // into float64 sums (n*mean_r, M2_r + n*mean_r^2) and merged over the ranks — the parallel
    // (Chan) update in sum form; in float64 its cancellation is 1e-16 * mean^2, not the
    // 6e-8 * mean^2 of fp32 partial rows (the synchronised path kept E[x^2] - mean^2
    // on fp32 partials where the single-process path already took the statistics two-pass)
    double v[3] = {count * mean, ss + count * mean * mean, count};
    if (moments_out) {  // torch.distributed path: the all-reduce and seg_bn_finalize follow
      if (fin) {
        moments_out[c] = v[0];
        moments_out[C + c] = v[1];
      }
      if (blockIdx.x == 0 && threadIdx.x == 0) moments_out[2 * C] = count;
      return;
    }
    p2p_block_exchange<3>(p2p, blockIdx.x, gridDim.x, ry == 0, cx, 8, v);
    count = v[2];
    mean_g = v[0] / count;
    var = v[1] / count - mean_g * mean_g;
    if (var < 0.0) var = 0.0;
    if (count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = count;
  }
  if (!fin) return;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  mean_o[c] = (float)mean_g;
  invstd_o[c] = (float)invstd;
  scale_o[c] = (float)((double)g * invstd);
  shift_o[c] = (float)((double)b - mean_g * (double)g * invstd);
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * (double)rm + momentum * (mean_g + (double)moff));
    running_var[c] = (float)((1.0 - momentum) * (double)rv + momentum * unbiased);
  }
}

// ---- BatchNorm BACKWARD over a small number of samples (the same layers as
// bn_finalize_small_kernel), the whole of it in one launch and in float64:
//   g'  = g * chan_mul * elem_mul * relu_mask          (the mask from the fp32 forward value)
//   dx  = gamma * invstd * (g' - mean(g') - xhat * mean(g' * xhat)),  xhat = (x - mean) * invstd
// With two samples per channel xhat = +-1 up to eps and dx is the small remainder of three terms
// that cancel: the general path's fp32 `scale*g' - c0 - c1*x` keeps 6e-8 * |terms| of it — on the
// 513x1025 conditioned C3 step that one BatchNorm (ASPP image pooling, module.py:52-64) put
// 4.6e-3 into the weight gradient of its convolution and made up 94 % of the whole model's fp32
// gradient distance to the float64 oracle (profiles/r05_parity.txt).
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_small_kernel(
    const T* __restrict__ g, long ldg, const T* __restrict__ x, long ldx, T* __restrict__ dx,
    long lddx, int M, int C, int mode, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ chan_mul, int rows_per_n, const T* __restrict__ elem_mul, long ldm,
    double count, int training, float* dgamma, float* dbeta) {
  __shared__ double red[2][32][9];
  __shared__ double s_m[2][8];
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cx;
  const bool live = c < C;
  float sc = 1.f, sh = 0.f, gm = 1.f;
  double mu = 0.0, is = 1.0;
  if (live) {
    if (mode & PRO_AFFINE) { sc = scale[c]; sh = shift[c]; }
    if (gamma) gm = gamma[c];
    mu = mean[c];
    is = invstd[c];
  }
  auto masked = [&](int m, double& xh) -> double {
    const float xv = Vec<T>::load1(x + (long)m * ldx + c);
    double gv = (double)Vec<T>::load1(g + (long)m * ldg + c);
    if (chan_mul) gv *= (double)chan_mul[(long)(m / rows_per_n) * C + c];
    if (elem_mul) gv *= (double)Vec<T>::load1(elem_mul + (long)m * ldm + c);
    if (mode & PRO_RELU) {
      const float y = (mode & PRO_AFFINE) ? fmaf(xv, sc, sh) : xv;
      const bool on = y > 0.f && (!(mode & PRO_CLAMP6) || y < 6.f);
      gv = on ? gv : 0.0;
    }
    xh = ((double)xv - mu) * is;
    return gv;
  };
  double a0 = 0.0, a1 = 0.0;
  if (live)
    for (int m = ry; m < M; m += 32) {
      double xh;
      const double gv = masked(m, xh);
      a0 += gv;
      a1 += gv * xh;
    }
  red[0][ry][cx] = a0;
  red[1][ry][cx] = a1;
  __syncthreads();
  if (ry == 0) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      s0 += red[0][k][cx];
      s1 += red[1][k][cx];
    }
    s_m[0][cx] = s0;
    s_m[1][cx] = s1;
    if (live) {
      if (dgamma) dgamma[c] = (float)s1;
      if (dbeta) dbeta[c] = (float)s0;
    }
  }
  __syncthreads();
  if (!live) return;
  const double m1 = training ? s_m[0][cx] / count : 0.0, m2 = training ? s_m[1][cx] / count : 0.0;
  const double s = (double)gm * is;
  for (int m = ry; m < M; m += 32) {
    double xh;
    const double gv = masked(m, xh);  // (reads g[m][c] before dx[m][c] is written: in place is fine)
    Vec<T>::store1(dx + (long)m * lddx + c, (float)(s * (gv - m1 - xh * m2)));
  }
}

template <typename TIN>
__device__ __forceinline__ void bn_bwd_finalize_block(
    const TIN* __restrict__ part, int R, double count, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, float* dgamma, float* dbeta,
    float* c0_o, float* c1_o, int C, int block, double (&red)[2][32][9],
    const double* __restrict__ count_dev, double grad_scale, const P2PDev& p2p, int nblocks) {
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int c = block * 8 + cx;
  const bool fin = ry == 0 && c < C;
  float muf = 0.f, isf = 1.f, gf = 1.f;
  if (fin) {
    muf = mean[c];
    isf = invstd[c];
    if (gamma) gf = gamma[c];
    if (count_dev) count = *count_dev;  // SyncBatchNorm: the global element count
  }
  double sg, sgx;
  reduce_two_columns<TIN>(part, R, C, c, ry, cx, red, sg, sgx);
  if (p2p.world) {
    double v[2] = {sg, sgx};
    p2p_block_exchange<2>(p2p, block, nblocks, ry == 0, cx, 8, v);
    sg = v[0]; sgx = v[1];
  }
  if (!fin) return;
  const double mu = muf, is = isf;
  const double dg = (sgx - mu * sg) * is;
  const double s = (double)gf * is;
  const double m1 = sg / count, m2 = dg / count;
  const double c1 = s * m2 * is;
  if (dgamma) dgamma[c] = (float)(dg * grad_scale);
  if (dbeta) dbeta[c] = (float)(sg * grad_scale);
  c1_o[c] = (float)c1;
  c0_o[c] = (float)(s * m1 - c1 * mu);
}

template <typename TIN>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_finalize_p_kernel(
    const TIN* __restrict__ part, int R, double count, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, float* dgamma, float* dbeta,
    float* c0_o, float* c1_o, int C, const double* __restrict__ count_dev, double grad_scale,
    const P2PDev p2p) {
  __shared__ double red[2][32][9];
  bn_bwd_finalize_block<TIN>(part, R, count, mean, invstd, gamma, dgamma, dbeta, c0_o, c1_o, C,
                             blockIdx.x, red, count_dev, grad_scale, p2p, gridDim.x);
}

// The two reductions behind the fused depthwise backward in ONE launch (they were two: 65 + 78
// launches of 4-5 us per C3 step): blocks [0, nb_bn) finish the BatchNorm backward of the
// depthwise INPUT from its [Rb][2][C] partials, blocks [nb_bn, ..) sum the weight-gradient
// partials [Rw][9][C] into torch's [C,1,3,3] layout (fixed order, fp64).
__global__ __launch_bounds__(EW_THREADS) void dw_bwd_finalize_kernel(
    const float* __restrict__ part_bn, int Rb, double count, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, float* dgamma, float* dbeta,
    float* c0_o, float* c1_o, int C, int nb_bn, const float* __restrict__ part_w, int Rw,
    float* __restrict__ dw_out, const double* __restrict__ count_dev, double grad_scale,
    const P2PDev p2p) {
  __shared__ double red[2][32][9];
  if ((int)blockIdx.x < nb_bn) {  // (SyncBatchNorm: only these blocks exchange; dW stays local)
    bn_bwd_finalize_block<float>(part_bn, Rb, count, mean, invstd, gamma, dgamma, dbeta, c0_o,
                                 c1_o, C, blockIdx.x, red, count_dev, grad_scale, p2p, nb_bn);
    return;
  }
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int col = ((int)blockIdx.x - nb_bn) * 8 + cx;  // column of the [9*C] row: tap * C + c
  const int L = 9 * C;
  double acc = 0.0;
  if (col < L) {
    for (int r0 = ry; r0 < Rw; r0 += 32 * 8) {  // (eight loads in flight, as reduce_two_columns)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part_w[(long)min(r0 + 32 * u, Rw - 1) * L + col];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + 32 * u < Rw) acc += (double)v[u];
    }
  }
  red[0][ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && col < L) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[0][k][cx];
    const int tap = col / C, c = col - tap * C;
    dw_out[c * 9 + tap] = (float)t;
  }
}

static int pick_cvb_log2_ew(int CV) {
  int best = 5;
  double bu = 0;
  for (int l = 5; l >= 3; --l) {
    const int b = 1 << l;
    const double u = (double)CV / (double)(((CV + b - 1) / b) * b);
    if (u > bu + 1e-9) { bu = u; best = l; }
  }
  return best;
}

// Launch geometry of the row-tile kernels.  r06: a block covers WHOLE rows whenever a row has more
// than 32 channel vectors (C > 256 in bf16).  With power-of-two column blocks the 91 vectors of a
// 728-channel row were split over three blocks (512 + 512 + 432 bytes of every 1456-byte pixel,
// one block each): bn_bwd_apply moved its 73 MB at 4.6 TB/s where a flat 2-read-1-write stream
// of the same bytes runs at 6.9 TB/s (tools/lab/wb_lab).  Now lanes = the row's vectors, rows per
// block chosen so that rows x vectors fills a multiple of 64 threads (<= 512) best: a block
// iteration reads EW_UN x rows x C contiguous elements.
struct EwGeom { int lpr, rpb, threads, gx; };
static EwGeom ew_geom(int CV) {
  EwGeom g;
  if (CV <= 32 || CV > EW_MAX_THREADS) {
    const int l = pick_cvb_log2_ew(CV);
    g.lpr = 1 << l; g.rpb = EW_THREADS >> l; g.threads = EW_THREADS;
    g.gx = (CV + g.lpr - 1) >> l;
    return g;
  }
  g.lpr = CV; g.gx = 1; g.rpb = 1; g.threads = ((CV + 63) / 64) * 64;
  double best = -1.0;
  for (int r = 1; r * CV <= EW_MAX_THREADS; ++r) {
    const int t = ((r * CV + 63) / 64) * 64;
    // lane utilisation first, then a block near 256-384 threads
    const double score = (double)(r * CV) / t - 0.02 * (t < 256 ? (256 - t) / 256.0 : t > 384 ? (t - 384) / 384.0 : 0.0);
    if (score > best + 1e-9) { best = score; g.rpb = r; g.threads = t; }
  }
  return g;
}
static dim3 ew_grid2(const EwGeom& g, long M) {
  long gy = (M + (long)g.rpb * EW_UN - 1) / ((long)g.rpb * EW_UN);  // one batch of EW_UN rows per thread
  long cap = EW_CAP / g.gx;
  if (cap < 1) cap = 1;
  if (gy > cap) gy = cap;
  if (gy < 1) gy = 1;
  return dim3(g.gx, (unsigned)gy);
}

}  // namespace seg

// out = column sums of in[R][L]; ws must hold >= 64*L doubles.  Exactly one of out_d/out_f may be
// null.  Two-level when R is large.
extern "C" int seg_colsum(const float* in, long R, int L, double* out_d, float* out_f, double* ws,
                          void* stream) {
  using namespace seg;
  SEG_REQUIRE(R >= 1 && L >= 1, "colsum: empty");
  SEG_REQUIRE(out_d || out_f, "colsum: no output");
  hipStream_t st = (hipStream_t)stream;
  int gy = (int)((R + 127) / 128);
  if (gy > 64) gy = 64;
  const dim3 grid((L + 31) / 32, gy);
  if (gy == 1) {
    if (out_d)
      hipLaunchKernelGGL((colsum_kernel<double>), grid, dim3(EW_THREADS), 0, st, in, R, L, out_d, 0.0, 0);
    if (out_f)
      hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(EW_THREADS), 0, st, in, R, L, out_f, 0.0, 0);
    return check_launch("colsum");
  }
  SEG_REQUIRE(ws != nullptr, "colsum: workspace required for R=%ld", R);
  hipLaunchKernelGGL((colsum_kernel<double>), grid, dim3(EW_THREADS), 0, st, in, R, L, ws, 0.0, 0);
  hipLaunchKernelGGL(colsum_f64_kernel, dim3((L + EW_THREADS - 1) / EW_THREADS), dim3(EW_THREADS),
                     0, st, ws, gy, L, out_d, out_f, 0.0, 0);
  return check_launch("colsum");
}

// float64 column sums with the local element count appended: out_d[0..L) = sums, out_d[L] = count
extern "C" int seg_colsum_count(const float* in, long R, int L, double* out_d, double count,
                                double* ws, void* stream) {
  using namespace seg;
  SEG_REQUIRE(R >= 1 && L >= 1 && out_d, "colsum_count: empty");
  hipStream_t st = (hipStream_t)stream;
  int gy = (int)((R + 127) / 128);
  if (gy > 64) gy = 64;
  const dim3 grid((L + 31) / 32, gy);
  if (gy == 1) {
    hipLaunchKernelGGL((colsum_kernel<double>), grid, dim3(EW_THREADS), 0, st, in, R, L, out_d,
                       count, 1);
    return check_launch("colsum_count");
  }
  SEG_REQUIRE(ws != nullptr, "colsum_count: workspace required for R=%ld", R);
  hipLaunchKernelGGL((colsum_kernel<double>), grid, dim3(EW_THREADS), 0, st, in, R, L, ws, 0.0, 0);
  hipLaunchKernelGGL(colsum_f64_kernel, dim3((L + EW_THREADS - 1) / EW_THREADS), dim3(EW_THREADS),
                     0, st, ws, gy, L, out_d, (float*)nullptr, count, 1);
  return check_launch("colsum_count");
}

extern "C" int seg_bn_finalize(const double* sums, double count, const double* count_dev,
                               const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* mean, float* invstd, float* scale,
                               float* shift, int C, const float* mean_offset, void* stream) {
  using namespace seg;
  SEG_REQUIRE((count_dev || count >= 1.0) && C >= 1, "bn_finalize: bad count/C");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     sums, count, count_dev, gamma, beta, eps, momentum, running_mean, running_var, mean,
                     invstd, scale, shift, C, mean_offset);
  return check_launch("bn_finalize");
}

extern "C" int seg_bn_eval_affine(const float* gamma, const float* beta, const float* rm,
                                  const float* rv, float eps, float* scale, float* shift, int C,
                                  void* stream) {
  using namespace seg;
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, gamma, beta, rm, rv, eps, scale, shift, C);
  return check_launch("bn_eval_affine");
}

extern "C" int seg_bn_eval_affine_multi(int n, const float* const* gamma, const float* const* beta,
                                        const float* const* rm, const float* const* rv,
                                        const float* eps, float* const* out, const int* C,
                                        void* stream) {
  using namespace seg;
  SEG_REQUIRE(n >= 1, "bn_eval_affine_multi: no jobs");
  for (int j0 = 0; j0 < n; j0 += EVAL_MULTI_MAX) {
    EvalMultiArgs a;
    const int m = n - j0 < EVAL_MULTI_MAX ? n - j0 : EVAL_MULTI_MAX;
    for (int j = 0; j < m; ++j) {
      SEG_REQUIRE(C[j0 + j] >= 1 && C[j0 + j] <= 256 * EVAL_MULTI_BPJ && rm[j0 + j] && rv[j0 + j] &&
                      out[j0 + j],
                  "bn_eval_affine_multi: job %d: C=%d (1..%d) or a missing operand", j0 + j,
                  C[j0 + j], 256 * EVAL_MULTI_BPJ);
      a.gamma[j] = gamma[j0 + j]; a.beta[j] = beta[j0 + j]; a.rm[j] = rm[j0 + j];
      a.rv[j] = rv[j0 + j]; a.out[j] = out[j0 + j]; a.eps[j] = eps[j0 + j]; a.C[j] = C[j0 + j];
    }
    hipLaunchKernelGGL(bn_eval_affine_multi_kernel, dim3(m * EVAL_MULTI_BPJ), dim3(256), 0,
                       (hipStream_t)stream, a);
  }
  return check_launch("bn_eval_affine_multi");
}

namespace seg {
// ---- n-ary sum: y = x_0 + x_1 + ... + x_{n-1} (2 <= n <= 8), fp32 accumulation in index order,
// ONE rounding — the gradient of an activation with several consumers (the five ASPP branches of
// c4, module.py:52-70; shortcut + first separable conv of a conv-skip block, xception.py:36-40),
// which torch autograd would accumulate with n-1 element-wise `add` launches.
constexpr int SUMN_MAX = 8;
struct SumNArgs {
  const void* x[SUMN_MAX];
  long ld[SUMN_MAX];
  void* y;
  long ldy, M;
  int n, CV, lpr, rpb;
};

template <typename T>
__global__ __launch_bounds__(EW_MAX_THREADS) void sum_n_kernel(const SumNArgs a) {
  constexpr int VEC = Vec<T>::N;
  const int cvb = a.lpr, rpb = a.rpb;
  const int sy = threadIdx.x / cvb, cx = threadIdx.x - sy * cvb;
  const int cv = blockIdx.x * cvb + cx;
  if (cv >= a.CV || sy >= rpb) return;
  const int c0 = cv * VEC;
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const int M = (int)a.M, step = gridDim.y * rpb;
  for (int row = blockIdx.y * rpb + sy; row < M; row += step) {
    uint4 r[SUMN_MAX];
#pragma unroll
    for (int k = 0; k < SUMN_MAX; ++k)
      if (k < a.n) r[k] = ldg16(reinterpret_cast<const T*>(a.x[k]) + (long)row * a.ld[k] + c0);
    float acc[VEC];
    Vec<T>::unpack(r[0], acc);
#pragma unroll
    for (int k = 1; k < SUMN_MAX; ++k)
      if (k < a.n) {
        float f[VEC];
        Vec<T>::unpack(r[k], f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += f[i];
      }
    stg16(Y + (long)row * a.ldy + c0, Vec<T>::pack(acc));
  }
}
}  // namespace seg

// y = post_relu?( act_x(x) * chan_mul + act_r(r) )
extern "C" int seg_bn_apply(int dtype, const void* x, long ldx, int mode_x, const float* sx,
                            const float* tx, const void* r, long ldr, int mode_r, const float* sr,
                            const float* tr, const float* chan_mul, long rows_per_n,
                            const void* elem_mul, long ldm, int post_relu, void* y, long ldy,
                            long M, int C, void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "bn_apply: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldx % vec == 0 && ldy % vec == 0 && (r == nullptr || ldr % vec == 0),
              "bn_apply: C/ld must be multiples of %d", vec);
  SEG_REQUIRE(((mode_x & PRO_AFFINE) == 0) || (sx && tx), "bn_apply: missing scale/shift (x)");
  SEG_REQUIRE(r == nullptr || ((mode_r & PRO_AFFINE) == 0) || (sr && tr),
              "bn_apply: missing scale/shift (r)");
  ApplyArgs a;
  a.x = x; a.r = r; a.y = y; a.sx = sx; a.tx = tx; a.sr = sr; a.tr = tr; a.chan_mul = chan_mul;
  a.elem_mul = elem_mul; a.ldm = ldm;
  a.ldx = ldx; a.ldr = ldr; a.ldy = ldy; a.M = M; a.C = C; a.CV = C / vec;
  a.mode_x = mode_x; a.mode_r = mode_r; a.post_relu = post_relu;
  a.rows_per_n = rows_per_n > 0 ? rows_per_n : 1;
  SEG_REQUIRE(M < (1L << 31), "bn_apply: M overflows int");
  const EwGeom geo = ew_geom(a.CV);
  a.lpr = geo.lpr; a.rpb = geo.rpb;
  const dim3 grid = ew_grid2(geo, M);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((bn_apply_kernel<bf16_t>), grid, dim3(geo.threads), 0,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((bn_apply_kernel<float>), grid, dim3(geo.threads), 0,
                       (hipStream_t)stream, a);
  return check_launch("bn_apply");
}

extern "C" int seg_sum_n(int dtype, int n, const void* const* xs, const long* lds, void* y, long ldy,
                         long M, int C, void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "sum_n: bad dtype %d", dtype);
  SEG_REQUIRE(n >= 2 && n <= SUMN_MAX, "sum_n: 2..%d operands", SUMN_MAX);
  SEG_REQUIRE(C % vec == 0 && ldy % vec == 0 && M >= 1 && M < (1L << 31), "sum_n: C/ld/M");
  SumNArgs a;
  for (int k = 0; k < SUMN_MAX; ++k) {
    a.x[k] = k < n ? xs[k] : nullptr;
    a.ld[k] = k < n ? lds[k] : 0;
    SEG_REQUIRE(k >= n || (xs[k] != nullptr && lds[k] % vec == 0 && lds[k] >= C), "sum_n: operand %d", k);
  }
  a.y = y; a.ldy = ldy; a.M = M; a.n = n; a.CV = C / vec;
  const EwGeom geo = ew_geom(a.CV);
  a.lpr = geo.lpr; a.rpb = geo.rpb;
  const dim3 grid = ew_grid2(geo, M);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((sum_n_kernel<bf16_t>), grid, dim3(geo.threads), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((sum_n_kernel<float>), grid, dim3(geo.threads), 0, (hipStream_t)stream, a);
  return check_launch("sum_n");
}

extern "C" int seg_bn_bwd_grid_y(int dtype, int C, long M) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  const int CV = C / vec;
  const EwGeom geo = ew_geom(CV);
  const int spb = geo.rpb, gx = geo.gx;
  long gy = (M + (long)spb * 8 - 1) / ((long)spb * 8);  // >= 8 rows per thread
  long cap = 2048 / gx;
  // one block per CU: 524 partial rows made the reduce 12.3 us and the finalize behind it 6.1 us on
  // the 24 MB tensors; 256 rows: 10.7 + 4.0 us (128: 11.9 + 3.5, 384: 11.3 + 4.8)
  if (cap > 256) cap = 256;
  if (cap < 1) cap = 1;
  if (gy > cap) gy = cap;
  if (gy < 1) gy = 1;
  return (int)gy;
}

// partial[grid_y][2][C] = per-block (sum g', sum g'*x),  g' = g * chan_mul * relu_mask(mode)
extern "C" int seg_bn_bwd_reduce(int dtype, const void* g, long ldg, const void* x, long ldx,
                                 int mode, const float* scale, const float* shift,
                                 const float* chan_mul, long rows_per_n, const void* elem_mul,
                                 long ldm, long M, int C, float* partial, int grid_y,
                                 void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "bn_bwd_reduce: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldg % vec == 0 && ldx % vec == 0,
              "bn_bwd_reduce: C/ld must be multiples of %d", vec);
  SEG_REQUIRE(grid_y >= 1, "bn_bwd_reduce: grid_y");
  BwdArgs a;
  a.g = g; a.x = x; a.dx = nullptr; a.scale = scale; a.shift = shift; a.c0 = nullptr; a.c1 = nullptr;
  a.chan_mul = chan_mul; a.rows_per_n = rows_per_n > 0 ? rows_per_n : 1;
  a.elem_mul = elem_mul; a.ldm = ldm;
  a.partial = partial; a.ldg = ldg; a.ldx = ldx; a.lddx = 0; a.M = M; a.C = C; a.CV = C / vec;
  a.mode = mode;
  const EwGeom geo = ew_geom(a.CV);
  a.lpr = geo.lpr; a.rpb = geo.rpb;
  const int gx = geo.gx;
  const size_t lds = (size_t)geo.threads * 2 * vec * sizeof(float);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t>), dim3(gx, grid_y), dim3(geo.threads), lds,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<float>), dim3(gx, grid_y), dim3(geo.threads), lds,
                       (hipStream_t)stream, a);
  return check_launch("bn_bwd_reduce");
}

// The whole BatchNorm backward of a SMALL tensor (M <= 4096 rows) in one launch, float64 inside:
// dx (may alias g), dgamma, dbeta.  training = 0: evaluation-mode BatchNorm (dx = scale * g').
extern "C" int seg_bn_bwd_small(int dtype, const void* g, long ldg, const void* x, long ldx,
                                void* dx, long lddx, long M, int C, int mode, const float* scale,
                                const float* shift, const float* mean, const float* invstd,
                                const float* gamma, const float* chan_mul, long rows_per_n,
                                const void* elem_mul, long ldm, double count, int training,
                                float* dgamma, float* dbeta, void* stream) {
  using namespace seg;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "bn_bwd_small: bad dtype %d", dtype);
  SEG_REQUIRE(M >= 1 && M <= 4096 && C >= 1 && count >= 1.0 && mean && invstd && dx,
              "bn_bwd_small: bad M/C/count or missing statistics");
  SEG_REQUIRE(((mode & PRO_AFFINE) == 0) || (scale && shift), "bn_bwd_small: affine without scale");
  const dim3 grid((C + 7) / 8);
  const int rpn = rows_per_n > 0 ? (int)rows_per_n : 1;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((bn_bwd_small_kernel<bf16_t>), grid, dim3(EW_THREADS), 0,
                       (hipStream_t)stream, (const bf16_t*)g, ldg, (const bf16_t*)x, ldx,
                       (bf16_t*)dx, lddx, (int)M, C, mode, scale, shift, mean, invstd, gamma,
                       chan_mul, rpn, (const bf16_t*)elem_mul, ldm, count, training, dgamma, dbeta);
  else
    hipLaunchKernelGGL((bn_bwd_small_kernel<float>), grid, dim3(EW_THREADS), 0,
                       (hipStream_t)stream, (const float*)g, ldg, (const float*)x, ldx, (float*)dx,
                       lddx, (int)M, C, mode, scale, shift, mean, invstd, gamma, chan_mul, rpn,
                       (const float*)elem_mul, ldm, count, training, dgamma, dbeta);
  return check_launch("bn_bwd_small");
}

extern "C" int seg_bn_bwd_finalize(const double* sums, double count, const double* count_dev,
                                   const float* mean,
                                   const float* invstd, const float* gamma, float* dgamma,
                                   float* dbeta, float* c0, float* c1, int C, void* stream) {
  using namespace seg;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, sums, count, count_dev, mean, invstd, gamma, dgamma,
                     dbeta, c0, c1, C, 1.0);
  return check_launch("bn_bwd_finalize");
}

// the same with dgamma / dbeta multiplied by grad_scale (SyncBatchNorm: 1 / world size)
extern "C" int seg_bn_bwd_finalize_s(const double* sums, double count, const double* count_dev,
                                     const float* mean, const float* invstd, const float* gamma,
                                     float* dgamma, float* dbeta, float* c0, float* c1, int C,
                                     double grad_scale, void* stream) {
  using namespace seg;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, sums, count, count_dev, mean, invstd, gamma, dgamma,
                     dbeta, c0, c1, C, grad_scale);
  return check_launch("bn_bwd_finalize");
}

// dx = scale * g' - c0 - c1 * x   (mode has AFFINE);  dx = g'  (mode without AFFINE: plain ReLU
// backward / dropout-mask backward).  c0/c1 null with AFFINE -> eval-mode BN backward (scale only).
extern "C" int seg_bn_bwd_apply(int dtype, const void* g, long ldg, const void* x, long ldx,
                                int mode, const float* scale, const float* shift, const float* c0,
                                const float* c1, const float* chan_mul, long rows_per_n,
                                const void* elem_mul, long ldm, void* dx, long lddx, long M, int C,
                                void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "bn_bwd_apply: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldg % vec == 0 && ldx % vec == 0 && lddx % vec == 0,
              "bn_bwd_apply: C/ld must be multiples of %d", vec);
  SEG_REQUIRE(((mode & PRO_AFFINE) == 0) || (scale && shift), "bn_bwd_apply: missing scale/shift");
  BwdArgs a;
  a.g = g; a.x = x; a.dx = dx; a.scale = scale; a.shift = shift; a.c0 = c0; a.c1 = c1;
  a.chan_mul = chan_mul; a.rows_per_n = rows_per_n > 0 ? rows_per_n : 1;
  a.elem_mul = elem_mul; a.ldm = ldm;
  a.partial = nullptr; a.ldg = ldg; a.ldx = ldx; a.lddx = lddx; a.M = M; a.C = C; a.CV = C / vec;
  a.mode = mode;
  SEG_REQUIRE(M < (1L << 31), "bn_bwd_apply: M overflows int");
  const EwGeom geo = ew_geom(a.CV);
  a.lpr = geo.lpr; a.rpb = geo.rpb;
  const dim3 grid = ew_grid2(geo, M);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t>), grid, dim3(geo.threads), 0,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), grid, dim3(geo.threads), 0,
                       (hipStream_t)stream, a);
  return check_launch("bn_bwd_apply");
}

// Fused (non-sync) variants: partial rows [R][2][C] fp32 straight from the conv / reduce kernels.
// ws: >= 64*2*C doubles, only touched when R > 1024 (two-level reduction).
extern "C" int seg_bn_finalize_p(const float* partial, long R, double count, const float* gamma,
                                 const float* beta, float eps, float momentum,
                                 float* running_mean, float* running_var, float* mean,
                                 float* invstd, float* scale, float* shift, int C,
                                 const float* mean_offset, double* ws, void* stream) {
  using namespace seg;
  SEG_REQUIRE(count >= 1.0 && C >= 1 && R >= 1, "bn_finalize_p: bad count/C/R");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((C + 7) / 8);
  if (R <= 1024) {
    hipLaunchKernelGGL((bn_finalize_p_kernel<float>), grid, dim3(EW_THREADS), 0, st, partial,
                       (int)R, count, gamma, beta, eps, momentum, running_mean, running_var, mean,
                       invstd, scale, shift, C, mean_offset, p2p_dev_none(), (double*)nullptr);
  } else {
    SEG_REQUIRE(ws != nullptr, "bn_finalize_p: workspace required for R=%ld", R);
    hipLaunchKernelGGL((colsum_kernel<double>), dim3((2 * C + 31) / 32, 64), dim3(EW_THREADS), 0,
                       st, partial, R, 2 * C, ws, 0.0, 0);
    hipLaunchKernelGGL((bn_finalize_p_kernel<double>), grid, dim3(EW_THREADS), 0, st, ws, 64,
                       count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd,
                       scale, shift, C, mean_offset, p2p_dev_none(), (double*)nullptr);
  }
  return check_launch("bn_finalize_p");
}

static int launch_bn_finalize_small(const char* what, int dtype, const void* y, long ldy, long M,
                                    int C, const float* gamma, const float* beta, float eps,
                                    float momentum, float* running_mean, float* running_var,
                                    float* mean, float* invstd, float* scale, float* shift,
                                    const float* mean_offset, const seg::P2PDev& d,
                                    double* count_out, double* moments_out, void* stream) {
  using namespace seg;
  SEG_REQUIRE(M >= 1 && M <= 4096 && C >= 1 && ldy >= C, "%s: bad M/C/pitch", what);
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "%s: bad dtype", what);
  const dim3 grid((C + 7) / 8);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((bn_finalize_small_kernel<bf16_t>), grid, dim3(EW_THREADS), 0,
                       (hipStream_t)stream, (const bf16_t*)y, ldy, (int)M, (double)M, gamma, beta,
                       eps, momentum, running_mean, running_var, mean, invstd, scale, shift, C,
                       mean_offset, d, count_out, moments_out);
  else
    hipLaunchKernelGGL((bn_finalize_small_kernel<float>), grid, dim3(EW_THREADS), 0,
                       (hipStream_t)stream, (const float*)y, ldy, (int)M, (double)M, gamma, beta,
                       eps, momentum, running_mean, running_var, mean, invstd, scale, shift, C,
                       mean_offset, d, count_out, moments_out);
  return check_launch(what);
}

extern "C" int seg_bn_finalize_small(int dtype, const void* y, long ldy, long M, int C,
                                     const float* gamma, const float* beta, float eps,
                                     float momentum, float* running_mean, float* running_var,
                                     float* mean, float* invstd, float* scale, float* shift,
                                     const float* mean_offset, void* stream) {
  return launch_bn_finalize_small("bn_finalize_small", dtype, y, ldy, M, C, gamma, beta, eps,
                                  momentum, running_mean, running_var, mean, invstd, scale, shift,
                                  mean_offset, seg::p2p_dev_none(), nullptr, nullptr, stream);
}

// SyncBatchNorm over a small tensor, torch.distributed path: this rank's two-pass moments as
// float64 sums [2C + 1] = (n*mean_r | M2_r + n*mean_r^2 | n) — all-reduce, then seg_bn_finalize.
extern "C" int seg_bn_moments_small(int dtype, const void* y, long ldy, long M, int C,
                                    double* moments, void* stream) {
  SEG_REQUIRE(moments != nullptr, "bn_moments_small: no output");
  return launch_bn_finalize_small("bn_moments_small", dtype, y, ldy, M, C, nullptr, nullptr, 0.f,
                                  0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                  nullptr, seg::p2p_dev_none(), nullptr, moments, stream);
}

// ... peer-mailbox path: moments merged inside the launch (as seg_bn_finalize_p_sync)
extern "C" int seg_bn_finalize_small_sync(void* p2p, int dtype, const void* y, long ldy, long M,
                                          int C, const float* gamma, const float* beta, float eps,
                                          float momentum, float* running_mean, float* running_var,
                                          float* mean, float* invstd, float* scale, float* shift,
                                          const float* mean_offset, double* count_out,
                                          void* stream) {
  using namespace seg;
  P2PDev d;
  if (!p2p_dev_of(p2p, d)) return 2;
  const int nb = (C + 7) / 8;
  SEG_REQUIRE(nb <= P2P_MAX_BLOCKS && (long)nb * 8 * 3 * 8 <= d.slot_bytes,
              "bn_finalize_small_sync: C=%d exceeds the mailbox", C);
  return launch_bn_finalize_small("bn_finalize_small_sync", dtype, y, ldy, M, C, gamma, beta, eps,
                                  momentum, running_mean, running_var, mean, invstd, scale, shift,
                                  mean_offset, d, count_out, nullptr, stream);
}

extern "C" int seg_bn_bwd_finalize_p(const float* partial, long R, double count,
                                     const float* mean, const float* invstd, const float* gamma,
                                     float* dgamma, float* dbeta, float* c0, float* c1, int C,
                                     double* ws, void* stream) {
  using namespace seg;
  SEG_REQUIRE(count >= 1.0 && C >= 1 && R >= 1, "bn_bwd_finalize_p: bad count/C/R");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((C + 7) / 8);
  if (R <= 1024) {
    hipLaunchKernelGGL((bn_bwd_finalize_p_kernel<float>), grid, dim3(EW_THREADS), 0, st, partial,
                       (int)R, count, mean, invstd, gamma, dgamma, dbeta, c0, c1, C,
                       (const double*)nullptr, 1.0, p2p_dev_none());
  } else {
    SEG_REQUIRE(ws != nullptr, "bn_bwd_finalize_p: workspace required for R=%ld", R);
    hipLaunchKernelGGL((colsum_kernel<double>), dim3((2 * C + 31) / 32, 64), dim3(EW_THREADS), 0,
                       st, partial, R, 2 * C, ws, 0.0, 0);
    hipLaunchKernelGGL((bn_bwd_finalize_p_kernel<double>), grid, dim3(EW_THREADS), 0, st, ws, 64,
                       count, mean, invstd, gamma, dgamma, dbeta, c0, c1, C, (const double*)nullptr,
                       1.0, p2p_dev_none());
  }
  return check_launch("bn_bwd_finalize_p");
}

extern "C" int seg_dw_bwd_finalize(const float* partial_bn, int Rb, double count, const float* mean,
                                   const float* invstd, const float* gamma, float* dgamma,
                                   float* dbeta, float* c0, float* c1, const float* partial_w,
                                   int Rw, float* dw_c9, int C, void* stream) {
  using namespace seg;
  SEG_REQUIRE(count >= 1.0 && C >= 1 && Rb >= 1 && Rw >= 1 && Rb <= 1024,
              "dw_bwd_finalize: bad count/C/rows");
  const int nb_bn = (C + 7) / 8;
  hipLaunchKernelGGL(dw_bwd_finalize_kernel, dim3(nb_bn + (9 * C + 7) / 8), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, partial_bn, Rb, count, mean, invstd, gamma, dgamma, dbeta,
                     c0, c1, C, nb_bn, partial_w, Rw, dw_c9, (const double*)nullptr, 1.0,
                     p2p_dev_none());
  return check_launch("dw_bwd_finalize");
}

// ---- the same three finalize steps for SyncBatchNorm: the sums are exchanged between the ranks
// INSIDE the kernel (p2p.h: one-hop peer writes), so the launch count equals plain BatchNorm's.
// `p2p` = handle of seg_p2p_create; every rank issues the same calls in the same order.
extern "C" int seg_bn_finalize_p_sync(void* p2p, const float* partial, long R, double local_count,
                                      const float* gamma, const float* beta, float eps,
                                      float momentum, float* running_mean, float* running_var,
                                      float* mean, float* invstd, float* scale, float* shift, int C,
                                      const float* mean_offset, double* count_out, double* ws,
                                      void* stream) {
  using namespace seg;
  SEG_REQUIRE(local_count >= 0.0 && C >= 1 && R >= 1, "bn_finalize_p_sync: bad count/C/R");
  P2PDev d;
  if (!p2p_dev_of(p2p, d)) return 2;
  const int nb = (C + 7) / 8;
  SEG_REQUIRE(nb <= P2P_MAX_BLOCKS && (long)nb * 8 * 3 * 8 <= d.slot_bytes,
              "bn_finalize_p_sync: C=%d exceeds the mailbox", C);
  hipStream_t st = (hipStream_t)stream;
  if (R <= 1024) {
    hipLaunchKernelGGL((bn_finalize_p_kernel<float>), dim3(nb), dim3(EW_THREADS), 0, st, partial,
                       (int)R, local_count, gamma, beta, eps, momentum, running_mean, running_var,
                       mean, invstd, scale, shift, C, mean_offset, d, count_out);
  } else {  // two-level, as seg_bn_finalize_p
    SEG_REQUIRE(ws != nullptr, "bn_finalize_p_sync: workspace required for R=%ld", R);
    hipLaunchKernelGGL((colsum_kernel<double>), dim3((2 * C + 31) / 32, 64), dim3(EW_THREADS), 0,
                       st, partial, R, 2 * C, ws, 0.0, 0);
    hipLaunchKernelGGL((bn_finalize_p_kernel<double>), dim3(nb), dim3(EW_THREADS), 0, st, ws, 64,
                       local_count, gamma, beta, eps, momentum, running_mean, running_var, mean,
                       invstd, scale, shift, C, mean_offset, d, count_out);
  }
  return check_launch("bn_finalize_p_sync");
}

extern "C" int seg_bn_bwd_finalize_p_sync(void* p2p, const float* partial, long R,
                                          const double* count_dev, const float* mean,
                                          const float* invstd, const float* gamma, float* dgamma,
                                          float* dbeta, float* c0, float* c1, int C,
                                          double grad_scale, double* ws, void* stream) {
  using namespace seg;
  SEG_REQUIRE(count_dev && C >= 1 && R >= 1, "bn_bwd_finalize_p_sync: bad count/C/R");
  P2PDev d;
  if (!p2p_dev_of(p2p, d)) return 2;
  const int nb = (C + 7) / 8;
  SEG_REQUIRE(nb <= P2P_MAX_BLOCKS && (long)nb * 8 * 2 * 8 <= d.slot_bytes,
              "bn_bwd_finalize_p_sync: C=%d exceeds the mailbox", C);
  hipStream_t st = (hipStream_t)stream;
  if (R <= 1024) {
    hipLaunchKernelGGL((bn_bwd_finalize_p_kernel<float>), dim3(nb), dim3(EW_THREADS), 0, st,
                       partial, (int)R, 1.0, mean, invstd, gamma, dgamma, dbeta, c0, c1, C,
                       count_dev, grad_scale, d);
  } else {
    SEG_REQUIRE(ws != nullptr, "bn_bwd_finalize_p_sync: workspace required for R=%ld", R);
    hipLaunchKernelGGL((colsum_kernel<double>), dim3((2 * C + 31) / 32, 64), dim3(EW_THREADS), 0,
                       st, partial, R, 2 * C, ws, 0.0, 0);
    hipLaunchKernelGGL((bn_bwd_finalize_p_kernel<double>), dim3(nb), dim3(EW_THREADS), 0, st, ws,
                       64, 1.0, mean, invstd, gamma, dgamma, dbeta, c0, c1, C, count_dev,
                       grad_scale, d);
  }
  return check_launch("bn_bwd_finalize_p_sync");
}

extern "C" int seg_dw_bwd_finalize_sync(void* p2p, const float* partial_bn, int Rb,
                                        const double* count_dev, const float* mean,
                                        const float* invstd, const float* gamma, float* dgamma,
                                        float* dbeta, float* c0, float* c1, const float* partial_w,
                                        int Rw, float* dw_c9, int C, double grad_scale,
                                        void* stream) {
  using namespace seg;
  SEG_REQUIRE(count_dev && C >= 1 && Rb >= 1 && Rw >= 1 && Rb <= 1024,
              "dw_bwd_finalize_sync: bad count/C/rows");
  P2PDev d;
  if (!p2p_dev_of(p2p, d)) return 2;
  const int nb_bn = (C + 7) / 8;
  SEG_REQUIRE(nb_bn <= P2P_MAX_BLOCKS && (long)nb_bn * 8 * 2 * 8 <= d.slot_bytes,
              "dw_bwd_finalize_sync: C=%d exceeds the mailbox", C);
  hipLaunchKernelGGL(dw_bwd_finalize_kernel, dim3(nb_bn + (9 * C + 7) / 8), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, partial_bn, Rb, 1.0, mean, invstd, gamma, dgamma, dbeta,
                     c0, c1, C, nb_bn, partial_w, Rw, dw_c9, count_dev, grad_scale, d);
  return check_launch("dw_bwd_finalize_sync");
}
