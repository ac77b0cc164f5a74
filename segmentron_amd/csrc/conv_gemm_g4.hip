// 1x1 / stride-1 (and stride-1 KxK) convolution forward and data gradient, fourth generation (r05):
// the direct-to-LDS ring of gemm_glds.h (bf16, no element-wise prologue on the pixel operand)
// with (a) the tile ROWS chosen per launch — 256 or 192 — so that the tiles fill the 256 CUs in
// fewer or shorter rounds, and (b) a register-direct epilogue.  Same argument block and
// BatchNorm-statistics format as the earlier generations (per-tile (sum, sum of squares) rows
// taken from the values AS STORED; folded-BN backward correction y = acc - c0[o] - c1[o]*x[p][o]
// in the store path), except that a statistics row describes 256 OR 192 pixels
// (seg_conv_gemm_stat_rows tells the caller how many rows there are).
//
// Epilogue.  v_mfma_f32_32x32x16_bf16 with the weights as the row operand leaves lane
// (px = lane & 31, hh = lane >> 5) with channels 8g + 4hh .. +3 (g = 0..3) of pixel px for every
// 32-channel block.  For the pair (g = 2t, 2t+1) one v_permlane32_swap per register (upper half
// of X <-> lower half of Y) gives every lane 8 CONSECUTIVE channels of its pixel, 16t + 8hh .. +7:
// one 16-byte NHWC store per lane and pair, no LDS patch and no LDS round trip between the
// accumulators and the stores (r02-r04: 8-byte pieces -> LDS patch -> 16-byte vectors, ~7 us of
// a 30 us launch; the data-gradient variant also rounded twice there — up to 50 bf16 ulps of
// error where the correction cancels the accumulator, tools/lab/gemm_ab).  The statistics are
// lane-local sums over the wave's pixel blocks; the two column blocks of a wave are folded with
// one v_permlane16_swap per pair of values, the 16 pixel lanes of a row with four DPP adds.
//
// Tried and dropped in r05 (profiles/r05_gemm_ab.md): the same kernel on v_mfma_f32_16x16x32_bf16
// with lane-linear fragment reads and 224-row tiles (gemm_g4 16x16 variant) — its epilogue was
// as fast, 225 tiles of 224 rows beat 198 of 256 on 728 -> 728 (27.0 vs 28.4 us), but the
// 16x16x32 main loop is 13-17 % slower per k-step than the 32x32x16 one (74 vs 63 us on
// 1024 -> 1536, 151 vs 134 us on 1536 -> 2048).
#include "conv_gemm.h"
#include "conv_gemm_args.h"
#include "gemm_glds.h"

namespace seg {

// main loop: 0 = the lockstep ring, 1 = ping-pong wave groups, 2 = ping-pong + s_setprio around
// the MFMA interval (lab switch; the production choice is the default)
#ifndef G4_LOOP
#define G4_LOOP 0
#endif
#if G4_LOOP == 0
#define G4_MAINLOOP(KXK_) gl_mainloop_ring<KXK_, IMS>
#elif G4_LOOP == 1
#define G4_MAINLOOP(KXK_) gl_mainloop_pingpong<KXK_, IMS, false>
#else
#define G4_MAINLOOP(KXK_) gl_mainloop_pingpong<KXK_, IMS, true>
#endif

constexpr int G4_BN = 256;
constexpr int G4_THREADS = GL_THREADS;
constexpr int G4_LDS_BYTES = GL_LDS_BYTES;

__device__ __forceinline__ void g4_swap32(uint32_t& x, uint32_t& y) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  x = r[0];
  y = r[1];
}

// u, w: the same statistic of column blocks 0 and 1.  Returns, in 16-lane rows of even parity
// (lanes 0-15, 32-47) u summed over the two rows of its half-wave, in rows of odd parity w — and
// every lane of a row then gets the row total (quad_perm x2, row_half_mirror, row_mirror).
__device__ __forceinline__ float g4_fold(float u, float w) {
  uint32_t a = __float_as_uint(u), b = __float_as_uint(w);
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  float v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, true));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, true));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xF, 0xF, true));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xF, 0xF, true));
  return v;
}

// IMS: 32-pixel blocks per wave (4: 256-row tile, 3: 192-row tile); EP: folded-BatchNorm backward
// correction in the store path; STATS: BatchNorm partial sums; KXK: stride-1 KxK convolution as an
// implicit GEMM (per-lane gather in the DMA source address)
template <int IMS, bool EP, bool STATS, bool KXK>
__global__ __launch_bounds__(G4_THREADS, 2) void conv_gemm_g4_kernel(const ConvGemmArgs a) {
  typedef bf16_t T;
  constexpr int BM = 64 * IMS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lds_byte_t* lds = (lds_byte_t*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int L = xcd_remap(blockIdx.x, a.tiles_m * a.tiles_n);
  const int tile_m = L / a.tiles_n, tile_n = L - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * G4_BN;

  GemmOperand A, B;
  A.base = reinterpret_cast<const unsigned char*>(a.x);
  A.ld_bytes = a.ldx * 2;
  A.rows = a.M;
  B.base = reinterpret_cast<const unsigned char*>(a.w);
  B.ld_bytes = (long)a.K * 2;
  B.rows = a.O;

  f32x16 acc[2][IMS];  // (the ring starts from a constant-zero accumulator INPUT)
  if (KXK) {
    const GlConvA cg = {a.M, a.Hi, a.Wi, a.Ho, a.Wo, a.KW, a.pad, a.dil, a.C / 32};
    G4_MAINLOOP(true)(A, B, a.K, m0, n0, lds, acc, &cg);
  } else {
    G4_MAINLOOP(false)(A, B, a.K, m0, n0, lds, acc);
  }

  // ---- epilogue.  Lane (px, hh); vector (jn, t): channels n0 + wn*64 + jn*32 + 16t + 8hh .. +7
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const int px = lane & 31, hh = lane >> 5;
  const int o0 = n0 + wn * 64 + 8 * hh;  // + jn * 32 + t * 16
  bool ook[2][2];                         // (O % 8 == 0: a vector is inside or outside)
#pragma unroll
  for (int jn = 0; jn < 2; ++jn)
#pragma unroll
    for (int t = 0; t < 2; ++t) ook[jn][t] = o0 + jn * 32 + t * 16 < a.O;
  float ssum[2][2][8], ssq[2][2][8];
  if (STATS) {
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k = 0; k < 8; ++k) ssum[jn][t][k] = ssq[jn][t][k] = 0.f;
  }
#pragma unroll
  for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int o = o0 + jn * 32 + t * 16;
      const int oc = ook[jn][t] ? o : 0;
      float c0v[8], c1v[8];
      uint4 xr[IMS];
      if (EP) {  // this vector column's ep_x loads of all pixel blocks are requested together
        load_params<8>(a.ep_c0, oc, c0v);
        load_params<8>(a.ep_c1, oc, c1v);
#pragma unroll
        for (int im = 0; im < IMS; ++im) {
          const int p = m0 + (wm * IMS + im) * 32 + px;
          const long pc = p < a.M ? p : a.M - 1;
          xr[im] = ldg16(reinterpret_cast<const T*>(a.ep_x) + pc * a.ldep + oc);
        }
      }
#pragma unroll
      for (int im = 0; im < IMS; ++im) {
        const int p = m0 + (wm * IMS + im) * 32 + px;
        uint4 val;
        if (EP) {  // swap in fp32, correct, round ONCE
          float f[8], xv[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            uint32_t x = __float_as_uint(acc[jn][im][8 * t + k]);
            uint32_t y = __float_as_uint(acc[jn][im][8 * t + 4 + k]);
            g4_swap32(x, y);
            f[k] = __uint_as_float(x);
            f[4 + k] = __uint_as_float(y);
          }
          Vec<T>::unpack(xr[im], xv);
#pragma unroll
          for (int k = 0; k < 8; ++k) f[k] = f[k] - c0v[k] - c1v[k] * xv[k];
          val = Vec<T>::pack(f);
        } else {
          uint32_t x01 = pack_bf16x2(acc[jn][im][8 * t + 0], acc[jn][im][8 * t + 1]);
          uint32_t x23 = pack_bf16x2(acc[jn][im][8 * t + 2], acc[jn][im][8 * t + 3]);
          uint32_t y01 = pack_bf16x2(acc[jn][im][8 * t + 4], acc[jn][im][8 * t + 5]);
          uint32_t y23 = pack_bf16x2(acc[jn][im][8 * t + 6], acc[jn][im][8 * t + 7]);
          g4_swap32(x01, y01);
          g4_swap32(x23, y23);
          val = make_uint4(x01, x23, y01, y23);
        }
        if (STATS) {  // of the values as stored; rows beyond M are exact zeros
          float f[8];
          Vec<T>::unpack(val, f);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            ssum[jn][t][k] += f[k];
            ssq[jn][t][k] = fmaf(f[k], f[k], ssq[jn][t][k]);
          }
        }
        if (p < a.M && ook[jn][t]) stg16(Y + (long)p * a.ldy + o, val);
      }
    }
  }
  if (STATS) {
    // fold: column blocks jn = 0 / 1 into 16-lane rows of even / odd parity, then the 16 lanes;
    // the two row halves (wm) through LDS: red[wm][2][256] (the ring is idle: every wave passed
    // the main loop's closing barrier)
    float* red = reinterpret_cast<float*>(smem_raw);
    const int jrow = (lane >> 4) & 1;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float s = g4_fold(ssum[0][t][k], ssum[1][t][k]);
        const float q = g4_fold(ssq[0][t][k], ssq[1][t][k]);
        if ((lane & 15) == 0) {
          const int cl = wn * 64 + jrow * 32 + t * 16 + 8 * hh + k;
          red[(wm * 2 + 0) * 256 + cl] = s;
          red[(wm * 2 + 1) * 256 + cl] = q;
        }
      }
    __syncthreads();
    if (tid < 256) {
      const int oc = n0 + tid;
      if (oc < a.O) {
        float* dst = a.stat_partial + (long)tile_m * 2 * a.O;
        dst[oc] = red[0 * 256 + tid] + red[2 * 256 + tid];
        dst[a.O + oc] = red[1 * 256 + tid] + red[3 * 256 + tid];
      }
    }
  }
}

// ---- tile-row choice.  One launch = ceil(tiles / 256) rounds of (fixed cost + rows) on the
// 256 CUs (one 128 KiB block per CU): the row count with the smallest product wins; on a tie
// the larger tile (less operand traffic per output).  G4_FIXED_ROWS: launch + first DMA round
// trip + epilogue in units of tile rows (tools/lab/gemm_ab, r05).
#ifndef G4_FORCE_NA
#define G4_FORCE_NA 0
#endif
constexpr int G4_FIXED_ROWS = 96;

int g4_rows_per_tile(long M, int O) {
  if (G4_FORCE_NA) return 64 * G4_FORCE_NA;
  const long tn = (O + G4_BN - 1) / G4_BN;
  int best = 256;
  long best_cost = -1;
  for (int bm = 256; bm >= 192; bm -= 64) {
    const long tiles = ((M + bm - 1) / bm) * tn;
    const long cost = ((tiles + 255) / 256) * (bm + G4_FIXED_ROWS);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = bm;
    }
  }
  return best;
}

int g4_tiles_m(long M, int O) {
  const int bm = g4_rows_per_tile(M, O);
  return (int)((M + bm - 1) / bm);
}

template <int IMS, bool EP, bool STATS, bool KXK>
static int launch_g4_inst(const ConvGemmArgs& a, hipStream_t stream) {
  static const int once = [] {
    return (int)hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv_gemm_g4_kernel<IMS, EP, STATS, KXK>),
        hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
  }();
  if (once != 0) {
    set_error("conv_gemm_g4: cannot reserve %d bytes of LDS", G4_LDS_BYTES);
    return 2;
  }
  const dim3 grid(a.tiles_m * a.tiles_n), block(G4_THREADS);
  hipLaunchKernelGGL((conv_gemm_g4_kernel<IMS, EP, STATS, KXK>), grid, block, G4_LDS_BYTES, stream,
                     a);
  return check_launch(KXK ? "conv_gemm_fwd (g4 KxK)" : "conv_gemm_fwd (g4)");
}

template <bool EP, bool STATS, bool KXK>
static int launch_g4_na(const ConvGemmArgs& a, int bm, hipStream_t stream) {
  if (bm == 256) return launch_g4_inst<4, EP, STATS, KXK>(a, stream);
  return launch_g4_inst<3, EP, STATS, KXK>(a, stream);
}

// (forward convs take statistics, data gradients the folded-BN correction; never both)
bool conv_gemm_g4_usable(int dtype, const ConvGemmArgs& a) {
  return conv_gemm_glds_usable(dtype, a) && !(a.ep_x != nullptr && a.stat_partial != nullptr);
}

int launch_conv_gemm_g4(ConvGemmArgs a, hipStream_t stream) {
  const int bm = g4_rows_per_tile(a.M, a.O);
  a.tiles_m = (a.M + bm - 1) / bm;
  a.tiles_n = (a.O + G4_BN - 1) / G4_BN;
  if (a.ep_x != nullptr) return launch_g4_na<true, false, false>(a, bm, stream);
  if (a.stat_partial != nullptr) return launch_g4_na<false, true, false>(a, bm, stream);
  return launch_g4_na<false, false, false>(a, bm, stream);
}

int launch_conv_gemm_g4_kxk(ConvGemmArgs a, hipStream_t stream) {
  const int bm = g4_rows_per_tile(a.M, a.O);
  a.tiles_m = (a.M + bm - 1) / bm;
  a.tiles_n = (a.O + G4_BN - 1) / G4_BN;
  if (a.stat_partial != nullptr) return launch_g4_na<false, true, true>(a, bm, stream);
  return launch_g4_na<false, false, true>(a, bm, stream);
}

}  // namespace seg
