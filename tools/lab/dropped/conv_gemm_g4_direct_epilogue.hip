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
// 32-channel block.  Two register-exchange stages (v_permlane32_swap, then v_permlane16_swap; one
// instruction per register, no LDS) regroup a block so that four lanes hold the four consecutive
// 8-channel vectors of one pixel: 16-byte NHWC stores, 64 contiguous bytes per pixel and
// instruction, no LDS patch and no LDS round trip between the accumulators and the stores
// (r02-r04: 8-byte pieces -> LDS patch -> 16-byte vectors, ~7 us of a 30 us launch; the
// data-gradient variant also rounded twice there — up to 50 bf16 ulps of error where the
// correction cancels the accumulator, tools/lab/gemm_ab).  The statistics are lane-local sums
// over the wave's pixel blocks, folded over the 16 pixel lanes of a row with four DPP adds.
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

__device__ __forceinline__ void g4_swap16(uint32_t& x, uint32_t& y) {
  const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  x = r[0];
  y = r[1];
}

// sum over the 16 lanes of a DPP row (every lane of the row ends up with the total)
__device__ __forceinline__ float g4_row_sum(float v) {
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xF, 0xF, true)); // row_half_mirror
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xF, 0xF, true)); // row_mirror
  return v;
}

#ifdef LAB_TICKET
__device__ unsigned g_lab_g4_ticket[64];
__device__ float g_lab_g4_sink[1024];
__device__ __forceinline__ void lab_g4_last_arriver(const float* base, long row_pitch, int rows,
                                                    int cols, long sub_pitch, unsigned* ticket,
                                                    int expected) {
  // (the flag lives in the ring's LDS, free by now: a second __shared__ object would cost the
  // main loop a vmcnt(0) per k-step — cdna_hip_programming.md)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* s_last = reinterpret_cast<int*>(smem_raw + 8192);
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = (t % (unsigned)expected) == (unsigned)expected - 1u;
    if (*s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!*s_last) return;
  for (int e = threadIdx.x; e < cols * 2; e += blockDim.x) {
    const int sub = e / cols, c = e - sub * cols;
    float tot = 0.f;
    for (int r = 0; r < rows; ++r) tot += base[(long)r * row_pitch + sub * sub_pitch + c];
    g_lab_g4_sink[e & 1023] = tot;
  }
}
#endif

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

  // ---- epilogue.  Stage 1 (v_permlane32_swap) gives lane (px, hh) the vectors V[t] = channels
  // 16t + 8hh .. +7 of pixel px; stage 2 (v_permlane16_swap of V[0] with V[1]) regroups them so
  // that the four lanes {lp, lp+16, lp+32, lp+48} hold the four consecutive vectors of ONE pixel:
  //   P[0]: pixel lp,      P[1]: pixel 16 + lp      (lp = lane & 15, 16-lane row r = lane >> 4)
  //   channels jn*32 + 16*(r & 1) + 8*(r >> 1) .. +7
  // -> a store instruction writes 64 contiguous bytes per pixel.  (Measured r05, gemm_ab: with
  // stage 1 alone — 32 contiguous bytes per pixel and instruction — the stores and above all the
  // ep_x loads of the data-gradient variant were slower than the LDS patch they replace: 44.7
  // vs 31.0 us on 728 -> 728.)
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const int lp = lane & 15, r4 = lane >> 4;
  const int o0 = n0 + wn * 64 + 16 * (r4 & 1) + 8 * (r4 >> 1);  // + jn * 32
  const bool ook[2] = {o0 < a.O, o0 + 32 < a.O};  // (O % 8 == 0: a vector is inside or outside)
  float ssum[2][8], ssq[2][8];
  if (STATS) {
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int k = 0; k < 8; ++k) ssum[jn][k] = ssq[jn][k] = 0.f;
  }
#pragma unroll
  for (int jn = 0; jn < 2; ++jn) {
    const int o = o0 + jn * 32;
    const int oc = ook[jn] ? o : 0;
    float c0v[8], c1v[8];
    uint4 xr[IMS][2];
    if (EP) {  // this column block's ep_x vectors of all pixel blocks are requested together
      load_params<8>(a.ep_c0, oc, c0v);
      load_params<8>(a.ep_c1, oc, c1v);
#pragma unroll
      for (int im = 0; im < IMS; ++im)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int p = m0 + (wm * IMS + im) * 32 + 16 * u + lp;
          const long pc = p < a.M ? p : a.M - 1;
          xr[im][u] = ldg16(reinterpret_cast<const T*>(a.ep_x) + pc * a.ldep + oc);
        }
    }
#pragma unroll
    for (int im = 0; im < IMS; ++im) {
      uint4 val[2];
      if (EP) {  // both swap stages in fp32, correct, round ONCE
        uint32_t v[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[t][k] = __float_as_uint(acc[jn][im][8 * t + k]);
            v[t][4 + k] = __float_as_uint(acc[jn][im][8 * t + 4 + k]);
            g4_swap32(v[t][k], v[t][4 + k]);
          }
#pragma unroll
        for (int k = 0; k < 8; ++k) g4_swap16(v[0][k], v[1][k]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float f[8], xv[8];
          Vec<T>::unpack(xr[im][u], xv);
#pragma unroll
          for (int k = 0; k < 8; ++k)
            f[k] = __uint_as_float(v[u][k]) - c0v[k] - c1v[k] * xv[k];
          val[u] = Vec<T>::pack(f);
        }
      } else {
        uint32_t v[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          v[t][0] = pack_bf16x2(acc[jn][im][8 * t + 0], acc[jn][im][8 * t + 1]);
          v[t][1] = pack_bf16x2(acc[jn][im][8 * t + 2], acc[jn][im][8 * t + 3]);
          v[t][2] = pack_bf16x2(acc[jn][im][8 * t + 4], acc[jn][im][8 * t + 5]);
          v[t][3] = pack_bf16x2(acc[jn][im][8 * t + 6], acc[jn][im][8 * t + 7]);
          g4_swap32(v[t][0], v[t][2]);
          g4_swap32(v[t][1], v[t][3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) g4_swap16(v[0][k], v[1][k]);
#pragma unroll
        for (int u = 0; u < 2; ++u) val[u] = make_uint4(v[u][0], v[u][1], v[u][2], v[u][3]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int p = m0 + (wm * IMS + im) * 32 + 16 * u + lp;
        if (STATS) {  // of the values as stored; rows beyond M are exact zeros
          float f[8];
          Vec<T>::unpack(val[u], f);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            ssum[jn][k] += f[k];
            ssq[jn][k] = fmaf(f[k], f[k], ssq[jn][k]);
          }
        }
        if (p < a.M && ook[jn]) stg16(Y + (long)p * a.ldy + o, val[u]);
      }
    }
  }
  if (STATS) {
    // the 16 pixel lanes of a row by four DPP adds per value; the two row halves (wm) through
    // LDS: red[wm][2][256] (the ring is idle: every wave passed the main loop's closing barrier)
    float* red = reinterpret_cast<float*>(smem_raw);
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float s = g4_row_sum(ssum[jn][k]);
        const float q = g4_row_sum(ssq[jn][k]);
        if (lp == 0) {
          const int cl = o0 - n0 + jn * 32 + k;
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
#ifdef LAB_TICKET
    // LAB ONLY: the cost of a last-arriver BatchNorm finalize behind this kernel (one release +
    // ticket per block; the last block of a column tile re-reads that tile's statistic rows)
    lab_g4_last_arriver(a.stat_partial + n0, 2 * a.O, a.tiles_m, min(256, a.O - n0), a.O,
                        &g_lab_g4_ticket[tile_n & 63], a.tiles_m);
#endif
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
