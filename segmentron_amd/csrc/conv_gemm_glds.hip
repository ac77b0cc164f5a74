// 1x1 / stride-1 convolution forward and data gradient on the direct-to-LDS GEMM core
// (gemm_glds.h): bf16, no element-wise prologue on the pixel operand.  Same argument block,
// epilogue features and BatchNorm-statistics format as conv_gemm_px256.hip (folded-BN
// backward correction y = acc - c0[o] - c1[o]*x[p][o], channel-slice output, ragged O, per
// pixel-tile (sum, sum of squares) rows taken from the values as stored).
//
// r05: the tile ROWS are chosen per launch — 256 or 192 (IMS = 4 / 3 row blocks per wave) — so
// that the tiles fill the 256 CUs in fewer or shorter rounds (glds_rows_per_tile); a statistics
// row then describes 256 OR 192 pixels and seg_conv_gemm_stat_rows tells the caller how many
// rows there are.  Measured (tools/lab/gemm_ab, profiles/r05_gemm_ab.md): 728 -> 1024 @16770
// pixels 54.1 -> 47.1 us, 1536 -> 2048 136.6 -> 124.2, dilated 3x3 256 -> 256 @66306 143.5 ->
// 130.5, 728 -> 728 @4290 22.7 -> 20.1; 728 -> 728 @16770 stays on 256 rows (198 tiles, one
// round; 192 rows would need two).
// Tried in r05 and dropped (same file, same harness; numbers in DESIGN.md section 3):
//  * a register-direct epilogue (v_permlane32_swap + v_permlane16_swap regroup the 32x32
//    accumulators into 16-byte vectors, 64 contiguous bytes per pixel and store, no LDS patch):
//    31.0 vs 30.9 us forward, 38.3 vs 33.1 us data gradient on 728 -> 728 — the ~130 lane
//    exchanges per wave cost what the LDS round trips did; with 32 contiguous bytes per pixel
//    (one stage) 44.7 us;
//  * the whole kernel on v_mfma_f32_16x16x32_bf16 (lane-linear fragment reads, 224-row tiles,
//    one-stage v_permlane16_swap epilogue): 27.0 vs 29.1 us on 728 -> 728 (225 tiles instead of
//    198, cheaper epilogue) but a 13-17 % slower main loop (74 vs 63 us on 1024 -> 1536, 151 vs
//    134 us on 1536 -> 2048);
//  * ping-pong wave groups (waves 4-7 one barrier interval behind waves 0-3; interval X = DMA
//    issue + all 12 fragment reads, interval Y = the slot's 16 MFMAs): 74.2 vs 68.4 us on
//    1024 -> 1536 — the load interval is longer than the MFMA interval, and the lockstep ring
//    already hides its fragment reads under its own MFMAs (~70 % MFMA duty in the loop).
#include "conv_gemm.h"
#include "conv_gemm_args.h"
#include "gemm_glds.h"

namespace seg {

// EP: folded-BatchNorm backward correction in the store path; STATS: BatchNorm partial sums
// KXK: stride-1 KxK convolution as an implicit GEMM (per-lane gather in the DMA source address,
// gemm_glds.h GlConvA) — ResNet bottleneck / PSP-head 3x3s, C % 32 == 0

// IMS: 32-pixel blocks per wave (4: 256-row tile, 3: 192-row tile)
template <bool EP, bool STATS, bool KXK = false, int IMS = 4>
__global__ __launch_bounds__(GL_THREADS, 2) void conv_gemm_glds_kernel(const ConvGemmArgs a) {
  typedef bf16_t T;
  constexpr int VEC = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lds_byte_t* lds = (lds_byte_t*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int L = xcd_remap(blockIdx.x, a.tiles_m * a.tiles_n);
  const int tile_m = L / a.tiles_n, tile_n = L - tile_m * a.tiles_n;
  const int m0 = tile_m * (64 * IMS), n0 = tile_n * GL_BN;

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
    gl_mainloop_ring<true, IMS>(A, B, a.K, m0, n0, lds, acc, &cg);
  } else {
    gl_mainloop_ring<false, IMS>(A, B, a.K, m0, n0, lds, acc);
  }

  // ---- epilogue, per wave and 32-pixel tile: channel groups -> LDS patch [32 px][64 ch] ->
  // 16-byte NHWC vectors (+ statistics, + folded-BatchNorm correction) -> global
  // (EP: the patch holds the fp32 accumulators — the folded-BatchNorm correction is applied to
  // them and the result rounded ONCE.  r02-r04 parked bf16 there and rounded twice: up to 50 bf16
  // ulps of error wherever the correction cancels the accumulator, tools/lab/gemm_ab r05.)
  typedef typename std::conditional<EP, float, T>::type PT;
  constexpr int EP_STRIDE = 64 * (int)sizeof(PT) + 16;
  constexpr int VPR = 64 / VEC;  // vectors per patch row
  unsigned char* ep = smem_raw + wave * 32 * EP_STRIDE;
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const int r32 = lane & 31, hh = lane >> 5;
  const int v = lane & (VPR - 1);         // this lane's vector column in every patch row
  const int o = n0 + wn * 64 + v * VEC;   // ... = these output channels
  float ssum[VEC], ssq[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) ssum[k] = ssq[k] = 0.f;
  float c0v[VEC], c1v[VEC];
  const bool epc = EP && o < a.O;
  if (epc) {
    load_params<VEC>(a.ep_c0, o, c0v);
    load_params<VEC>(a.ep_c1, o, c1v);
  }
  // r06, EP: the x vectors of ALL the wave's pixels are requested before the first patch is
  // written.  They were requested per 32-pixel block, at the top of the block's iteration, and
  // each iteration then sat out its own global round trip (~1.5 us under load, four times per
  // wave: most of what the data gradient's epilogue cost over the forward's).
  constexpr int NQ_ALL = (32 * VPR) / 64;
  uint4 xr_all[EP ? IMS : 1][EP ? NQ_ALL : 1];
  if (EP) {
#pragma unroll
    for (int im = 0; im < IMS; ++im)
#pragma unroll
      for (int q = 0; q < NQ_ALL; ++q) {
        const int r = (q * 64 + lane) / VPR;
        const int p = m0 + wm * 32 * IMS + im * 32 + r;
        const long pc = p < a.M ? p : a.M - 1;
        xr_all[im][q] = ldg16(reinterpret_cast<const T*>(a.ep_x) + pc * a.ldep + (epc ? o : 0));
      }
  }
#pragma unroll
  for (int im = 0; im < IMS; ++im) {
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = jn * 32 + 8 * g + 4 * hh;  // first of 4 consecutive channels
        float f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = acc[jn][im][4 * g + k];
        HVec<PT>::store(reinterpret_cast<PT*>(ep + r32 * EP_STRIDE) + ch, f);
      }
    }
    // the patch is private to this wave (in-order LDS): a compiler-level fence is enough
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // all four patch reads (and, EP, the four x loads) are issued before the first use: no
    // load sits behind a branch inside the loop — hipcc answers that with s_waitcnt vmcnt(0)
    // per iteration, which also waits for the previous iteration's STORE to be acknowledged
    // (16 serialized write round trips per wave: 6 of the 8 us this epilogue used to take)
    constexpr int NQ = (32 * VPR) / 64;
    uint4 val[NQ], xr[NQ], vhi[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int r = (q * 64 + lane) / VPR;
      if (EP) {  // 8 fp32 = two 16-byte pieces
        val[q] = *reinterpret_cast<const uint4*>(ep + r * EP_STRIDE + v * 32);
        vhi[q] = *reinterpret_cast<const uint4*>(ep + r * EP_STRIDE + v * 32 + 16);
      } else {
        val[q] = *reinterpret_cast<const uint4*>(ep + r * EP_STRIDE + v * 16);
      }
      if (EP) xr[q] = xr_all[im][q];
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int r = (q * 64 + lane) / VPR;
      const int p = m0 + wm * 32 * IMS + im * 32 + r;
      if (EP) {
        float f[VEC], xv[VEC];
        Vec<float>::unpack(val[q], f);
        Vec<float>::unpack(vhi[q], f + 4);
        Vec<T>::unpack(xr[q], xv);
        if (epc) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) f[k] = f[k] - c0v[k] - c1v[k] * xv[k];
        }
        val[q] = Vec<T>::pack(f);
      }
      // of the values as stored; rows beyond M are exact zeros (EP: corrected garbage — skipped)
      if (STATS && (!EP || p < a.M)) {
        float f[VEC];
        Vec<T>::unpack(val[q], f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          ssum[k] += f[k];
          ssq[k] = fmaf(f[k], f[k], ssq[k]);
        }
      }
      // (O % 8 == 0 on this kernel — conv_gemm_glds_usable: a vector is inside or outside)
      if (p < a.M && o < a.O) stg16(Y + (long)p * a.ldy + o, val[q]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (STATS) {
    // lanes sharing a vector column differ in lane bits >= log2(VPR): fold them, then the two
    // pixel halves through LDS: red[wm][2][256]
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
#pragma unroll
      for (int m = VPR; m < 64; m <<= 1) {
        ssum[k] += __shfl_xor(ssum[k], m, 64);
        ssq[k] += __shfl_xor(ssq[k], m, 64);
      }
    }
    __syncthreads();  // every wave is done with its patch: the region is reused below
    float* red = reinterpret_cast<float*>(smem_raw);
    if (lane < VPR) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        red[(wm * 2 + 0) * 256 + wn * 64 + v * VEC + k] = ssum[k];
        red[(wm * 2 + 1) * 256 + wn * 64 + v * VEC + k] = ssq[k];
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

bool conv_gemm_glds_usable(int dtype, const ConvGemmArgs& a) {
  return dtype == DT_BF16 && a.pro_mode == PRO_NONE && a.bias == nullptr && a.KH == 1 && a.KW == 1 && a.stride == 1 &&
         a.pad == 0 && a.tconv == 0 && a.out_s == 1 && (a.K % 8) == 0 && (a.ldx % 8) == 0 && (a.O % 8) == 0 &&
         (a.ldy % 8) == 0;
}

// ---- tile-row choice.  One launch = ceil(tiles / 256) rounds of (fixed cost + rows) on the
// 256 CUs (one 128 KiB block per CU): the row count with the smallest product wins; on a tie the
// larger tile (less operand traffic per output).  GLDS_FIXED_ROWS: launch + first DMA round trip
// + epilogue in units of tile rows; with 96 the rule picks the faster row count on 11 of the 12
// shapes of tools/lab/gemm_ab (the twelfth, 304 -> 256 @263682 pixels, 5 vs 6 rounds, by 3 %).
constexpr int GLDS_FIXED_ROWS = 96;

// r06: + 224 rows, on the four-wave generation (conv_gemm_glds4.hip: 1 x 4 waves of 224 x 64), for
// 1x1 launches of at most two rounds — where the tile count decides: 16770 pixels x 728 channels are
// 66 x 3 = 198 tiles of 256 rows (one round, 77 % of the CUs) or 75 x 3 = 225 of 224 (one round,
// 88 %, 7/8 of the MFMAs per wave); x 1536 channels 396 (two rounds) or 450 (two shorter rounds).
// In the captured step (tools/lab/prof_ab.sh + trace_ab.py, per layer, profiles/r06_gemm_w4.md):
// 728 -> 728 forward 29.96 -> 28.88 us, 1024 / 1536 -> 1536 82.1 -> 75.0 and 106.8 -> 97.6.  With
// many rounds the four-wave kernel is no better than this one (one wave per SIMD: slower ramp and
// epilogue; 304 -> 256 @263682 pixels: 85.2 -> 83.6 forward, 51.9 -> 57.1 / 76.4 -> 84.0 on the
// short-K data gradients), and neither are its 256- and 192-row forms (net zero over the exit
// flow) — they are not instantiated.
constexpr long GLDS_W4_MAX_TILES = 512;

static int glds_pick_rows(long M, int O, bool with224) {
  const long tn = (O + GL_BN - 1) / GL_BN;
  int best = 256;
  long best_cost = -1;
  for (int bm = 256; bm >= 192; bm -= 32) {
    const long tiles = ((M + bm - 1) / bm) * tn;
    if (bm == 224 && (!with224 || tiles > GLDS_W4_MAX_TILES)) continue;
    const long cost = ((tiles + 255) / 256) * (bm + GLDS_FIXED_ROWS);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = bm;
    }
  }
  return best;
}

int glds_rows_per_tile(long M, int O, bool kxk) { return glds_pick_rows(M, O, !kxk); }

int glds_tiles_m(long M, int O, bool kxk) {
  const int bm = glds_rows_per_tile(M, O, kxk);
  return (int)((M + bm - 1) / bm);
}

template <bool EP, bool STATS, bool KXK, int IMS>
static int launch_glds_inst(const ConvGemmArgs& a, hipStream_t stream) {
  static const int once = [] {
    return (int)hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv_gemm_glds_kernel<EP, STATS, KXK, IMS>),
        hipFuncAttributeMaxDynamicSharedMemorySize, GL_LDS_BYTES);
  }();
  if (once != 0) {
    set_error("conv_gemm_glds: cannot reserve %d bytes of LDS", GL_LDS_BYTES);
    return 2;
  }
  const dim3 grid(a.tiles_m * a.tiles_n), block(GL_THREADS);
  hipLaunchKernelGGL((conv_gemm_glds_kernel<EP, STATS, KXK, IMS>), grid, block, GL_LDS_BYTES,
                     stream, a);
  return check_launch(KXK ? "conv_gemm_fwd (glds KxK)" : "conv_gemm_fwd (glds)");
}

template <bool EP, bool STATS, bool KXK>
static int launch_glds_rows(ConvGemmArgs a, hipStream_t stream) {
  // The data gradient with the folded-BatchNorm correction stays on eight waves: its epilogue
  // (fp32 patch, unpack / correct / round) is instruction-bound and wants the second wave per SIMD
  // — 728 -> 728 in the step: 33.7 us on 256 rows / eight waves, 34.9 on 224 / four.  It writes no
  // statistics rows, so its tile height is its own business.
  const int bm = glds_pick_rows(a.M, a.O, !KXK && !(EP && !STATS));
  a.tiles_m = (a.M + bm - 1) / bm;
  a.tiles_n = (a.O + GL_BN - 1) / GL_BN;
  if constexpr (!KXK) {
    if (bm == 224) return launch_conv_gemm_glds4(a, bm, stream);
  }
  if (bm == 256) return launch_glds_inst<EP, STATS, KXK, 4>(a, stream);
  if (bm == 192) return launch_glds_inst<EP, STATS, KXK, 3>(a, stream);
  set_error("conv_gemm_glds: no %d-row tile on the eight-wave kernel", bm);
  return 2;
}

int launch_conv_gemm_glds(ConvGemmArgs a, hipStream_t stream) {
  // (forward convs take statistics, data gradients the folded-BN correction; never both)
  if (a.ep_x != nullptr && a.stat_partial != nullptr)
    return launch_glds_rows<true, true, false>(a, stream);
  if (a.ep_x != nullptr) return launch_glds_rows<true, false, false>(a, stream);
  if (a.stat_partial != nullptr) return launch_glds_rows<false, true, false>(a, stream);
  return launch_glds_rows<false, false, false>(a, stream);
}

// ---- stride-1 KxK on the same pipeline
bool conv_gemm_glds_kxk_usable(int dtype, const ConvGemmArgs& a) {
  return dtype == DT_BF16 && a.pro_mode == PRO_NONE && a.bias == nullptr && a.ep_x == nullptr &&
         a.KH * a.KW > 1 && a.stride == 1 && a.tconv == 0 && a.out_s == 1 && (a.C % 32) == 0 &&
         (a.ldx % 8) == 0 && (a.O % 8) == 0 && (a.ldy % 8) == 0 &&
         (long)a.N * a.Hi * a.Wi < (1L << 31);
}

int launch_conv_gemm_glds_kxk(ConvGemmArgs a, hipStream_t stream) {
  if (a.stat_partial != nullptr) return launch_glds_rows<false, true, true>(a, stream);
  return launch_glds_rows<false, false, true>(a, stream);
}

}  // namespace seg
