// 1x1 / stride-1 convolution forward and data gradient on the FOUR-wave direct-to-LDS GEMM core
// (gemm_glds4.h): 224 x 256 block tile, 1 x 4 waves of 224 x 64 (7 x 2 MFMA 32x32x16 tiles, 224
// fp32 accumulators in AGPRs, one wave per SIMD).  Same argument block, statistics rows and
// folded-BatchNorm correction as conv_gemm_glds.hip, whose launcher picks the generation per
// launch (launch_glds_rows / glds_rows_per_tile: this one for the 1x1 launches of at most two
// rounds, where 224 rows fill the 256 CUs better than 256 or 192).
//
// Epilogue, per wave: a UNIT = one 32-pixel block x 64 channels = one [32 px][64 ch] patch through
// LDS -> 16-byte NHWC vectors.  The wave is alone on its SIMD, so units are taken two at a time on
// TWO patches: both are written, then both are read and stored — the second unit's LDS round trip
// runs under the first one's arithmetic and stores.
#include "conv_gemm.h"
#include "conv_gemm_args.h"
#include "gemm_glds4.h"

namespace seg {

template <bool EP, bool STATS, int WM, int IM, int JN>
__global__ __launch_bounds__(GL4_THREADS, 1) void conv_gemm_glds4_kernel(const ConvGemmArgs a) {
  typedef bf16_t T;
  constexpr int VEC = 8;
  constexpr int WN = 4 / WM;
  constexpr int BM_ROWS = WM * IM * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lds_byte_t* lds = (lds_byte_t*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int L = xcd_remap(blockIdx.x, a.tiles_m * a.tiles_n);
  const int tile_m = L / a.tiles_n, tile_n = L - tile_m * a.tiles_n;
  const int m0 = tile_m * BM_ROWS, n0 = tile_n * GL_BN;

  GemmOperand A, B;
  A.base = reinterpret_cast<const unsigned char*>(a.x);
  A.ld_bytes = a.ldx * 2;
  A.rows = a.M;
  B.base = reinterpret_cast<const unsigned char*>(a.w);
  B.ld_bytes = (long)a.K * 2;
  B.rows = a.O;

  f32x16 acc[JN][IM];  // (the ring starts from a constant-zero accumulator INPUT)
  gl4_mainloop<WM, IM, JN>(A, B, a.K, m0, n0, lds, acc);

  // ---- epilogue.  A UNIT = one 32-pixel block x one 64-channel half of the wave's tile =
  // one [32 px][64 ch] patch through LDS; units are taken two at a time (two patches).
  // (EP: the patch holds the fp32 accumulators — the folded-BatchNorm correction is applied to
  // them and the result rounded ONCE, as in conv_gemm_glds.hip)
  typedef typename std::conditional<EP, float, T>::type PT;
  constexpr int EP_STRIDE = 64 * (int)sizeof(PT) + 16;
  constexpr int VPR = 64 / VEC;           // vectors per patch row
  constexpr int NQ = (32 * VPR) / 64;     // vectors per lane and patch
  constexpr int PATCH = 32 * EP_STRIDE;
  constexpr int NH = JN / 2;              // 64-channel halves per wave
  constexpr int NU = IM * NH;             // units per wave
  unsigned char* ep = smem_raw + wave * 2 * PATCH;
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const int r32 = lane & 31, hh = lane >> 5;
  const int v = lane & (VPR - 1);                   // this lane's vector column in every patch row
  const int ob = n0 + wn * 32 * JN + v * VEC;       // ... = these output channels (+ 64 * half)
  float ssum[NH][VEC], ssq[NH][VEC];
#pragma unroll
  for (int jp = 0; jp < NH; ++jp)
#pragma unroll
    for (int k = 0; k < VEC; ++k) ssum[jp][k] = ssq[jp][k] = 0.f;
  float c0v[NH][VEC], c1v[NH][VEC];
  bool epc[NH];
#pragma unroll
  for (int jp = 0; jp < NH; ++jp) {
    epc[jp] = EP && ob + jp * 64 < a.O;
    if (epc[jp]) {
      load_params<VEC>(a.ep_c0, ob + jp * 64, c0v[jp]);
      load_params<VEC>(a.ep_c1, ob + jp * 64, c1v[jp]);
    }
  }
  // EP: the x vectors (forward activations saved ~10 ms ago: an HBM round trip, ~2 us under load)
  // of ALL the wave's units are requested before the first patch is written where the registers
  // allow it (XALL: 7 units = 112 registers beside 224 accumulators), else one pair of units ahead
  constexpr bool XALL = EP && NU <= 7;
  constexpr int XB = XALL ? (NU + 1) / 2 : 2;  // buffers of one pair of units
  uint4 xr[EP ? XB : 1][EP ? 2 : 1][EP ? NQ : 1];
  auto request_x = [&](int u0, int buf) {
#pragma unroll
    for (int du = 0; du < 2; ++du) {
      const int u = u0 + du;
      if (u < NU) {
        const int im = u / NH, jp = u % NH;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int r = (q * 64 + lane) / VPR;
          const int p = m0 + wm * 32 * IM + im * 32 + r;
          const long pc = p < a.M ? p : a.M - 1;
          xr[buf][du][q] = ldg16(reinterpret_cast<const T*>(a.ep_x) + pc * a.ldep +
                                 (epc[jp] ? ob + jp * 64 : 0));
        }
      }
    }
  };
  if (EP) {
    if (XALL) {
#pragma unroll
      for (int u0 = 0; u0 < NU; u0 += 2) request_x(u0, u0 >> 1);
    } else {
      request_x(0, 0);
    }
  }
#pragma unroll
  for (int u0 = 0; u0 < NU; u0 += 2) {
    if (EP && !XALL && u0 + 2 < NU) request_x(u0 + 2, ((u0 >> 1) + 1) & 1);
#pragma unroll
    for (int du = 0; du < 2; ++du) {
      const int u = u0 + du;
      if (u < NU) {
        const int im = u / NH, jp = u % NH;
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int ch = j2 * 32 + 8 * g + 4 * hh;  // first of 4 consecutive channels
            float f[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) f[k] = acc[jp * 2 + j2][im][4 * g + k];
            HVec<PT>::store(reinterpret_cast<PT*>(ep + du * PATCH + r32 * EP_STRIDE) + ch, f);
          }
        }
      }
    }
    // the patches are private to this wave (in-order LDS): a compiler-level fence is enough
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint4 val[2][NQ], vhi[2][EP ? NQ : 1];
#pragma unroll
    for (int du = 0; du < 2; ++du)
      if (u0 + du < NU) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int r = (q * 64 + lane) / VPR;
          const unsigned char* src = ep + du * PATCH + r * EP_STRIDE;
          if (EP) {  // 8 fp32 = two 16-byte pieces
            val[du][q] = *reinterpret_cast<const uint4*>(src + v * 32);
            vhi[du][q] = *reinterpret_cast<const uint4*>(src + v * 32 + 16);
          } else {
            val[du][q] = *reinterpret_cast<const uint4*>(src + v * 16);
          }
        }
      }
#pragma unroll
    for (int du = 0; du < 2; ++du) {
      const int u = u0 + du;
      if (u < NU) {
        const int im = u / NH, jp = u % NH;
        const int o = ob + jp * 64;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int r = (q * 64 + lane) / VPR;
          const int p = m0 + wm * 32 * IM + im * 32 + r;
          uint4 out = val[du][q];
          if (EP) {
            float f[VEC], xv[VEC];
            Vec<float>::unpack(val[du][q], f);
            Vec<float>::unpack(vhi[du][q], f + 4);
            Vec<T>::unpack(xr[XALL ? (u0 >> 1) : ((u0 >> 1) & 1)][du][q], xv);
            if (epc[jp]) {
#pragma unroll
              for (int k = 0; k < VEC; ++k) f[k] = f[k] - c0v[jp][k] - c1v[jp][k] * xv[k];
            }
            out = Vec<T>::pack(f);
          }
          // of the values as stored; rows beyond M are exact zeros (EP: corrected garbage — skipped)
          if (STATS && (!EP || p < a.M)) {
            float f[VEC];
            Vec<T>::unpack(out, f);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
              ssum[jp][k] += f[k];
              ssq[jp][k] = fmaf(f[k], f[k], ssq[jp][k]);
            }
          }
          // (O % 8 == 0 on this kernel — conv_gemm_glds_usable: a vector is inside or outside)
          if (p < a.M && o < a.O) stg16(Y + (long)p * a.ldy + o, out);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (STATS) {
    // lanes sharing a vector column differ in lane bits >= log2(VPR): fold them, then the WM
    // pixel groups through LDS: red[WM][2][256]
#pragma unroll
    for (int jp = 0; jp < NH; ++jp)
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
#pragma unroll
        for (int m = VPR; m < 64; m <<= 1) {
          ssum[jp][k] += __shfl_xor(ssum[jp][k], m, 64);
          ssq[jp][k] += __shfl_xor(ssq[jp][k], m, 64);
        }
      }
    __syncthreads();  // every wave is done with its patches: the region is reused below
    float* red = reinterpret_cast<float*>(smem_raw);
    if (lane < VPR) {
#pragma unroll
      for (int jp = 0; jp < NH; ++jp)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          red[(wm * 2 + 0) * 256 + wn * 32 * JN + jp * 64 + v * VEC + k] = ssum[jp][k];
          red[(wm * 2 + 1) * 256 + wn * 32 * JN + jp * 64 + v * VEC + k] = ssq[jp][k];
        }
    }
    __syncthreads();
    const int oc = n0 + tid;
    if (oc < a.O) {
      float* dst = a.stat_partial + (long)tile_m * 2 * a.O;
      float s1 = red[tid], s2 = red[256 + tid];
      if (WM == 2) {
        s1 += red[2 * 256 + tid];
        s2 += red[3 * 256 + tid];
      }
      dst[oc] = s1;
      dst[a.O + oc] = s2;
    }
  }
}

template <bool EP, bool STATS, int WM, int IM, int JN>
static int launch_glds4_inst(const ConvGemmArgs& a, hipStream_t stream) {
  static const int once = [] {
    return (int)hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv_gemm_glds4_kernel<EP, STATS, WM, IM, JN>),
        hipFuncAttributeMaxDynamicSharedMemorySize, GL4_LDS_BYTES);
  }();
  if (once != 0) {
    set_error("conv_gemm_glds4: cannot reserve %d bytes of LDS", GL4_LDS_BYTES);
    return 2;
  }
  const dim3 grid(a.tiles_m * a.tiles_n), block(GL4_THREADS);
  hipLaunchKernelGGL((conv_gemm_glds4_kernel<EP, STATS, WM, IM, JN>), grid, block,
                     GL4_LDS_BYTES, stream, a);
  return check_launch("conv_gemm_fwd (glds4)");
}

// Only the 224-row form is instantiated: the 2 x 2 waves of 4 x 4 (256 rows) and 3 x 4 (192 rows)
// compile and were bit-identical on tools/lab/gemm_ab, but bought nothing in the captured step
// (profiles/r06_gemm_w4.md) over the eight-wave kernel of conv_gemm_glds.hip.
template <bool EP, bool STATS>
static int launch_glds4_rows(const ConvGemmArgs& a, int rows, hipStream_t stream) {
  if (rows == 224) return launch_glds4_inst<EP, STATS, 1, 7, 2>(a, stream);
  set_error("conv_gemm_glds4: no %d-row tile", rows);
  return 2;
}

// (tiles_m / tiles_n set by the caller for `rows` = 224 rows per tile)
int launch_conv_gemm_glds4(const ConvGemmArgs& a, int rows, hipStream_t stream) {
  const bool ep = a.ep_x != nullptr, st = a.stat_partial != nullptr;
  if (ep && !st) {  // (launch_glds_rows keeps the plain data gradient on the eight-wave kernel)
    set_error("conv_gemm_glds4: the correction-only epilogue is not instantiated");
    return 2;
  }
  if (ep) return launch_glds4_rows<true, true>(a, rows, stream);
  return st ? launch_glds4_rows<false, true>(a, rows, stream)
            : launch_glds4_rows<false, false>(a, rows, stream);
}

}  // namespace seg
