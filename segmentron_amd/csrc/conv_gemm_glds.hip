// 1x1 / stride-1 convolution forward and data gradient on the direct-to-LDS GEMM core
// (gemm_glds.h): bf16, no element-wise prologue on the pixel operand.  Same argument block,
// epilogue features and BatchNorm-statistics format as conv_gemm_px256.hip (folded-BN
// backward correction y = acc - c0[o] - c1[o]*x[p][o], channel-slice output, ragged O, per
// 256-pixel-tile (sum, sum of squares) rows taken from the values as stored).
#include "conv_gemm.h"
#include "conv_gemm_args.h"
#include "gemm_glds.h"

namespace seg {

// EP: folded-BatchNorm backward correction in the store path; STATS: BatchNorm partial sums
// KXK: stride-1 KxK convolution as an implicit GEMM (per-lane gather in the DMA source address,
// gemm_glds.h GlConvA) — ResNet bottleneck / PSP-head 3x3s, C % 32 == 0
template <bool EP, bool STATS, bool KXK = false>
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
  const int m0 = tile_m * GL_BM, n0 = tile_n * GL_BN;

  GemmOperand A, B;
  A.base = reinterpret_cast<const unsigned char*>(a.x);
  A.ld_bytes = a.ldx * 2;
  A.rows = a.M;
  B.base = reinterpret_cast<const unsigned char*>(a.w);
  B.ld_bytes = (long)a.K * 2;
  B.rows = a.O;

  f32x16 acc[2][4];  // (the ring starts from a constant-zero accumulator INPUT)

  if (KXK) {
    const GlConvA cg = {a.M, a.Hi, a.Wi, a.Ho, a.Wo, a.KW, a.pad, a.dil, a.C / 32};
    gl_mainloop_ring<true>(A, B, a.K, m0, n0, lds, acc, &cg);
  } else {
    gl_mainloop_ring<false>(A, B, a.K, m0, n0, lds, acc);
  }

  // ---- epilogue, per wave and 32-pixel tile: channel groups -> LDS patch [32 px][64 ch] ->
  // 16-byte NHWC vectors (+ statistics, + folded-BatchNorm correction) -> global
  constexpr int EP_STRIDE = 64 * (int)sizeof(T) + 16;
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
#pragma unroll
  for (int im = 0; im < 4; ++im) {
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = jn * 32 + 8 * g + 4 * hh;  // first of 4 consecutive channels
        float f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = acc[jn][im][4 * g + k];
        HVec<T>::store(reinterpret_cast<T*>(ep + r32 * EP_STRIDE) + ch, f);
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
    uint4 val[NQ], xr[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int r = (q * 64 + lane) / VPR;
      val[q] = *reinterpret_cast<const uint4*>(ep + r * EP_STRIDE + v * 16);
      if (EP) {
        const int p = m0 + wm * 128 + im * 32 + r;
        const long pc = p < a.M ? p : a.M - 1;
        xr[q] = ldg16(reinterpret_cast<const T*>(a.ep_x) + pc * a.ldep + (epc ? o : 0));
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int r = (q * 64 + lane) / VPR;
      const int p = m0 + wm * 128 + im * 32 + r;
      if (STATS) {  // rows beyond M are exact zeros
        float f[VEC];
        Vec<T>::unpack(val[q], f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          ssum[k] += f[k];
          ssq[k] = fmaf(f[k], f[k], ssq[k]);
        }
      }
      if (EP && epc) {
        float f[VEC], xv[VEC];
        Vec<T>::unpack(val[q], f);
        Vec<T>::unpack(xr[q], xv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) f[k] = f[k] - c0v[k] - c1v[k] * xv[k];
        val[q] = Vec<T>::pack(f);
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

template <bool EP, bool STATS>
static int launch_glds_inst(const ConvGemmArgs& a, hipStream_t stream) {
  static const int once = [] {
    return (int)hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv_gemm_glds_kernel<EP, STATS>),
        hipFuncAttributeMaxDynamicSharedMemorySize, GL_LDS_BYTES);
  }();
  if (once != 0) {
    set_error("conv_gemm_glds: cannot reserve %d bytes of LDS", GL_LDS_BYTES);
    return 2;
  }
  const dim3 grid(a.tiles_m * a.tiles_n), block(GL_THREADS);
  hipLaunchKernelGGL((conv_gemm_glds_kernel<EP, STATS>), grid, block, GL_LDS_BYTES,
                     stream, a);
  return check_launch("conv_gemm_fwd (glds)");
}

int launch_conv_gemm_glds(ConvGemmArgs a, hipStream_t stream) {
  a.tiles_m = px256_tiles_m(a.M);  // 256-pixel tiles: same statistics rows as the px256 kernel
  a.tiles_n = (a.O + GL_BN - 1) / GL_BN;
  // (forward convs take statistics, data gradients the folded-BN correction; never both)
  if (a.ep_x != nullptr && a.stat_partial != nullptr)
    return launch_glds_inst<true, true>(a, stream);
  if (a.ep_x != nullptr) return launch_glds_inst<true, false>(a, stream);
  if (a.stat_partial != nullptr) return launch_glds_inst<false, true>(a, stream);
  return launch_glds_inst<false, false>(a, stream);
}

// ---- stride-1 KxK on the same pipeline
bool conv_gemm_glds_kxk_usable(int dtype, const ConvGemmArgs& a) {
  return dtype == DT_BF16 && a.pro_mode == PRO_NONE && a.bias == nullptr && a.ep_x == nullptr &&
         a.KH * a.KW > 1 && a.stride == 1 && a.tconv == 0 && a.out_s == 1 && (a.C % 32) == 0 &&
         (a.ldx % 8) == 0 && (a.O % 8) == 0 && (a.ldy % 8) == 0 &&
         (long)a.N * a.Hi * a.Wi < (1L << 31);
}

template <bool STATS>
static int launch_glds_kxk_inst(const ConvGemmArgs& a, hipStream_t stream) {
  static const int once = [] {
    return (int)hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv_gemm_glds_kernel<false, STATS, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, GL_LDS_BYTES);
  }();
  if (once != 0) {
    set_error("conv_gemm_glds (KxK): cannot reserve %d bytes of LDS", GL_LDS_BYTES);
    return 2;
  }
  const dim3 grid(a.tiles_m * a.tiles_n), block(GL_THREADS);
  hipLaunchKernelGGL((conv_gemm_glds_kernel<false, STATS, true>), grid, block, GL_LDS_BYTES,
                     stream, a);
  return check_launch("conv_gemm_fwd (glds KxK)");
}

int launch_conv_gemm_glds_kxk(ConvGemmArgs a, hipStream_t stream) {
  a.tiles_m = px256_tiles_m(a.M);
  a.tiles_n = (a.O + GL_BN - 1) / GL_BN;
  if (a.stat_partial != nullptr) return launch_glds_kxk_inst<true>(a, stream);
  return launch_glds_kxk_inst<false>(a, stream);
}


}  // namespace seg
