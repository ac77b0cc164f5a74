// Implicit-GEMM convolution, forward orientation (also used for dgrad with transposed weights):
//   Y[p, o] = sum_{kh,kw,c} act(X[n, ho*s - pad + kh*d, wo*s - pad + kw*d, c]) * W[o, kh, kw, c]
// with p = (n*Ho + ho)*Wo + wo, NHWC activations, packed weights [O][KH*KW*C] (k contiguous).
// Replaces nn.Conv2d for 1x1 (any stride) and dense KxK convs — the reference call sites are
// segmentron/modules/basic.py:42 (pointwise), :69 (_ConvBNReLU), segmentron/modules/module.py:45,57
// (ASPP), segmentron/models/backbones/xception.py:21,81 (shortcut / conv2),
// segmentron/models/deeplabv3_plus.py:64 (classifier).
//
// Fusions (the net is HBM-bound unless BN/ReLU ride along with the convs):
//   prologue : the *producer's* BatchNorm (+ReLU) applied to X while staging it into LDS
//              (scale/shift per input channel), zero padding applied after the activation
//   epilogue : + bias; per-channel sum / sum-of-squares partials of the fp32 accumulators for
//              the BatchNorm that follows (one row of partials per M-tile, reduced in fp64 by
//              bn_finalize — deterministic, no atomics); output written into a channel slice
//              of a wider buffer (ldy) so torch.cat never copies; optional strided row scatter
//              (dgrad of a stride-2 1x1 conv).
#include "conv_gemm.h"
#include "conv_gemm_args.h"

namespace seg {


// FAST : 1x1, stride 1, no padding — the 96 %-of-FLOPs case: operand rows are addressed by
//        pointers set up once and advanced by a constant per K-slab (no per-slab index math).
// DBUF : two LDS stages, one barrier per slab (2 blocks/CU) vs one stage, two barriers per slab
//        (3 blocks/CU).  The single-stage variant is the one launched (measured faster on every C3 shape).
// WIDE : block tile 256 pixels x 64 output channels (waves 4x1) instead of 128 x 128 (2x2): KxK
//        convolutions with O <= 64 at large spatial sizes (network stems, the ResNet layer1
//        3x3s) would otherwise spend half of their MFMAs and B staging on zero columns.
template <typename T, bool FAST, bool DBUF, bool WIDE = false>
__global__ __launch_bounds__(GEMM_THREADS, (DBUF || !FAST) ? 2 : 3) void conv_gemm_fwd_kernel(
    const ConvGemmArgs a) {
  static_assert(!WIDE || (!FAST && !DBUF), "the wide tile exists for the general path only");
  constexpr int VEC = Vec<T>::N;
  constexpr int BK = ROW_BYTES / (int)sizeof(T);
  constexpr int NSTAGE = DBUF ? 2 : 1;
  constexpr int TBM = WIDE ? 256 : BM, TBN = WIDE ? 64 : BN;   // block tile
  constexpr int RA = TBM / 32, RB = TBN / 32;                  // staged rows per thread
  constexpr int WMS = WIDE ? 4 : 2;                            // waves along M
  constexpr int A_BYTES = TBM * ROW_STRIDE, B_BYTES = TBN * ROW_STRIDE;
  static_assert(A_BYTES + B_BYTES == 2 * TILE_BYTES || WIDE, "stage layout");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * (A_BYTES + B_BYTES)];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = WIDE ? wave : wave >> 1, wn = WIDE ? 0 : wave & 1;
  const int L = xcd_remap(blockIdx.x, a.tiles_m * a.tiles_n);
  const int tile_m = L / a.tiles_n, tile_n = L - tile_m * a.tiles_n;
  const int m0 = tile_m * TBM, n0 = tile_n * TBN;

  // ---- staging assignment: thread -> vector column vc (16 B) of rows rb + 32*j
  const int vc = tid & 7, rb = tid >> 3;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ W = reinterpret_cast<const T*>(a.w);

  // general path state
  long a_base[RA];
  int a_hi0[RA], a_wi0[RA];
  // fast path state: row pointers (advanced by the slab offset) and validity
  const T* pa[RA];
  const T* pb[RB];
  unsigned row_ok = 0, col_ok = 0;
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int o = n0 + rb + 32 * j;
    pb[j] = W + (long)(o < a.O ? o : 0) * a.K;
    if (o < a.O) col_ok |= 1u << j;
  }
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int p = m0 + rb + 32 * j;
    if (FAST) {
      pa[j] = X + (long)(p < a.M ? p : 0) * a.ldx;
      if (p < a.M) row_ok |= 1u << j;
      a_base[j] = 0; a_hi0[j] = 0; a_wi0[j] = 0;
    } else {
      pa[j] = X;
      if (p < a.M) {
        const int wo = p % a.Wo;
        const int t = p / a.Wo;
        const int ho = t % a.Ho;
        const int n = t / a.Ho;
        a_base[j] = (long)n * a.Hi * a.Wi;
        a_hi0[j] = a.tconv ? ho + a.pad : ho * a.stride - a.pad;
        a_wi0[j] = a.tconv ? wo + a.pad : wo * a.stride - a.pad;
      } else {
        a_base[j] = 0;
        a_hi0[j] = -(1 << 28);  // never in range
        a_wi0[j] = -(1 << 28);
      }
    }
  }
  const bool single_tap = (a.KH * a.KW == 1);

  uint4 ra[RA], rbv[RB];
  unsigned a_ok_mask = 0;  // bit j: row j of the staged A slab is a real (in-bounds) pixel
  unsigned b_ok_mask = 0;  // bit j: row j of the staged B slab is a real output channel / k
  // BatchNorm prologue parameters of this thread's channel vector in the slab being staged
  // (fetched with the slab's data, so the LDS staging never waits on its own loads)
  float ps[VEC], pt[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { ps[i] = 1.f; pt[i] = 0.f; }
  const bool affine = (a.pro_mode & PRO_AFFINE) != 0;
  // Every load below is UNCONDITIONAL (out-of-range vectors read element 0 of the operand and
  // are replaced by zeros afterwards): a load under a per-lane branch makes hipcc wait for it
  // inside the branch, which serialised the 8 loads of a slab.
  auto load_slab = [&](int kt) {
    const int kv = kt * BK + vc * VEC;
    const bool kok = kv < a.K;
    int cur_c;
    if (FAST) {
      cur_c = kv;
      a_ok_mask = kok ? row_ok : 0u;
      const int koff = kok ? kv : 0;  // (a k range that does not exist reads the row start)
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        ra[j] = ldg16(pa[j] + koff);  // (masked when staged)
      }
    } else {
      int c = kv, dh = 0, dw = 0;
      if (!single_tap) {
        const int kidx = kv / a.C;
        c = kv - kidx * a.C;
        const int kh = kidx / a.KW;
        dh = kh * a.dil;
        dw = (kidx - kh * a.KW) * a.dil;
      }
      cur_c = c;
      a_ok_mask = 0;
      long off[RA];  // (all addresses first, then the loads back to back)
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        int hi = a_hi0[j] + dh, wi = a_wi0[j] + dw;
        bool ok = kok;
        if (a.tconv) {
          // dx[h,w] += dy[(h + pad - kh*dil)/s, (w + pad - kw*dil)/s] * W[.,kh,kw,.] when the
          // divisions are exact
          hi = a_hi0[j] - dh;
          wi = a_wi0[j] - dw;
          ok = ok && hi >= 0 && wi >= 0 && (hi % a.stride) == 0 && (wi % a.stride) == 0;
          hi /= a.stride;
          wi /= a.stride;
        }
        ok = ok && hi >= 0 && hi < a.Hi && wi >= 0 && wi < a.Wi;
        off[j] = ok ? (a_base[j] + (long)hi * a.Wi + wi) * a.ldx + c : 0;
        a_ok_mask |= ok ? (1u << j) : 0u;
      }
#pragma unroll
      for (int j = 0; j < RA; ++j) ra[j] = ldg16(X + off[j]);  // (masked when staged)
    }
    {
      const int koff = kok ? kv : 0;
#pragma unroll
      for (int j = 0; j < RB; ++j) rbv[j] = ldg16(pb[j] + koff);
      b_ok_mask = kok ? col_ok : 0u;
    }
    if (affine) {  // (uniform branch) channels beyond C only occur with kok == false
      const int cc = kok ? cur_c : 0;
      load_params<VEC>(a.pro_scale, cc, ps);
      load_params<VEC>(a.pro_shift, cc, pt);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (a.K + BK - 1) / BK;
  auto stage = [&](int buf) {
    // registers -> LDS (fused BN/ReLU prologue on the activation operand; padding stays zero)
    unsigned char* sA = smem + (DBUF ? buf : 0) * (A_BYTES + B_BYTES);
    unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int j = 0; j < RB; ++j)
      *reinterpret_cast<uint4*>(sB + (rb + 32 * j) * ROW_STRIDE + vc * 16) =
          mask_u4(rbv[j], (b_ok_mask >> j) & 1u);
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      uint4 v = ra[j];
      if (a.pro_mode != PRO_NONE) {  // (uniform; padding and tails are zeroed AFTER the prologue)
        float f[VEC];
        Vec<T>::unpack(v, f);
        apply_prologue_regs<VEC>(f, a.pro_mode, ps, pt);
        v = Vec<T>::pack(f);
      }
      v = mask_u4(v, (a_ok_mask >> j) & 1u);
      *reinterpret_cast<uint4*>(sA + (rb + 32 * j) * ROW_STRIDE + vc * 16) = v;
    }
  };
  load_slab(0);
  if (DBUF) {
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) load_slab(kt + 1);  // global loads in flight under the MFMAs
      mma_slab<T>(smem + cur * (A_BYTES + B_BYTES), smem + cur * (A_BYTES + B_BYTES) + A_BYTES,
                  wm, wn, lane, acc);
      if (kt + 1 < nk) stage(cur ^ 1);
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      stage(0);
      __syncthreads();
      if (kt + 1 < nk) load_slab(kt + 1);
      mma_slab<T>(smem, smem + A_BYTES, wm, wn, lane, acc);
      __syncthreads();
    }
  }

  // ---- epilogue: accumulators -> (bias) -> LDS (per-wave 32x64 patch, row-major) -> 16-byte
  // coalesced stores.  Column sums for the BatchNorm statistics are taken from the registers.
  const int col = lane & 31, hh = lane >> 5;
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};
  constexpr int EP_STRIDE = 64 * (int)sizeof(T) + 16;  // bytes per staged row (+pad)
  unsigned char* ep = smem + wave * 32 * EP_STRIDE;     // 4 waves x 32 rows
  float bias2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int o = n0 + wn * 64 + j * 32 + col;
    bias2[j] = (a.bias != nullptr && o < a.O) ? a.bias[o] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * hh;
        const float v = acc[i][j][e];
        csum[j] += v;
        csq[j] += v * v;
        Vec<T>::store1(reinterpret_cast<T*>(ep + r * EP_STRIDE) + j * 32 + col, v + bias2[j]);
      }
    }
    // each wave only touches its own patch -> wave-level ordering is enough, but the compiler
    // needs a barrier-class fence for LDS; all four waves run this in lockstep anyway
    __syncthreads();
    constexpr int VPR = 64 / VEC;  // vectors per staged row
#pragma unroll
    for (int q = 0; q < (32 * VPR) / 64; ++q) {
      const int idx = q * 64 + lane;
      const int r = idx / VPR, v = idx - r * VPR;
      const int p = m0 + wm * 64 + i * 32 + r;
      const int o = n0 + wn * 64 + v * VEC;
      if (p < a.M && o < a.O) {
        long row = p;
        if (a.out_s != 1) {
          const int wo = p % a.Wo;
          const int t = p / a.Wo;
          const int ho = t % a.Ho;
          const int n = t / a.Ho;
          row = ((long)n * a.out_H + (long)ho * a.out_s) * a.out_W + (long)wo * a.out_s;
        }
        uint4 val = *reinterpret_cast<const uint4*>(ep + r * EP_STRIDE + v * 16);
        T* dst = Y + row * a.ldy + o;
        if (a.ep_x != nullptr && o + VEC <= a.O) {
          float f[VEC], xv[VEC];
          Vec<T>::unpack(val, f);
          Vec<T>::unpack(ldg16(reinterpret_cast<const T*>(a.ep_x) + row * a.ldep + o), xv);
#pragma unroll
          for (int k = 0; k < VEC; ++k) f[k] = f[k] - a.ep_c0[o + k] - a.ep_c1[o + k] * xv[k];
          val = Vec<T>::pack(f);
        }
        if (o + VEC <= a.O) {
          stg16(dst, val);
        } else {  // ragged channel tail (e.g. 19 classes): never write past O
          float f[VEC];
          Vec<T>::unpack(val, f);
#pragma unroll
          for (int k = 0; k < VEC; ++k)
            if (o + k < a.O) Vec<T>::store1(dst + k, f[k]);
        }
      }
    }
    __syncthreads();
  }
  if (a.stat_partial != nullptr) {
    // rows beyond M were staged as zeros (after the prologue), so they add nothing.
    float* red = reinterpret_cast<float*>(smem);  // [WMS wm][2][TBN]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = csum[j] + __shfl_xor(csum[j], 32, 64);
      float q = csq[j] + __shfl_xor(csq[j], 32, 64);
      if (hh == 0) {
        const int cl = wn * 64 + j * 32 + col;
        red[(wm * 2 + 0) * TBN + cl] = s;
        red[(wm * 2 + 1) * TBN + cl] = q;
      }
    }
    __syncthreads();
    if (tid < TBN) {
      const int o = n0 + tid;
      if (o < a.O) {
        float* dst = a.stat_partial + (long)tile_m * 2 * a.O;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WMS; ++w) {  // (fixed order: deterministic)
          s += red[(w * 2 + 0) * TBN + tid];
          q += red[(w * 2 + 1) * TBN + tid];
        }
        dst[o] = s;
        dst[a.O + o] = q;
      }
    }
  }
}

// ---- a handful of pixels (r05): the ASPP image-pooling conv (2 pixels x 2048 -> 256) and the
// PSP pyramid bins (2..72 pixels x 2048 -> 512) in float32 — module.py:45-62 / 18-38.  On the
// 128x128 tile kernel ONE block walks K = 2048 as 64 dependent slabs (156-164 us per call).
// Here a wave owns 8 pixels x 4 output channels, lanes split K (16 bytes per lane and row), every
// row is read once per wave straight from L2; butterfly reduction, lane (r, j) stores.  The
// BatchNorm partial rows: one per 8-pixel chunk (seg_conv_gemm_stat_rows agrees).
constexpr int SK_ROWS = 8, SK_COLS = 4, SK_MAX_M = 128;

static bool skinny_geometry(int dtype, long M, int C, int KH, int KW, int stride, int pad,
                            int tconv, int has_bias, int pro_mode) {
  return dtype == DT_F32 && KH * KW == 1 && stride == 1 && pad == 0 && !tconv && !has_bias &&
         pro_mode == PRO_NONE && M <= SK_MAX_M && C >= 256 && C % 4 == 0;
}

__global__ __launch_bounds__(256) void conv_gemm_skinny_kernel(const ConvGemmArgs a, int ogs,
                                                               int nwaves) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= nwaves) return;
  const int mc = wid / ogs, og = wid - mc * ogs;
  const int m0 = mc * SK_ROWS, o0 = og * SK_COLS;
  const float* __restrict__ X = reinterpret_cast<const float*>(a.x);
  const float* __restrict__ W = reinterpret_cast<const float*>(a.w);
  float* __restrict__ Y = reinterpret_cast<float*>(a.y);
  const float* xr[SK_ROWS];
  const float* wr[SK_COLS];
#pragma unroll
  for (int r = 0; r < SK_ROWS; ++r) xr[r] = X + (long)min(m0 + r, a.M - 1) * a.ldx;
#pragma unroll
  for (int j = 0; j < SK_COLS; ++j) wr[j] = W + (long)min(o0 + j, a.O - 1) * a.K;
  float acc[SK_ROWS][SK_COLS];
#pragma unroll
  for (int r = 0; r < SK_ROWS; ++r)
#pragma unroll
    for (int j = 0; j < SK_COLS; ++j) acc[r][j] = 0.f;
  for (int k = lane * 4; k < a.K; k += 256) {
    float4 wv[SK_COLS], xv[SK_ROWS];
#pragma unroll
    for (int j = 0; j < SK_COLS; ++j) wv[j] = *reinterpret_cast<const float4*>(wr[j] + k);
#pragma unroll
    for (int r = 0; r < SK_ROWS; ++r) xv[r] = *reinterpret_cast<const float4*>(xr[r] + k);
#pragma unroll
    for (int r = 0; r < SK_ROWS; ++r)
#pragma unroll
      for (int j = 0; j < SK_COLS; ++j) {
        acc[r][j] = fmaf(xv[r].x, wv[j].x, acc[r][j]);
        acc[r][j] = fmaf(xv[r].y, wv[j].y, acc[r][j]);
        acc[r][j] = fmaf(xv[r].z, wv[j].z, acc[r][j]);
        acc[r][j] = fmaf(xv[r].w, wv[j].w, acc[r][j]);
      }
  }
  float mine = 0.f;
#pragma unroll
  for (int r = 0; r < SK_ROWS; ++r)
#pragma unroll
    for (int j = 0; j < SK_COLS; ++j) {
      acc[r][j] = wave_sum(acc[r][j]);  // (xor butterfly: every lane ends with the same sum)
      if (lane == r * SK_COLS + j) mine = acc[r][j];
    }
  if (lane < SK_ROWS * SK_COLS) {
    const int r = lane / SK_COLS, j = lane - r * SK_COLS;
    if (m0 + r < a.M && o0 + j < a.O) Y[(long)(m0 + r) * a.ldy + o0 + j] = mine;
  }
  if (a.stat_partial != nullptr && lane < SK_COLS && o0 + lane < a.O) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < SK_ROWS; ++r) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < SK_COLS; ++j) v = lane == j ? acc[r][j] : v;
      if (m0 + r < a.M) {
        s1 += v;
        s2 = fmaf(v, v, s2);
      }
    }
    a.stat_partial[((long)mc * 2 + 0) * a.O + o0 + lane] = s1;
    a.stat_partial[((long)mc * 2 + 1) * a.O + o0 + lane] = s2;
  }
}

static int launch_conv_gemm_skinny(const ConvGemmArgs& a, hipStream_t stream) {
  const int ogs = (a.O + SK_COLS - 1) / SK_COLS, mcs = (a.M + SK_ROWS - 1) / SK_ROWS;
  const int nwaves = ogs * mcs;
  hipLaunchKernelGGL(conv_gemm_skinny_kernel, dim3((nwaves + 3) / 4), dim3(256), 0, stream, a, ogs,
                     nwaves);
  return check_launch("conv_gemm_fwd (skinny)");
}

// 1x1 stride-1 convs: 0 = first-generation 128x128 kernel, 1 = 256x128 kernel
// (conv_gemm_px256.hip), 2 = direct-to-LDS 256x256 kernel where it applies (conv_gemm_glds.hip:
// bf16, no prologue), 256x128 otherwise
// (fixed selection; the A/B history is in profiles/r01..r02: single LDS stage, 3 blocks/CU,
// measured faster than two stages on every C3 shape — 709 vs 622 TF on 1536->2048 @65x129)
constexpr int g_gemm_px256 = 2;
constexpr bool g_conv3x3_direct = true;

// The 256 x 64 tile of the general path: few output channels, many pixels.
static bool gemm_use_wide(int KH, int KW, int stride, int pad, int tconv, int O, long M) {
  const bool fast = KH * KW == 1 && stride == 1 && pad == 0 && !tconv;
  return !fast && O <= 64 && M >= 16384;
}

template <typename T>
static int launch_conv_gemm_fwd(const ConvGemmArgs& a, hipStream_t stream) {
  const dim3 grid(a.tiles_m * a.tiles_n), block(GEMM_THREADS);
  const bool fast = a.KH * a.KW == 1 && a.stride == 1 && a.pad == 0 && !a.tconv;
  if (gemm_use_wide(a.KH, a.KW, a.stride, a.pad, a.tconv, a.O, a.M)) {
    hipLaunchKernelGGL((conv_gemm_fwd_kernel<T, false, false, true>), grid, block, 0, stream, a);
  } else if (fast) {
    hipLaunchKernelGGL((conv_gemm_fwd_kernel<T, true, false>), grid, block, 0, stream, a);
  } else {
    hipLaunchKernelGGL((conv_gemm_fwd_kernel<T, false, false>), grid, block, 0, stream, a);
  }
  return check_launch("conv_gemm_fwd");
}

}  // namespace seg

// The 256x128 kernel pays off where an output row is wide enough to amortise its longer
// per-block pipeline (tools/gemm_bench.py, bf16 TFLOP/s, 128x128 -> 256x128: 728->728 @65x129
// 419 -> 487, 1536->2048 675 -> 675, 304->256 @257x513 434 -> 430, 128->128 @513x1025 326 -> 315)
// (r02: also O >= 256 when there are >= 256 pixel tiles — the decoder's 304->256 / 256->256
// convs at 257x513: one 256-wide column tile per 256-pixel tile still fills the chip four times)
// (r05: O >= 224 there was tried for HRNet's biased 240 -> 240 last layer at 16 x 256 x 512
// pixels: 866 us on this kernel against 592 on the 128x128 one)
static bool gemm_use_px256(int KH, int KW, int stride, int pad, int tconv, int O, long M) {
  return KH * KW == 1 && stride == 1 && pad == 0 && !tconv &&
         ((O >= 384 && M >= 4096) || (O >= 256 && M >= 65536));
}

// rows of the statistics partial buffer [rows][2][O] the forward kernel will write
// stride-1 KxK convolutions wide enough for the 256x256 direct-to-LDS tile
static bool gemm_use_glds_kxk(int dtype, const seg::ConvGemmArgs& a) {
  return seg::g_gemm_px256 >= 2 && a.O >= 256 && a.M >= 4096 &&
         seg::conv_gemm_glds_kxk_usable(dtype, a);
}

extern "C" int seg_conv_gemm_stat_rows(int dtype, int N, int Ho, int Wo, int C, int O, int KH,
                                       int KW, int stride, int pad, int dil, int tconv,
                                       int has_bias, int pro_mode) {
  const long M = (long)N * Ho * Wo;
  if (seg::skinny_geometry(dtype, M, C, KH, KW, stride, pad, tconv, has_bias, pro_mode))
    return (int)((M + seg::SK_ROWS - 1) / seg::SK_ROWS);
  if (seg::g_gemm_px256 && gemm_use_px256(KH, KW, stride, pad, tconv, O, M)) {
    // the direct-to-LDS kernel (bf16, no prologue / bias: conv_gemm_glds_usable up to the pitch
    // checks, which seg_conv_gemm_fwd settles by zero-filling the rows a fallback leaves out)
    if (seg::g_gemm_px256 >= 2 && dtype == seg::DT_BF16 && pro_mode == seg::PRO_NONE &&
        !has_bias && (C % 8) == 0 && (O % 8) == 0)
      return seg::glds_tiles_m(M, O, false);
    return seg::px256_tiles_m(M);
  }
  {  // KxK on the direct-to-LDS pipeline (the predicate does not look at the input size)
    seg::ConvGemmArgs probe = {};
    probe.KH = KH; probe.KW = KW; probe.stride = stride; probe.pad = pad; probe.dil = dil;
    probe.tconv = tconv; probe.out_s = 1; probe.C = C; probe.O = O; probe.ldx = 8; probe.ldy = 8;
    probe.N = 1; probe.Hi = 1; probe.Wi = 1; probe.M = (int)M; probe.pro_mode = pro_mode;
    probe.bias = has_bias ? reinterpret_cast<const float*>(&probe) : nullptr;
    if (gemm_use_glds_kxk(dtype, probe)) return seg::glds_tiles_m(M, O, true);
  }
  {  // the direct 3x3 kernel (stride 1, pad 1: the input has the output's size)
    seg::ConvGemmArgs probe = {};
    probe.KH = KH; probe.KW = KW; probe.stride = stride; probe.pad = pad; probe.dil = dil;
    probe.tconv = tconv; probe.out_s = 1; probe.C = C; probe.O = O; probe.ldx = 8; probe.ldy = 8;
    probe.Ho = probe.Hi = Ho; probe.Wo = probe.Wi = Wo; probe.M = (int)M;
    probe.bias = has_bias ? reinterpret_cast<const float*>(&probe) : nullptr;
    if (seg::g_conv3x3_direct && seg::conv3x3_direct_usable(dtype, probe))
      return seg::conv3x3_direct_blocks(N, Ho, Wo, C);
  }
  if (seg::gemm_use_wide(KH, KW, stride, pad, tconv, O, M)) return (int)((M + 255) / 256);
  return (int)((M + seg::BM - 1) / seg::BM);
}

extern "C" int seg_conv_gemm_fwd(int dtype, const void* x, long ldx, int N, int Hi, int Wi, int C,
                                 const void* w, int O, int KH, int KW, int stride, int pad,
                                 int dil, int pro_mode, const float* pro_scale,
                                 const float* pro_shift, const float* bias, void* y, long ldy,
                                 int Ho, int Wo, int out_H, int out_W, int out_s,
                                 float* stat_partial, const void* ep_x, long ldep,
                                 const float* ep_c0, const float* ep_c1, int tconv,
                                 void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "conv_gemm_fwd: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldx % vec == 0,
              "conv_gemm_fwd: C=%d / ldx=%ld must be multiples of %d", C, ldx, vec);
  SEG_REQUIRE(N > 0 && Ho > 0 && Wo > 0 && O > 0, "conv_gemm_fwd: empty problem");
  SEG_REQUIRE(((pro_mode & PRO_AFFINE) == 0) || (pro_scale && pro_shift),
              "conv_gemm_fwd: affine prologue without scale/shift");
  SEG_REQUIRE((long)N * Ho * Wo < (1L << 31), "conv_gemm_fwd: M overflows int");
  ConvGemmArgs a;
  a.x = x; a.w = w; a.y = y;
  a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.bias = bias; a.stat_partial = stat_partial;
  a.ep_x = ep_x; a.ep_c0 = ep_c0; a.ep_c1 = ep_c1; a.ldep = ldep;
  SEG_REQUIRE(ep_x == nullptr || (ep_c0 && ep_c1 && O % vec == 0 && ldep % vec == 0),
              "conv_gemm_fwd: epilogue correction needs c0/c1 and vector-aligned O/ldep");
  a.ldx = ldx; a.ldy = ldy;
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.C = C; a.Ho = Ho; a.Wo = Wo; a.O = O;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.dil = dil;
  a.pro_mode = pro_mode; a.tconv = tconv;
  a.M = N * Ho * Wo; a.K = KH * KW * C;
  a.out_H = out_H; a.out_W = out_W; a.out_s = out_s;
  a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (O + BN - 1) / BN;
  if (gemm_use_wide(KH, KW, stride, pad, tconv, O, a.M)) {
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (O + 63) / 64;
  }
  SEG_REQUIRE(out_s == 1 || stat_partial == nullptr, "conv_gemm_fwd: no statistics with scatter");
  if (out_s == 1 && skinny_geometry(dtype, a.M, C, KH, KW, stride, pad, tconv, bias != nullptr,
                                    pro_mode)) {
    if (ep_x == nullptr) return launch_conv_gemm_skinny(a, (hipStream_t)stream);
    // (a correction epilogue on a handful of pixels: the tile kernel below writes ONE statistics
    // row where seg_conv_gemm_stat_rows promised one per 8 pixels — the others read as zero)
    const int promised = (a.M + SK_ROWS - 1) / SK_ROWS;
    if (stat_partial != nullptr && promised > 1 &&
        hipMemsetAsync(stat_partial + 2L * O, 0, (size_t)(promised - 1) * 2 * O * sizeof(float),
                       (hipStream_t)stream) != hipSuccess) {
      set_error("conv_gemm_fwd: cannot clear the unused statistics rows");
      return 2;
    }
  }
  if (g_gemm_px256 && gemm_use_px256(KH, KW, stride, pad, tconv, O, a.M) && out_s == 1) {
    if (g_gemm_px256 >= 2 && conv_gemm_glds_usable(dtype, a))
      return launch_conv_gemm_glds(a, (hipStream_t)stream);
    if (stat_partial != nullptr) {
      // the caller sized the statistics rows with seg_conv_gemm_stat_rows, which cannot see the
      // pitches: rows this kernel will not write must read as zero
      const int promised = seg_conv_gemm_stat_rows(dtype, N, Ho, Wo, C, O, KH, KW, stride, pad, dil,
                                                   tconv, bias != nullptr, pro_mode);
      const int written = px256_tiles_m(a.M);
      if (promised > written &&
          hipMemsetAsync(stat_partial + (long)written * 2 * O, 0,
                         (size_t)(promised - written) * 2 * O * sizeof(float),
                         (hipStream_t)stream) != hipSuccess) {
        set_error("conv_gemm_fwd: cannot clear the unused statistics rows");
        return 2;
      }
    }
    return launch_conv_gemm_px256(dtype, a, (hipStream_t)stream);
  }
  if (g_conv3x3_direct && conv3x3_direct_usable(dtype, a))
    return launch_conv3x3_direct(a, (hipStream_t)stream);
  if (out_s == 1 && gemm_use_glds_kxk(dtype, a))
    return launch_conv_gemm_glds_kxk(a, (hipStream_t)stream);
  if (dtype == DT_BF16) return launch_conv_gemm_fwd<bf16_t>(a, (hipStream_t)stream);
  return launch_conv_gemm_fwd<float>(a, (hipStream_t)stream);
}

extern "C" int seg_conv_gemm_tiles_m(int N, int Ho, int Wo) {
  return (N * Ho * Wo + seg::BM - 1) / seg::BM;
}
