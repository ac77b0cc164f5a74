// Direct 3x3 / stride-1 / pad-1 convolution for FEW channels at LARGE spatial sizes (bf16):
// xception conv2 32->64 and its data gradient 64->32 at 513x1025; r05: HRNet's 16 -> 16 basic
// blocks of the full-resolution branch (hrnet.py BasicBlock, 12 convolutions per forward at
// 16 x 256 x 512 pixels: 186-260 us each on the implicit GEMM for 134 MB of traffic)
// (segmentron/models/backbones/xception.py:66-75; C = O = 64 does not fit the register budget and
// stays on the implicit GEMM's 256x64 tile), where the implicit GEMM's K slabs are latency-bound: a 128-pixel block re-gathers its operand once per 64-k slab
// and waits a full memory round trip each time (conv2 forward: 191 us for 215 MB / 39 GFLOP).
//
// Here a block owns an 8 x 32 output tile and loads its input ONCE:
//   * halo tile (10 x 34 pixels x C) -> registers -> BatchNorm/ReLU prologue -> LDS, pixel pitch
//     2C + 16 bytes (80 / 144 B: the 16 lanes of a ds_read_b128 phase — consecutive pixels, one
//     16-byte k chunk — hit 16 different 4-bank groups, conflict-free); halo pixels outside the
//     image are zero AFTER the prologue, as the reference's zero padding is;
//   * the WEIGHTS of a wave's 32 output channels stay in registers for the whole kernel
//     (9 taps x C/16 k-steps x 16 B per lane: 72 / 144 VGPRs), blocks are persistent
//     (<= 2 per CU) and walk a contiguous, XCD-local range of tiles;
//   * per LDS row of the tile a wave reads each pixel fragment once and feeds it to the up to
//     three (kh, output row) pairs that use it: (PXG + 2) * 3 * C/16 ds_read_b128 for
//     9 * C/16 * PXG MFMAs (v_mfma_f32_32x32x16_bf16, A = weights so that a lane ends up with
//     4 consecutive output channels of ONE pixel);
//   * epilogue through a per-wave LDS patch -> 16-byte NHWC stores, BatchNorm statistics of the
//     values as stored, accumulated in registers across the block's tiles: one partial row per
//     block.
// Wave layout: NOG = O / 32 output-channel groups; wave w owns group w % NOG and output rows
// (w / NOG) * PXG .. + PXG, PXG = 2 * NOG.
#include "conv_gemm.h"
#include "conv_gemm_args.h"

namespace seg {

constexpr int D3_TH = 8, D3_TW = 32, D3_HH = D3_TH + 2, D3_HW = D3_TW + 2;
constexpr int D3_THREADS = 256;
constexpr int D3_MAX_BLOCKS = 512;
// C = 16: 20 KB of LDS and ~120 VGPRs per block — four blocks per CU hide each other's staging
constexpr int D3_C16_BLOCKS = 1024;

typedef __attribute__((address_space(3))) unsigned char d3_lds_t;

template <int C, int NOG> struct D3Geom {
  static constexpr int PXG = 2 * NOG;               // output rows per wave
  static constexpr int KS = C / 16;                 // k-steps per tap
  static constexpr int PP = C * 2 + 16;             // halo pixel pitch (bytes)
  static constexpr int HALO_BYTES = D3_HH * D3_HW * PP;
  static constexpr int OP = 32 * 2 + 16;            // patch pixel pitch: 32 channels (bytes)
  static constexpr int PATCH_BYTES = 4 * PXG * 32 * OP;  // 4 waves x PXG rows of 32 pixels
  static constexpr int LDS_BYTES = HALO_BYTES > PATCH_BYTES ? HALO_BYTES : PATCH_BYTES;
  static constexpr int VPP = C / 8;                 // 16-byte vectors per pixel
  static constexpr int NVEC = D3_HH * D3_HW * VPP;  // vectors of the halo tile
  static constexpr int PER = (NVEC + D3_THREADS - 1) / D3_THREADS;
  static constexpr int CHUNK = 6;  // staging vectors in flight per thread (register budget)
};

template <int C, int NOG, bool STATS, bool PRO>
__global__ __launch_bounds__(D3_THREADS, 2) void conv3x3_direct_kernel(const ConvGemmArgs a,
                                                                       int tiles_h, int tiles_w,
                                                                       int ntiles) {
  using G = D3Geom<C, NOG>;
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) unsigned char d3_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int og = wave % NOG, row0 = (wave / NOG) * G::PXG;
  const int r32 = lane & 31, h = lane >> 5;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ W = reinterpret_cast<const T*>(a.w);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);

  // ---- this wave's weights: [9 taps][KS] fragments of output channel og*32 + r32
  bf16x8 wf[9][G::KS];
  {
    // (O = 16, r05: the upper 16 rows of the only channel group are zero weights)
    const bool orow = og * 32 + r32 < a.O;
    const T* wrow = W + (long)(orow ? og * 32 + r32 : 0) * (9 * C) + h * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        wf[t][ks] = *reinterpret_cast<const bf16x8*>(wrow + t * C + ks * 16);
        if (!orow) wf[t][ks] = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
      }
  }
  // ---- prologue parameters of this thread's staging vector (its channel slot never changes:
  // 256 % VPP == 0)
  const int svec = tid % G::VPP;
  float ps[8], pt[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { ps[i] = 1.f; pt[i] = 0.f; }
  if (PRO && (a.pro_mode & PRO_AFFINE)) {
    load_params<8>(a.pro_scale, svec * 8, ps);
    load_params<8>(a.pro_shift, svec * 8, pt);
  }
  float s1[8], s2[8];  // statistics of channels og*32 + (lane & 3)*8 .. +8 (STATS only)
#pragma unroll
  for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;

  const int nblk = gridDim.x;
  const int L = xcd_remap(blockIdx.x, nblk);
  const int per = (ntiles + nblk - 1) / nblk;
  const int t_end = min(ntiles, (L + 1) * per);
  for (int t = L * per; t < t_end; ++t) {
    const int tw = t % tiles_w, tq = t / tiles_w;
    const int th = tq % tiles_h, n = tq / tiles_h;
    const int h0 = th * D3_TH, w0 = tw * D3_TW;
    // ---- halo tile, CHUNK vectors per thread at a time: the loads of a chunk first
    // (unconditional, clamped), then prologue + LDS
#pragma unroll
    for (int q0 = 0; q0 < G::PER; q0 += G::CHUNK) {
      constexpr int CH = G::CHUNK;
      uint4 raw[CH];
      unsigned okm = 0;
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        if (q0 + q >= G::PER) break;
        const int idx = tid + (q0 + q) * D3_THREADS;
        const int pix = idx / G::VPP;
        const int hr = pix / D3_HW, hc = pix - hr * D3_HW;
        const int hi = h0 - 1 + hr, wi = w0 - 1 + hc;
        const bool ok = idx < G::NVEC && hi >= 0 && hi < a.Hi && wi >= 0 && wi < a.Wi;
        const long off = ok ? (((long)n * a.Hi + hi) * a.Wi + wi) * a.ldx + svec * 8 : 0;
        raw[q] = ldg16(X + off);
        okm |= ok ? (1u << q) : 0u;
      }
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        if (q0 + q >= G::PER) break;
        const int idx = tid + (q0 + q) * D3_THREADS;
        uint4 v = raw[q];
        if (PRO) {
          float f[8];
          Vec<T>::unpack(v, f);
          apply_prologue_regs<8>(f, a.pro_mode, ps, pt);
          v = Vec<T>::pack(f);
        }
        v = mask_u4(v, (okm >> q) & 1u);
        if (idx < G::NVEC) {
          const int pix = idx / G::VPP;
          *reinterpret_cast<uint4*>(d3_smem + pix * G::PP + svec * 16) = v;
        }
      }
    }
    __syncthreads();

    // ---- MFMAs
    f32x16 acc[G::PXG];
#pragma unroll
    for (int pg = 0; pg < G::PXG; ++pg)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[pg][e] = 0.f;
    const d3_lds_t* hb = (const d3_lds_t*)d3_smem + (row0 * D3_HW + r32) * G::PP + h * 16;
#pragma unroll
    for (int rr = 0; rr < G::PXG + 2; ++rr) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
          const bf16x8 b =
              *(const __attribute__((address_space(3))) bf16x8*)(hb + (rr * D3_HW + kw) * G::PP +
                                                                 ks * 32);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int pg = rr - kh;
            if (pg >= 0 && pg < G::PXG)
              acc[pg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kh * 3 + kw][ks], b, acc[pg], 0,
                                                                0, 0);
          }
        }
      }
    }
    __syncthreads();  // every wave is done with the halo tile: its space becomes the patches

    // ---- epilogue: per wave, PXG rows of [32 px][32 ch] patches -> 16-byte stores
    unsigned char* patch = d3_smem + wave * (G::PXG * 32 * G::OP);
#pragma unroll
    for (int pg = 0; pg < G::PXG; ++pg) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint2 pk = make_uint2(pack_bf16x2(acc[pg][g * 4 + 0], acc[pg][g * 4 + 1]),
                                    pack_bf16x2(acc[pg][g * 4 + 2], acc[pg][g * 4 + 3]));
        *reinterpret_cast<uint2*>(patch + (pg * 32 + r32) * G::OP + (g * 8 + h * 4) * 2) = pk;
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < G::PXG * 2; ++it) {  // 64 lanes = 16 pixels x 4 vectors per pass
      const int pg = it >> 1;
      const int px = (it & 1) * 16 + (lane >> 2), v = lane & 3;
      const int ho = h0 + row0 + pg, wo = w0 + px;
      const uint4 val = *reinterpret_cast<const uint4*>(patch + (pg * 32 + px) * G::OP + v * 16);
      if (ho < a.Ho && wo < a.Wo && og * 32 + v * 8 < a.O) {
        stg16(Y + (((long)n * a.Ho + ho) * a.Wo + wo) * a.ldy + og * 32 + v * 8, val);
        if (STATS) {
          float f[8];
          Vec<T>::unpack(val, f);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            s1[i] += f[i];
            s2[i] = fmaf(f[i], f[i], s2[i]);
          }
        }
      }
    }
    __syncthreads();  // patches are read: the next tile's halo may overwrite them
  }

  if (STATS) {
    // lanes with the same (lane & 3) hold the same 8 channels: fold bits 2..5, then the waves
    // of one channel group through LDS (fixed order: deterministic)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int m = 4; m < 64; m <<= 1) {
        s1[i] += __shfl_xor(s1[i], m, 64);
        s2[i] += __shfl_xor(s2[i], m, 64);
      }
    }
    float* red = reinterpret_cast<float*>(d3_smem);  // [4 waves][2][32]
    if (lane < 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        red[(wave * 2 + 0) * 32 + lane * 8 + i] = s1[i];
        red[(wave * 2 + 1) * 32 + lane * 8 + i] = s2[i];
      }
    }
    __syncthreads();
    if (tid < 32 * NOG) {
      const int g = tid >> 5, c = tid & 31;
      float s = 0.f, q = 0.f;
      for (int w = g; w < 4; w += NOG) {
        s += red[(w * 2 + 0) * 32 + c];
        q += red[(w * 2 + 1) * 32 + c];
      }
      float* dst = a.stat_partial + (long)blockIdx.x * 2 * a.O;
      if (g * 32 + c < a.O) {
        dst[g * 32 + c] = s;
        dst[a.O + g * 32 + c] = q;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
bool conv3x3_direct_usable(int dtype, const ConvGemmArgs& a) {
  return dtype == DT_BF16 && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 &&
         !a.tconv && a.out_s == 1 && a.bias == nullptr && a.ep_x == nullptr &&
         ((a.C == 32 && (a.O == 32 || a.O == 64)) || (a.C == 64 && a.O == 32) ||
          (a.C == 16 && a.O == 16)) &&
         (a.ldx % 8) == 0 &&
         (a.ldy % 8) == 0 && a.Ho == a.Hi && a.Wo == a.Wi && (long)a.M >= 65536;
}

int conv3x3_direct_blocks(int N, int H, int W, int C) {
  const long nt = (long)N * ((H + D3_TH - 1) / D3_TH) * ((W + D3_TW - 1) / D3_TW);
  const int cap = C == 16 ? D3_C16_BLOCKS : D3_MAX_BLOCKS;
  return (int)(nt < cap ? nt : cap);
}

template <int C, int NOG>
static int launch_d3(const ConvGemmArgs& a, hipStream_t stream) {
  using G = D3Geom<C, NOG>;
  const int tiles_h = (a.Hi + D3_TH - 1) / D3_TH, tiles_w = (a.Wi + D3_TW - 1) / D3_TW;
  const int ntiles = a.N * tiles_h * tiles_w;
  const dim3 grid(conv3x3_direct_blocks(a.N, a.Hi, a.Wi, C)), block(D3_THREADS);
  const bool st = a.stat_partial != nullptr, pro = a.pro_mode != PRO_NONE;
#define SEG_D3(S, P)                                                                            \
  hipLaunchKernelGGL((conv3x3_direct_kernel<C, NOG, S, P>), grid, block, G::LDS_BYTES, stream, a, \
                     tiles_h, tiles_w, ntiles)
  if (st && pro) SEG_D3(true, true);
  else if (st) SEG_D3(true, false);
  else if (pro) SEG_D3(false, true);
  else SEG_D3(false, false);
#undef SEG_D3
  return check_launch("conv_gemm_fwd (direct 3x3)");
}

int launch_conv3x3_direct(const ConvGemmArgs& a, hipStream_t stream) {
  if (a.C == 16) return launch_d3<16, 1>(a, stream);  // O = 16: half of the channel group
  if (a.C == 32 && a.O == 32) return launch_d3<32, 1>(a, stream);
  if (a.C == 32 && a.O == 64) return launch_d3<32, 2>(a, stream);
  return launch_d3<64, 1>(a, stream);
}


// =============================================================================================
// Weight gradient of the same geometry (C = 32 input channels, O = 32 * NOG output channels):
//     dW[o][kh][kw][c] = sum_p dy[p][o] * act(x)[p + (kh-1, kw-1)][c]
// The first-generation kernel gathers every 64-pixel slab of x nine times (once per tap) with
// per-slab index math and reduces 128x128 tiles of which half are padding (conv2: 422 us).
// Here a block stages the x halo tile (prologue applied, pixel pitch 96 B) and the dy tile
// (pixel pitch 2*O + 16 B) ONCE per 8x32-pixel tile; both are pixel-major, which is the WRONG
// way round for MFMA operands whose k index is the pixel — the fragments are read with gfx950's
// ds_read_b64_tr_b16 (lane = channel, 4 consecutive pixels per read; the pitches put the four
// pixel rows of a 16-lane read into disjoint bank ranges).  3 * NOG waves: wave w owns kernel
// row kh = w / NOG and output-channel group og = w % NOG, i.e. three [32 o][32 c] accumulators
// (kw = 0..2) that live in registers over ALL tiles of the persistent block; per 16-pixel
// k-step it reads one dy fragment and three x fragments (shifted by kw) for three MFMAs.
// Every block writes one fp32 partial [O][9*C]; the caller sums them in a fixed order
// (seg_colsum), as for the other weight-gradient kernels.
typedef short d3_v4i16 __attribute__((ext_vector_type(4)));
typedef short d3_v8i16 __attribute__((ext_vector_type(8)));

template <int NOG> struct D3WGeom {
  static constexpr int C = 32, O = 32 * NOG;
  static constexpr int THREADS = 64 * 3 * NOG;
  static constexpr int XP = 96;                       // x halo pixel pitch (bytes)
  static constexpr int DP = O * 2 + 16;               // dy pixel pitch (bytes)
  static constexpr int X_BYTES = D3_HH * D3_HW * XP;
  static constexpr int DY_BYTES = D3_TH * D3_TW * DP;
  static constexpr int LDS_BYTES = X_BYTES + DY_BYTES;
  static constexpr int XVEC = D3_HH * D3_HW * (C / 8), DVEC = D3_TH * D3_TW * (O / 8);
  static constexpr int XPER = (XVEC + THREADS - 1) / THREADS;
  static constexpr int DPER = (DVEC + THREADS - 1) / THREADS;
};

__device__ __forceinline__ bf16x8 d3_tr_frag(const d3_lds_t* p, int second_off) {
  typedef __attribute__((address_space(3))) d3_v4i16 lds_v4;
  const d3_v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)p);
  const d3_v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p + second_off));
  const d3_v8i16 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf16x8, v);
}

template <int NOG, bool PRO>
__global__ __launch_bounds__(D3WGeom<NOG>::THREADS, 2) void conv3x3_wgrad_direct_kernel(
    const void* __restrict__ xv, long ldx, const void* __restrict__ dyv, long lddy,
    const float* __restrict__ pro_scale, const float* __restrict__ pro_shift, int pro_mode,
    float* __restrict__ partial, int N, int H, int Wd, int tiles_h, int tiles_w, int ntiles) {
  using G = D3WGeom<NOG>;
  typedef bf16_t T;
  constexpr int C = G::C, O = G::O;
  extern __shared__ __attribute__((aligned(16))) unsigned char d3_smem[];
  unsigned char* xs = d3_smem;
  unsigned char* ds = d3_smem + G::X_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave / NOG, og = wave % NOG;
  const T* __restrict__ X = reinterpret_cast<const T*>(xv);
  const T* __restrict__ DY = reinterpret_cast<const T*>(dyv);

  const int xvec = tid % (C / 8), dvec = tid % (O / 8);  // (THREADS % 8 == 0: fixed per thread)
  float ps[8], pt[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { ps[i] = 1.f; pt[i] = 0.f; }
  if (PRO && (pro_mode & PRO_AFFINE)) {
    load_params<8>(pro_scale, xvec * 8, ps);
    load_params<8>(pro_shift, xvec * 8, pt);
  }
  f32x16 acc[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[kw][e] = 0.f;

  // transpose-read lane addressing: group g = lane >> 4, i = lane & 15 -> pixel 8*(g >> 1) +
  // (i >> 2) (+4 for the second read) of the k-step, channels (g & 1)*16 + (i & 3)*4 .. +3
  const int g = lane >> 4, i = lane & 15;
  const int prow = 8 * (g >> 1) + (i >> 2), cq = (g & 1) * 16 + (i & 3) * 4;
  const d3_lds_t* a_base = (const d3_lds_t*)ds + prow * G::DP + (og * 32 + cq) * 2;
  const d3_lds_t* b_base = (const d3_lds_t*)xs + (kh * D3_HW + prow) * G::XP + cq * 2;

  const int nblk = gridDim.x;
  const int L = xcd_remap(blockIdx.x, nblk);
  const int per = (ntiles + nblk - 1) / nblk;
  const int t_end = min(ntiles, (L + 1) * per);
  for (int t = L * per; t < t_end; ++t) {
    const int tw = t % tiles_w, tq = t / tiles_w;
    const int th = tq % tiles_h, n = tq / tiles_h;
    const int h0 = th * D3_TH, w0 = tw * D3_TW;
    // ---- stage: all loads first (unconditional, clamped), then prologue / mask / LDS
    uint4 rx[G::XPER], rd[G::DPER];
    unsigned okx = 0, okd = 0;
#pragma unroll
    for (int q = 0; q < G::XPER; ++q) {
      const int idx = tid + q * G::THREADS;
      const int pix = idx / (C / 8);
      const int hr = pix / D3_HW, hc = pix - hr * D3_HW;
      const int hi = h0 - 1 + hr, wi = w0 - 1 + hc;
      const bool ok = idx < G::XVEC && hi >= 0 && hi < H && wi >= 0 && wi < Wd;
      const long off = ok ? (((long)n * H + hi) * Wd + wi) * ldx + xvec * 8 : 0;
      rx[q] = ldg16(X + off);
      okx |= ok ? (1u << q) : 0u;
    }
#pragma unroll
    for (int q = 0; q < G::DPER; ++q) {
      const int idx = tid + q * G::THREADS;
      const int pix = idx / (O / 8);
      const int r = pix / D3_TW, c = pix - r * D3_TW;
      const int ho = h0 + r, wo = w0 + c;
      const bool ok = idx < G::DVEC && ho < H && wo < Wd;
      const long off = ok ? (((long)n * H + ho) * Wd + wo) * lddy + dvec * 8 : 0;
      rd[q] = ldg16(DY + off);
      okd |= ok ? (1u << q) : 0u;
    }
#pragma unroll
    for (int q = 0; q < G::XPER; ++q) {
      const int idx = tid + q * G::THREADS;
      uint4 v = rx[q];
      if (PRO) {
        float f[8];
        Vec<T>::unpack(v, f);
        apply_prologue_regs<8>(f, pro_mode, ps, pt);
        v = Vec<T>::pack(f);
      }
      v = mask_u4(v, (okx >> q) & 1u);
      if (idx < G::XVEC)
        *reinterpret_cast<uint4*>(xs + (idx / (C / 8)) * G::XP + xvec * 16) = v;
    }
#pragma unroll
    for (int q = 0; q < G::DPER; ++q) {
      const int idx = tid + q * G::THREADS;
      if (idx < G::DVEC)
        *reinterpret_cast<uint4*>(ds + (idx / (O / 8)) * G::DP + dvec * 16) =
            mask_u4(rd[q], (okd >> q) & 1u);
    }
    __syncthreads();
    // ---- 16 k-steps of 16 pixels: output row r, column half hf
#pragma unroll
    for (int r = 0; r < D3_TH; ++r) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bf16x8 fa = d3_tr_frag(a_base + (r * D3_TW + hf * 16) * G::DP, 4 * G::DP);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const bf16x8 fb =
              d3_tr_frag(b_base + (r * D3_HW + hf * 16 + kw) * G::XP, 4 * G::XP);
          acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[kw], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // tiles are consumed: the next one may overwrite them
  }
  // ---- this block's partial: dW[o][kh][kw][c], o = og*32 + row, c = lane & 31
  float* P = partial + (long)blockIdx.x * O * (9 * C);
  const int c = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int o = og * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      P[(long)o * (9 * C) + (kh * 3 + kw) * C + c] = acc[kw][e];
    }
}

bool conv3x3_wgrad_direct_usable(int dtype, int C, int O, int KH, int KW, int stride, int pad,
                                 int dil, long M, long ldx, long lddy) {
  return dtype == DT_BF16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && dil == 1 && C == 32 &&
         (O == 32 || O == 64) && M >= 65536 && (ldx % 8) == 0 && (lddy % 8) == 0;
}

int launch_conv3x3_wgrad_direct(const void* x, long ldx, const void* dy, long lddy, int N, int H,
                                int W, int O, int pro_mode, const float* pro_scale,
                                const float* pro_shift, float* partial, hipStream_t stream) {
  const int tiles_h = (H + D3_TH - 1) / D3_TH, tiles_w = (W + D3_TW - 1) / D3_TW;
  const int ntiles = N * tiles_h * tiles_w;
  const dim3 grid(conv3x3_direct_blocks(N, H, W));
  const bool pro = pro_mode != PRO_NONE;
#define SEG_D3W(NOG, P)                                                                         \
  hipLaunchKernelGGL((conv3x3_wgrad_direct_kernel<NOG, P>), grid, dim3(D3WGeom<NOG>::THREADS),  \
                     D3WGeom<NOG>::LDS_BYTES, stream, x, ldx, dy, lddy, pro_scale, pro_shift,   \
                     pro_mode, partial, N, H, W, tiles_h, tiles_w, ntiles)
  static const int once = [] {
    int rc = (int)hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv3x3_wgrad_direct_kernel<2, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, D3WGeom<2>::LDS_BYTES);
    rc |= (int)hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv3x3_wgrad_direct_kernel<2, false>),
        hipFuncAttributeMaxDynamicSharedMemorySize, D3WGeom<2>::LDS_BYTES);
    return rc;
  }();
  if (once != 0) {
    set_error("conv3x3_wgrad_direct: cannot reserve %d bytes of LDS", D3WGeom<2>::LDS_BYTES);
    return 2;
  }
  if (O == 64) { if (pro) SEG_D3W(2, true); else SEG_D3W(2, false); }
  else { if (pro) SEG_D3W(1, true); else SEG_D3W(1, false); }
#undef SEG_D3W
  return check_launch("conv_gemm_wgrad (direct 3x3)");
}

}  // namespace seg
