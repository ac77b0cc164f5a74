// Weight gradient of a 1x1 / stride-1 convolution on the direct-to-LDS pipeline (bf16):
//     dW[o][c] = sum_p dy[p][o] * x[p][c]
// Both operands are PIXEL-major in HBM (NHWC), i.e. the reduction index p is the slow one — the
// first-generation kernel (conv_gemm_wgrad.hip) transposes 8x8 blocks in registers while staging
// (5 VALU per MFMA).  Here the tiles go to LDS as they are (global_load_lds, 16 B per lane, no
// VGPRs, no VALU) and the MFMA fragments come out of LDS already transposed with gfx950's
// ds_read_b64_tr_b16 (measured semantics, tools/lab/tr_probe: within a 16-lane group, lane i of
// the result holds rows 4j + (i >> 2), j = 0..3, of the 4-element words addressed by the group's
// lanes — with the addressing below: lane = channel, 4 consecutive pixels per read).
//
// Block tile 128 (o) x 128 (c), eight waves as 4 (o blocks of 32) x 2 (c halves of 64): two
// 32x32x16 MFMA tiles per wave.  K loop over pixels in slots of 64: slot = dy sub-tile
// [64 px][128 ch] + x sub-tile [64 px][128 ch] = 32 KiB; four slots in LDS (128 KiB), three in
// flight (as conv_gemm_glds: the DMA round trip under load is longer than one slot of MFMAs).
// The pixel range is split over `splits` blocks per output tile so that ~256 blocks fill the
// chip; split s writes its fp32 partial tile to partial[s][O][K] (summed in fixed order by the
// consumer: seg_fold_bwd_reduce / seg_colsum), exactly like the first-generation kernel.
//
// Sub-tile LDS image: 1 KiB chunk = 4 pixel rows x 256 B; the 16-byte piece c16 (8 channels) of
// row r sits at (r & 3)*256 + ((c16 + 4*(r & 3)) & 15)*16 inside its chunk — a rotation by four
// pieces per row, so the four rows x 64 B one transpose read touches fall into four different
// bank quarters.  The rotation is applied to the DMA's per-lane SOURCE address (the LDS side of
// global_load_lds is lane-linear).
#include "conv_gemm.h"
#include "conv_gemm_wgrad_args.h"
#include "gemm_glds.h"

namespace seg {

constexpr int WG_BM = 128, WG_BN = 128, WG_BK = 64;
constexpr int WG_THREADS = 512;
constexpr int WG_SUB = 16 * 1024, WG_SLOT = 32 * 1024, WG_LDS = 4 * WG_SLOT;

typedef unsigned long long u64_t;

__device__ __forceinline__ u64_t tr_read(unsigned addr) {
  u64_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}
template <int OFF>
__device__ __forceinline__ u64_t tr_read_off(unsigned addr) {
  u64_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}

struct WgFrags { u64_t a[2], b[2][2]; };  // [k half] ; b[c block][k half]

__device__ __forceinline__ bf16x8 wg_join(u64_t lo, u64_t hi) {
  typedef u64_t u64x2 __attribute__((ext_vector_type(2)));
  u64x2 v = {lo, hi};
  return __builtin_bit_cast(bf16x8, v);
}

// KXK: stride-1 KxK convolution with C % 128 == 0 — a 128-wide tile of the K = KH*KW*C axis
// then lies inside ONE tap, and the x sub-tile of a slot is the same 64 output pixels shifted by
// that tap (per-lane DMA source; the zero word outside the image).
template <bool KXK>
__global__ __launch_bounds__(WG_THREADS, 2) void conv_wgrad_glds_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lds_byte_t* lds = (lds_byte_t*)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave >> 1, wc = wave & 1;
  // block -> (split, o tile, c tile).  Logical ids are split-major and every XCD works on a
  // contiguous range of them (xcd_remap), i.e. on the ~32 output tiles of ONE pixel range
  // (at most two): those blocks walk the same dy / x rows at the same pace, so a row is pulled
  // into that XCD's L2 once instead of once per block (first version: blockIdx % tiles put the
  // six blocks sharing a dy column panel on six XCDs — 278 MB fetched per 728x728 launch for
  // 49 MB of operands, L2 hit rate 24 %)
  const int tiles = a.tiles_o * a.tiles_k;
  const int L = xcd_remap(blockIdx.x, tiles * a.splits);
  const int tile = L % tiles, split = L / tiles;
  const int o0 = (tile / a.tiles_k) * WG_BM, c0 = (tile % a.tiles_k) * WG_BN;
  const int p0 = split * a.chunk;
  const int p1 = min(a.M, p0 + a.chunk);
  const int nslot = (p1 - p0 + WG_BK - 1) / WG_BK;

  // ---- DMA pieces of this thread: chunks wave*2 + {0,1} (4 pixel rows each) of dy and of x
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_gl_zero);
  const int rr = lane >> 4;                          // row inside the chunk
  const int c16 = ((lane & 15) - 4 * rr) & 15;       // source piece of this lane's LDS position
  const bool ch_dy = o0 + c16 * 8 < a.O, ch_x = c0 + c16 * 8 < (KXK ? a.K : a.C);
  const unsigned char* src[4];
  int prow[2];
  // KXK: tap of this K tile and the (n, ho, wo) of this thread's two pixel rows, advanced by
  // 64 pixels per slot (Wo >= 64: at most one row wrap per step)
  int kdh = 0, kdw = 0, kc0 = c0, pn[2] = {0, 0}, ph[2] = {0, 0}, pw[2] = {0, 0};
  if (KXK) {
    const int tap = c0 / a.C;
    kc0 = c0 - tap * a.C;
    const int kh = tap / a.KW;
    kdh = kh * a.dil - a.pad;
    kdw = (tap - kh * a.KW) * a.dil - a.pad;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    prow[j] = p0 + (wave * 2 + j) * 4 + rr;          // pixel of slot 0; +64 per slot
    src[j] = reinterpret_cast<const unsigned char*>(a.dy) +
             ((long)prow[j] * a.lddy + o0 + c16 * 8) * 2;
    src[2 + j] = reinterpret_cast<const unsigned char*>(a.x) +
                 ((long)prow[j] * a.ldx + c0 + c16 * 8) * 2;
    if (KXK) {
      pw[j] = prow[j] % a.Wo;
      const int t = prow[j] / a.Wo;
      ph[j] = t % a.Ho;
      pn[j] = t / a.Ho;
    }
  }
  const long inc_dy = (long)WG_BK * a.lddy * 2, inc_x = (long)WG_BK * a.ldx * 2;
  auto issue = [&](int slot) {
    lds_byte_t* base = lds + (slot & 3) * WG_SLOT + (wave * 2) * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool rok = prow[j] + slot * WG_BK < p1;
      const unsigned char* pd = (rok && ch_dy) ? src[j] : zero;
      const unsigned char* px = (rok && ch_x) ? src[2 + j] : zero;
      if (KXK) {
        const int hi = ph[j] + kdh, wi = pw[j] + kdw;
        const bool ok = rok && ch_x && hi >= 0 && hi < a.Hi && wi >= 0 && wi < a.Wi;
        px = ok ? reinterpret_cast<const unsigned char*>(a.x) +
                      ((((long)pn[j] * a.Hi + hi) * a.Wi + wi) * a.ldx + kc0 + c16 * 8) * 2
                : zero;
        pw[j] += WG_BK;  // next slot: 64 pixels on
        if (pw[j] >= a.Wo) {
          pw[j] -= a.Wo;
          if (++ph[j] == a.Ho) { ph[j] = 0; ++pn[j]; }
        }
      }
      src[j] += inc_dy;
      src[2 + j] += inc_x;
      __builtin_amdgcn_global_load_lds((glb_byte_t*)pd, base + j * 1024, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_byte_t*)px, base + j * 1024 + WG_SUB, 16, 0, 0);
    }
  };
  issue(0); issue(1); issue(2); issue(3);

  // ---- transpose-read addressing: lane -> (pixel row kk + (i >> 2), channel cb*32 + (g&1)*16
  // + (i & 3)*4) of a sub-tile; kk = 8*(g >> 1) (+4 for the second read of a k-step)
  const int g = lane >> 4, i = lane & 15;
  auto lane_base = [&](int cb) {
    const int q = cb * 4 + (g & 1) * 2 + ((i & 3) >> 1);        // 16-byte piece of the row
    const int cpos = (q + 4 * (i >> 2)) & 15;                   // rotated position
    return (g >> 1) * 2048 + (i >> 2) * 256 + cpos * 16 + (i & 1) * 8;
  };
  const unsigned lds0 = (unsigned)(unsigned long)lds;
  const unsigned addr_a = lds0 + lane_base(wo);                  // dy sub-tile, o block wo
  const unsigned addr_b0 = lds0 + WG_SUB + lane_base(wc * 2);    // x sub-tile, c blocks 2wc, 2wc+1
  const unsigned addr_b1 = lds0 + WG_SUB + lane_base(wc * 2 + 1);
  // k-step s of a slot: pixel rows 16s .. 16s+15 = chunks 4s .. 4s+3: +4096 B per step,
  // +1024 B for the second half (rows +4)
  auto read = [&](WgFrags& f, int slot, auto step) {
    constexpr int S = decltype(step)::value;
    const unsigned off = (slot & 3) * WG_SLOT;  // (one add per base; the rest are immediates)
    f.a[0] = tr_read_off<S * 4096>(addr_a + off);
    f.a[1] = tr_read_off<S * 4096 + 1024>(addr_a + off);
    f.b[0][0] = tr_read_off<S * 4096>(addr_b0 + off);
    f.b[0][1] = tr_read_off<S * 4096 + 1024>(addr_b0 + off);
    f.b[1][0] = tr_read_off<S * 4096>(addr_b1 + off);
    f.b[1][1] = tr_read_off<S * 4096 + 1024>(addr_b1 + off);
  };
  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;
  typedef std::integral_constant<int, 2> S2;
  typedef std::integral_constant<int, 3> S3;
  f32x16 acc[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[0][e] = acc[1][e] = 0.f;
  auto mma = [&](const WgFrags& f) {
    const bf16x8 fa = wg_join(f.a[0], f.a[1]);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, wg_join(f.b[0][0], f.b[0][1]), acc[0], 0,
                                                     0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, wg_join(f.b[1][0], f.b[1][1]), acc[1], 0,
                                                     0, 0);
  };
  // the transpose reads are inline asm, so hipcc inserts no waits for them: every wait is
  // explicit, and the fragments are passed THROUGH the wait so that their consumers cannot be
  // scheduled above it
#define WG_WAIT_FRAGS(f, cnt)                                                                  \
  asm volatile("s_waitcnt " cnt                                                               \
               : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), \
                 "+v"(f.b[1][1])                                                               \
               :                                                                               \
               : "memory")
  WgFrags f0, f1;
  GL_WAIT_VM(12);
  __builtin_amdgcn_s_barrier();
  read(f0, 0, S0{});
  for (int j = 0; j < nslot; ++j) {
    read(f1, j, S1{});
    WG_WAIT_FRAGS(f0, "lgkmcnt(6)");
    mma(f0);
    read(f0, j, S2{});
    WG_WAIT_FRAGS(f1, "lgkmcnt(6)");
    mma(f1);
    read(f1, j, S3{});
    WG_WAIT_FRAGS(f0, "lgkmcnt(6)");
    mma(f0);
    // my reads of slot j have returned; my DMA of slot j+1 has landed (j+2, j+3 in flight)
    WG_WAIT_FRAGS(f1, "vmcnt(8) lgkmcnt(0)");
    __builtin_amdgcn_s_barrier();
    issue(j + 4);
    read(f0, j + 1, S0{});
    mma(f1);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef WG_WAIT_FRAGS

  // ---- epilogue: fp32 partial tile of this split.  C layout: col = lane & 31 -> c,
  // row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5) -> o
  float* __restrict__ P = a.partial + (long)split * a.O * a.K;
  const int r32 = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int c = c0 + (wc * 2 + jb) * 32 + r32;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int o = o0 + wo * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (o < a.O && c < a.K) P[(long)o * a.K + c] = acc[jb][e];
    }
  }
}

bool conv_wgrad_glds_usable(int dtype, const WgradArgs& a) {
  const bool common = dtype == DT_BF16 && a.pro_mode == PRO_NONE && a.stride == 1 &&
                      (a.O % 8) == 0 && (a.ldx % 8) == 0 && (a.lddy % 8) == 0 && a.M >= 2048;
  if (!common) return false;
  // (narrow outputs — the decoder's 256 -> 48 and the 256 -> 19(24) classifier on 263 k pixels —
  // waste most of a 128-wide tile's MFMAs but are bound by the one pass over x anyway: 48 / 115 us
  // on the first-generation kernel, whose 128 x 128 register-transposed tiles re-read x per split)
  if (a.KH == 1 && a.KW == 1)
    return a.pad == 0 && (a.C % 8) == 0 && ((long)a.O * a.C >= 128 * 128 || a.M >= 32768);
  // KxK: a 128-wide K tile inside one tap, at most one row wrap per 64-pixel slot
  return (a.C % 128) == 0 && a.Wo >= WG_BK && a.O >= 128 &&
         (long)a.N * a.Hi * a.Wi < (1L << 31);
}

// splits so that tiles x splits ~ one block per CU in ONE round (up to 10 % over: a second,
// nearly empty round costs more than it saves), each split at least 8 slots deep
int conv_wgrad_glds_splits(long M, int O, int K) {
  const int tiles = ((O + WG_BM - 1) / WG_BM) * ((K + WG_BN - 1) / WG_BN);
  long want = (256 + tiles / 2) / tiles;
  if (want * tiles > 282) want = 256 / tiles;
  const long maxs = (M + 8 * WG_BK - 1) / (8 * WG_BK);
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  return (int)want;
}

int launch_conv_wgrad_glds(WgradArgs a, hipStream_t stream) {
  a.tiles_o = (a.O + WG_BM - 1) / WG_BM;
  a.tiles_k = (a.K + WG_BN - 1) / WG_BN;
  const int slots = (a.M + WG_BK - 1) / WG_BK;
  a.chunk = ((slots + a.splits - 1) / a.splits) * WG_BK;
  static const int once = [] {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_glds_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_glds_kernel<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS);
    return rc;
  }();
  if (once != 0) {
    set_error("conv_wgrad_glds: cannot reserve %d bytes of LDS", WG_LDS);
    return 2;
  }
  const dim3 grid(a.tiles_o * a.tiles_k * a.splits), block(WG_THREADS);
  if (a.KH * a.KW > 1)
    hipLaunchKernelGGL(conv_wgrad_glds_kernel<true>, grid, block, WG_LDS, stream, a);
  else
    hipLaunchKernelGGL(conv_wgrad_glds_kernel<false>, grid, block, WG_LDS, stream, a);
  return check_launch("conv_gemm_wgrad (glds)");
}

}  // namespace seg
