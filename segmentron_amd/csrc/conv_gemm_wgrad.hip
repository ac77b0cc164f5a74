// Implicit-GEMM convolution weight gradient (the "TN" GEMM of the backward pass):
//   dW[o, (kh,kw,c)] = sum_p dY[p, o] * act(X[gather(p,kh,kw), c])        p over N*Ho*Wo
// = autograd's conv2d weight gradient for the nn.Conv2d call sites listed in conv_gemm_fwd.hip.
//
// Both operands are stored pixel-major (NHWC rows), i.e. the reduction index p is the SLOW
// index of both, while MFMA wants each lane to hold consecutive reduction elements.  The
// transposition is done in registers while staging: a thread loads VEC pixel rows x one 16-byte
// channel vector (a VEC x VEC block), transposes it (free renaming for f32, 32 v_perm-class ops
// for bf16) and writes VEC 16-byte vectors [channel][p..p+VEC) into the same k-contiguous LDS
// image the forward kernel uses — so the MFMA core (conv_gemm.h) is shared.
//
// The pixel range is split across gridDim "splits"; every block writes its fp32 partial tile to
// partial[split][O][K] and seg_colsum reduces the splits (deterministic, no atomics).
// The BatchNorm(+ReLU) prologue of the forward pass is re-applied to X on the fly (the
// activated tensor is never materialised in HBM, forward or backward).
#include "conv_gemm.h"
#include "conv_gemm_wgrad_args.h"
#include <cstdlib>

namespace seg {


template <typename T> struct Transpose;
template <> struct Transpose<float> {  // 4x4: pure renaming
  __device__ static __forceinline__ void run(const uint4 (&r)[4], uint4 (&w)[4]) {
    w[0] = make_uint4(r[0].x, r[1].x, r[2].x, r[3].x);
    w[1] = make_uint4(r[0].y, r[1].y, r[2].y, r[3].y);
    w[2] = make_uint4(r[0].z, r[1].z, r[2].z, r[3].z);
    w[3] = make_uint4(r[0].w, r[1].w, r[2].w, r[3].w);
  }
};
template <> struct Transpose<bf16_t> {  // 8x8 of 16-bit
  __device__ static __forceinline__ uint32_t lo(uint32_t a, uint32_t b) {
    return (a & 0xFFFFu) | (b << 16);
  }
  __device__ static __forceinline__ uint32_t hi(uint32_t a, uint32_t b) {
    return (a >> 16) | (b & 0xFFFF0000u);
  }
  __device__ static __forceinline__ void run(const uint4 (&r)[8], uint4 (&w)[8]) {
    // w[i] = column i of the 8x8 block: elements r[0..7][i]
    w[0] = make_uint4(lo(r[0].x, r[1].x), lo(r[2].x, r[3].x), lo(r[4].x, r[5].x), lo(r[6].x, r[7].x));
    w[1] = make_uint4(hi(r[0].x, r[1].x), hi(r[2].x, r[3].x), hi(r[4].x, r[5].x), hi(r[6].x, r[7].x));
    w[2] = make_uint4(lo(r[0].y, r[1].y), lo(r[2].y, r[3].y), lo(r[4].y, r[5].y), lo(r[6].y, r[7].y));
    w[3] = make_uint4(hi(r[0].y, r[1].y), hi(r[2].y, r[3].y), hi(r[4].y, r[5].y), hi(r[6].y, r[7].y));
    w[4] = make_uint4(lo(r[0].z, r[1].z), lo(r[2].z, r[3].z), lo(r[4].z, r[5].z), lo(r[6].z, r[7].z));
    w[5] = make_uint4(hi(r[0].z, r[1].z), hi(r[2].z, r[3].z), hi(r[4].z, r[5].z), hi(r[6].z, r[7].z));
    w[6] = make_uint4(lo(r[0].w, r[1].w), lo(r[2].w, r[3].w), lo(r[4].w, r[5].w), lo(r[6].w, r[7].w));
    w[7] = make_uint4(hi(r[0].w, r[1].w), hi(r[2].w, r[3].w), hi(r[4].w, r[5].w), hi(r[6].w, r[7].w));
  }
};

template <typename T, bool DBUF>
__global__ __launch_bounds__(GEMM_THREADS, 2) void conv_gemm_wgrad_kernel(
    const WgradArgs a) {
  constexpr int VEC = Vec<T>::N;
  constexpr int BKP = ROW_BYTES / (int)sizeof(T);  // pixels per slab (64 bf16 / 32 f32)
  constexpr int PG = BKP / VEC;                    // pixel groups per slab (8)
  constexpr int BPO = (128 / VEC) * PG;            // VECxVEC blocks per operand slab
  constexpr int NBLK = 2 * BPO / GEMM_THREADS;     // blocks per thread (1 bf16 / 2 f32)
  // (dY^T tile [o][p], X^T tile [k][p]) x {1,2} stages
  __shared__ __attribute__((aligned(16))) unsigned char smem[(DBUF ? 4 : 2) * TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntile = a.tiles_o * a.tiles_k;
  const int L = xcd_remap(blockIdx.x, ntile * a.splits);
  const int split = L / ntile;
  const int tile = L - split * ntile;
  const int tile_o = tile / a.tiles_k, tile_k = tile - tile_o * a.tiles_k;
  const int o0 = tile_o * BM, k0 = tile_k * BN;
  const int p_begin = split * a.chunk;
  const int p_end = min(a.M, p_begin + a.chunk);

  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ DY = reinterpret_cast<const T*>(a.dy);
  const bool simple = (a.KH * a.KW == 1) && a.stride == 1 && a.pad == 0;

  // per-thread block assignment (fixed across slabs)
  int b_op[NBLK], b_pg[NBLK], b_v[NBLK], b_c[NBLK], b_dh[NBLK], b_dw[NBLK];
  bool b_colok[NBLK];
#pragma unroll
  for (int q = 0; q < NBLK; ++q) {
    const int b = tid + q * GEMM_THREADS;
    b_op[q] = b / BPO;  // 0: dY, 1: X
    const int bb = b - b_op[q] * BPO;
    b_pg[q] = bb % PG;
    b_v[q] = bb / PG;
    b_c[q] = 0; b_dh[q] = 0; b_dw[q] = 0;
    if (b_op[q] == 0) {
      b_colok[q] = (o0 + b_v[q] * VEC) < a.O;  // O is a multiple of VEC or handled by row mask
    } else {
      const int kv = k0 + b_v[q] * VEC;
      b_colok[q] = kv < a.K;
      int c = kv;
      if (a.KH * a.KW != 1) {
        const int kidx = kv / a.C;
        c = kv - kidx * a.C;
        const int kh = kidx / a.KW;
        b_dh[q] = kh * a.dil;
        b_dw[q] = (kidx - kh * a.KW) * a.dil;
      }
      b_c[q] = c;
    }
  }

  // row pointers of the first slab (pixel p_begin + pg*VEC), advanced by a constant per slab
  const T* b_ptr[NBLK];
  bool b_vec[NBLK];
#pragma unroll
  for (int q = 0; q < NBLK; ++q) {
    const long p = p_begin + b_pg[q] * VEC;
    if (b_op[q] == 0) {
      const int o = o0 + b_v[q] * VEC;
      b_ptr[q] = DY + p * a.lddy + (b_colok[q] ? o : 0);
      b_vec[q] = o + VEC <= a.O;
    } else {
      b_ptr[q] = X + p * a.ldx + b_c[q];
      b_vec[q] = true;
    }
  }
  // BatchNorm prologue parameters of each block's channel vector (loop-invariant per thread)
  float ps[NBLK][VEC], pt[NBLK][VEC];
  const bool affine = (a.pro_mode & PRO_AFFINE) != 0;
#pragma unroll
  for (int q = 0; q < NBLK; ++q) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { ps[q][i] = 1.f; pt[q][i] = 0.f; }
    if (affine) {  // (uniform)
      const int cc = (b_op[q] == 1 && b_colok[q]) ? b_c[q] : 0;
      load_params<VEC>(a.pro_scale, cc, ps[q]);
      load_params<VEC>(a.pro_shift, cc, pt[q]);
    }
  }
  uint4 regs[NBLK][VEC];
  unsigned okm[NBLK];  // bit j: pixel row j of block q is real (masked when staged)
  // Every load of the slab is UNCONDITIONAL (rows that
  // do not exist read element 0 of their operand and are zeroed when staged): loads under
  // per-lane branches, each followed by its own prologue, serialised on the memory latency, and
  // the per-pixel p -> (n, ho, wo) divisions are replaced by one division per block and slab.
  auto load_slab = [&](int p0) {
#pragma unroll
    for (int q = 0; q < NBLK; ++q) {
      const bool is_x = b_op[q] == 1;
      const long ld = is_x ? a.ldx : a.lddy;
      const T* base = b_ptr[q] + (long)(p0 - p_begin) * ld;
      const int pf = p0 + b_pg[q] * VEC;
      okm[q] = 0;
      if (!is_x && !b_vec[q]) {  // ragged channel tail of dY (e.g. O = 19): element-wise
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          uint4 v = make_uint4(0, 0, 0, 0);
          if (pf + j < p_end && b_colok[q]) {
            const int o = o0 + b_v[q] * VEC;
            float f[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i)
              f[i] = (o + i < a.O) ? Vec<T>::load1(base + j * ld + i) : 0.f;
            v = Vec<T>::pack(f);
            okm[q] |= 1u << j;
          }
          regs[q][j] = v;
        }
        continue;
      }
      int wo = 0, ho = 0, n = 0;
      if (is_x && !simple) {
        wo = pf % a.Wo;
        const int t = pf / a.Wo;
        ho = t % a.Ho;
        n = t / a.Ho;
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        bool ok = pf + j < p_end && b_colok[q];
        const T* sp = base + j * ld;
        if (is_x && !simple) {
          const int hi = ho * a.stride - a.pad + b_dh[q];
          const int wi = wo * a.stride - a.pad + b_dw[q];
          ok = ok && hi >= 0 && hi < a.Hi && wi >= 0 && wi < a.Wi;
          sp = X + (((long)n * a.Hi + hi) * a.Wi + wi) * a.ldx + b_c[q];
          ++wo;
          const bool wrap = wo == a.Wo;
          wo = wrap ? 0 : wo;
          ho += wrap ? 1 : 0;
          const bool wrap2 = ho == a.Ho;
          ho = wrap2 ? 0 : ho;
          n += wrap2 ? 1 : 0;
        }
        regs[q][j] = ldg16(ok ? sp : (is_x ? X : DY));
        okm[q] |= ok ? (1u << j) : 0u;
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto stage = [&](int buf) {
    unsigned char* sA = smem + (DBUF ? buf : 0) * 2 * TILE_BYTES;
    unsigned char* sB = sA + TILE_BYTES;
#pragma unroll
    for (int q = 0; q < NBLK; ++q) {
      const bool pro = a.pro_mode != PRO_NONE && b_op[q] == 1;
      uint4 r[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        uint4 v = regs[q][j];
        if (a.pro_mode != PRO_NONE) {  // (uniform; the activation operand only)
          float f[VEC];
          Vec<T>::unpack(v, f);
          apply_prologue_regs<VEC>(f, a.pro_mode, ps[q], pt[q]);
          const uint4 u = Vec<T>::pack(f);
          const unsigned m = pro ? 0xFFFFFFFFu : 0u;
          v = make_uint4((u.x & m) | (v.x & ~m), (u.y & m) | (v.y & ~m), (u.z & m) | (v.z & ~m),
                         (u.w & m) | (v.w & ~m));
        }
        r[j] = mask_u4(v, (okm[q] >> j) & 1u);
      }
      uint4 w[VEC];
      Transpose<T>::run(r, w);
      unsigned char* dst = (b_op[q] == 0 ? sA : sB) + (b_v[q] * VEC) * ROW_STRIDE + b_pg[q] * 16;
#pragma unroll
      for (int i = 0; i < VEC; ++i) *reinterpret_cast<uint4*>(dst + i * ROW_STRIDE) = w[i];
    }
  };
  if (p_begin < p_end) {
    load_slab(p_begin);
    if (DBUF) {
      stage(0);
      __syncthreads();
      int cur = 0;
      for (int p0 = p_begin; p0 < p_end; p0 += BKP) {
        const bool more = p0 + BKP < p_end;
        if (more) load_slab(p0 + BKP);
        mma_slab<T>(smem + cur * 2 * TILE_BYTES, smem + cur * 2 * TILE_BYTES + TILE_BYTES, wm, wn,
                    lane, acc);
        if (more) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
      }
    } else {
      for (int p0 = p_begin; p0 < p_end; p0 += BKP) {
        stage(0);
        __syncthreads();
        if (p0 + BKP < p_end) load_slab(p0 + BKP);
        mma_slab<T>(smem, smem + TILE_BYTES, wm, wn, lane, acc);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: fp32 partial tile
  const int col = lane & 31, hh = lane >> 5;
  float* __restrict__ P = a.partial + (long)split * a.O * a.K;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = k0 + wn * 64 + j * 32 + col;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * hh;
        const int o = o0 + wm * 64 + i * 32 + r;
        if (o < a.O && k < a.K) P[(long)o * a.K + k] = acc[i][j][e];
      }
    }
  }
}

}  // namespace seg

namespace seg {
// kernel selection (fixed; the A/B history is in profiles/r01..r02): single LDS stage in the
// first-generation kernel (3 blocks/CU measured faster than two stages on every C3 shape),
// plain 1x1 / stride-1 KxK bf16 weight gradients on the direct-to-LDS transpose-read kernel,
// the 3x3 stride-1 stems (C = 32) on the direct halo-tile kernel
constexpr bool g_wgrad_glds = true;
constexpr bool g_wgrad_direct = true;
}

extern "C" int seg_conv_gemm_wgrad_splits(int dtype, int N, int Ho, int Wo, int C, int O, int KH,
                                          int KW, int stride, int pad, int dil, int pro_mode) {
  using namespace seg;
  const int bkp = dtype == DT_BF16 ? 64 : 32;
  const long M = (long)N * Ho * Wo;
  const int K = KH * KW * C;
  const bool plain_1x1 = KH == 1 && KW == 1 && stride == 1 && pad == 0 && pro_mode == PRO_NONE;
  if (g_wgrad_glds && pro_mode == PRO_NONE && stride == 1) {  // conv_gemm_wgrad_glds.hip
    WgradArgs probe = {};
    probe.KH = KH; probe.KW = KW; probe.stride = 1; probe.pad = pad; probe.dil = dil;
    probe.pro_mode = PRO_NONE; probe.C = C; probe.O = O; probe.M = (int)M; probe.ldx = 8;
    probe.lddy = 8; probe.N = 1; probe.Hi = 1; probe.Wi = 1; probe.Ho = Ho; probe.Wo = Wo;
    if (conv_wgrad_glds_usable(dtype, probe)) return conv_wgrad_glds_splits(M, O, K);
  }
  (void)plain_1x1;
  if (g_wgrad_direct && conv3x3_wgrad_direct_usable(dtype, C, O, KH, KW, stride, pad, dil, M, 8, 8))
    return conv3x3_direct_blocks(N, Ho, Wo);  // one partial per persistent block
  const int tiles = ((O + BM - 1) / BM) * ((K + BN - 1) / BN);
  // 512 blocks in total: 2 per CU.  (768 = all 3 resident blocks per CU measured 0.3-0.4 ms/step
  // slower on C3: a third more split partials to write and to reduce for the same GEMM.)
  long want = 512 / tiles;
  long maxs = (M + 8 * bkp - 1) / (8 * bkp);       // at least 8 slabs per split
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  return (int)want;
}

extern "C" int seg_conv_gemm_wgrad(int dtype, const void* x, long ldx, int N, int Hi, int Wi,
                                   int C, const void* dy, long lddy, int Ho, int Wo, int O, int KH,
                                   int KW, int stride, int pad, int dil, int pro_mode,
                                   const float* pro_scale, const float* pro_shift, float* partial,
                                   int splits, void* stream) {
  using namespace seg;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  const int bkp = dtype == DT_BF16 ? 64 : 32;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "conv_gemm_wgrad: bad dtype %d", dtype);
  SEG_REQUIRE(C % vec == 0 && ldx % vec == 0, "conv_gemm_wgrad: C/ldx must be multiples of %d", vec);
  SEG_REQUIRE(splits >= 1, "conv_gemm_wgrad: splits must be >= 1");
  SEG_REQUIRE(((pro_mode & PRO_AFFINE) == 0) || (pro_scale && pro_shift),
              "conv_gemm_wgrad: affine prologue without scale/shift");
  SEG_REQUIRE(lddy % vec == 0, "conv_gemm_wgrad: lddy must be a multiple of %d", vec);
  WgradArgs a;
  a.x = x; a.dy = dy; a.partial = partial; a.pro_scale = pro_scale; a.pro_shift = pro_shift;
  a.ldx = ldx; a.lddy = lddy;
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.C = C; a.Ho = Ho; a.Wo = Wo; a.O = O;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.dil = dil; a.pro_mode = pro_mode;
  a.M = N * Ho * Wo; a.K = KH * KW * C;
  a.splits = splits;
  if (g_wgrad_glds && conv_wgrad_glds_usable(dtype, a) &&
      splits == conv_wgrad_glds_splits(a.M, O, a.K))
    return launch_conv_wgrad_glds(a, (hipStream_t)stream);
  if (g_wgrad_direct &&
      conv3x3_wgrad_direct_usable(dtype, C, O, KH, KW, stride, pad, dil, a.M, ldx, lddy) &&
      Hi == Ho && Wi == Wo && splits == conv3x3_direct_blocks(N, Ho, Wo))
    return launch_conv3x3_wgrad_direct(x, ldx, dy, lddy, N, Ho, Wo, O, pro_mode, pro_scale,
                                       pro_shift, partial, (hipStream_t)stream);
  a.tiles_o = (O + BM - 1) / BM; a.tiles_k = (a.K + BN - 1) / BN;
  const int slabs = (a.M + bkp - 1) / bkp;
  a.chunk = ((slabs + splits - 1) / splits) * bkp;
  const int grid = a.tiles_o * a.tiles_k * splits;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid), b(GEMM_THREADS);
  if (dtype == DT_BF16) hipLaunchKernelGGL((conv_gemm_wgrad_kernel<bf16_t, false>), g, b, 0, st, a);
  else hipLaunchKernelGGL((conv_gemm_wgrad_kernel<float, false>), g, b, 0, st, a);
  return check_launch("conv_gemm_wgrad");
}
