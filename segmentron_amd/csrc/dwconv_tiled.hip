// LDS-tiled depthwise 3x3 (stride 1, dilation 1 or 2): forward / stride-1 data gradient (flipped
// taps) and weight gradient.  Reference call sites: segmentron/modules/basic.py:38-40,152-153.
//
// Why a second dw implementation (the strip kernels in dwconv.hip stay for stride 2 / wide
// dilations): with one thread fetching its own 3x6 neighbourhood straight from global memory every
// input vector crosses the L1/TA path 4.5 times and the activation is recomputed 4.5 times; on
// MI355X that path, not HBM, was the limit (1.1-2.0 TB/s algorithmic).  Here a 256-thread block
// owns an 8x16-pixel x 8-channel-vector tile: the (8+2d)x(16+2d) input halo tile is read from
// global memory ONCE (16-byte vectors, 128 contiguous bytes per pixel), the producer's
// BatchNorm(+ReLU) is applied ONCE, and the activated tile is parked in LDS in the storage
// dtype; the 9 taps then come from LDS (ds_read_b128, rows padded by one pixel so that the two
// rows a 16-lane phase touches hit disjoint bank halves).  Measured (tools/lab/dw_lab.hip, bf16,
// forward + statistics): [2,65,129,728] 39.4 -> 21.2 us, [2,513,1025,128] 261 -> 137 us.
//
// Blocks are persistent over tiles (grid.y = number of partial rows): depthwise weights are
// staged in LDS once, BatchNorm statistics / weight-gradient taps accumulate in registers across
// tiles and are reduced once per block (deterministic, no atomics).
#include "common.h"
#include "dwconv_tiled.h"

namespace seg {

constexpr int LT_TH = 8, LT_TW = 16, LT_CVB = 8, LT_THREADS = 256;
constexpr int DW_BWD_REM_PCT = 6;

struct DwTiledArgs {
  const void* x;       // tensor the taps read (fwd: input, dgrad: dy)
  const float* w;      // fp32 taps: [9][C] tap-major (w_layout 0) or torch's [C][9] (bit 0);
                       // bit 1: read tap 8-k for tap k (stride-1 data gradient)
  int w_layout;
  void* y;             // fwd/dgrad output
  const void* dy;      // wgrad: gradient wrt the dw output
  const float* sc; const float* sh;
  float* partial;      // fwd: [gridDim.y][2][C] statistics or null; wgrad: [gridDim.y][9][C]
  float* partial_bn;   // fused backward: [gridDim.y][2][C] or null
  const void* res;     // fused backward: tensor ADDED to the masked data gradient before the
  long ldr;            // store (the other gradient of a forked activation) or null
  long ldx, ldy, lddy;
  int N, H, W, C, CV, pro_mode, tiles_h, tiles_w, ntiles;
  // r05 remainder tiling (forward / fused backward; nb = nc = 0: the classic tiling).  Tiles
  // [0, ntiles_a) are the 8 x 16 tiles of rows [0, 8 * tiles_h) x columns [0, 16 * tiles_w);
  // then N * nb bottom-strip tiles (2 x 64, rows from hb0, the full width) and N * nc
  // right-strip tiles (32 x 4, columns from wc0, rows below hc).  A 65 x 129 map (8*8+1 x
  // 8*16+1, the Xception middle flow @1025x2049) takes 64 + 3 + 2 tiles per image instead of
  // 81 of which 17 compute one valid row or column.
  int ntiles_a, nb, nc, hb0, wc0, hc;
};


// A tile is TH x TW output pixels (TH * TW / 4 = 32 strips of four for the 32 pixel threads of
// a block).  8 x 16 everywhere, except for the REMAINDER tiles of the r05 tiling (see DwRem).
template <int TH_, int TW_, int DIL_> struct TileGeomS {
  static_assert(TH_ * TW_ == 128 && TW_ % 4 == 0, "32 pixel threads x 4-pixel strips");
  static constexpr int TH = TH_, TW = TW_, DIL = DIL_;
  static constexpr int IH = TH + 2 * DIL, IW = TW + 2 * DIL, IWP = IW + 1;
  static constexpr int NPIX = IH * IW;
  static constexpr int PER = (NPIX * LT_CVB + LT_THREADS - 1) / LT_THREADS;
  static constexpr int TILE_VECS = IH * IWP * LT_CVB;  // 16-byte units
  static constexpr int COLS = 4 + 2 * DIL;             // tile columns one 4-output strip reads
};
template <int DIL> using TileGeom = TileGeomS<LT_TH, LT_TW, DIL>;
// remainder tiles: bottom strip (<= 2 image rows) and right strip (<= 4 image columns)
template <int DIL> using TileGeomB = TileGeomS<2, 64, DIL>;
template <int DIL> using TileGeomC = TileGeomS<32, 4, DIL>;

// ---- global -> registers (all loads of the tile issued back to back)
// The address arithmetic of the six loads used to cost ~150 VALU instructions per tile and
// thread (division by the tile width, two clamps, 64-bit multiplies per load) — as much as a
// third of the multiply-accumulate work of the tile: these kernels are instruction-bound, not
// bandwidth-bound (7.6 M wave-instructions for 12.2 M outputs).  Tiles whose halo lies inside
// the image (block-uniform test) take the short path: pixel offsets relative to the tile origin
// are per-thread constants.
// pixel offsets (r * W + c) of this thread's PER halo-tile positions relative to the tile origin:
// the same for every interior tile of a launch — computed once per block, not once per tile
template <int DIL>
__device__ __forceinline__ void tile_offsets(const DwTiledArgs& a, int (&poff)[TileGeom<DIL>::PER]) {
  using G = TileGeom<DIL>;
#pragma unroll
  for (int u = 0; u < G::PER; ++u) {
    const int p = (threadIdx.x >> 3) + u * (LT_THREADS / LT_CVB);
    const int pc = p < G::NPIX ? p : G::NPIX - 1;
    const int r = pc / G::IW, c = pc - r * G::IW;
    poff[u] = r * a.W + c;
  }
}

template <typename T, typename V, int DIL>
__device__ __forceinline__ void tile_issue(const DwTiledArgs& a, const T* __restrict__ X, int n,
                                           int h0, int w0, int cv,
                                           typename V::raw_t (&raw)[TileGeom<DIL>::PER],
                                           unsigned& okmask,
                                           const int (&poff)[TileGeom<DIL>::PER]) {
  using G = TileGeom<DIL>;
  constexpr int VEC = V::N;
  okmask = 0;
  const int cvc = min(cv, a.CV - 1);
  const bool inside = h0 >= DIL && h0 + LT_TH + DIL <= a.H && w0 >= DIL && w0 + LT_TW + DIL <= a.W;
  if (inside) {
    const int origin = (n * a.H + h0 - DIL) * a.W + (w0 - DIL);  // (N*H*W < 2^31: checked on the host)
    const T* __restrict__ base = X + cvc * VEC;
#pragma unroll
    for (int u = 0; u < G::PER; ++u) {
      const int p = (threadIdx.x >> 3) + u * (LT_THREADS / LT_CVB);
      okmask |= (p < G::NPIX && cv < a.CV) ? (1u << u) : 0u;
      raw[u] = V::load_raw(base + (long)(origin + poff[u]) * a.ldx);
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < G::PER; ++u) {
    const int p = (threadIdx.x >> 3) + u * (LT_THREADS / LT_CVB);
    const int r = p / G::IW, c = p - r * G::IW;
    const int hi = h0 - DIL + r, wi = w0 - DIL + c;
    const bool ok = p < G::NPIX && cv < a.CV && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
    okmask |= ok ? (1u << u) : 0u;
    const int hic = min(max(hi, 0), a.H - 1), wic = min(max(wi, 0), a.W - 1);
    raw[u] = V::load_raw(X + (((long)n * a.H + hic) * a.W + wic) * a.ldx + cvc * VEC);
  }
}

// Remainder tiles (any shape G, on the image border by construction): global -> registers -> LDS
// in CHUNKS of three vectors per thread with clamped addressing — a block meets at most a few of
// them, what matters is that their register footprint stays below the 8 x 16 path's.
template <typename T, typename V, typename G, int MODE>
__device__ __forceinline__ void tile_stage_edge(const DwTiledArgs& a, const T* __restrict__ X,
                                                typename V::raw_t* __restrict__ tile,
                                                const float4* __restrict__ psm, int n, int h0,
                                                int w0, int cv) {
  constexpr int VEC = V::N, WQ = VEC / 4, DIL = G::DIL, CH = 3;
  const int mode = MODE >= 0 ? MODE : a.pro_mode;
  const int cx = threadIdx.x & (LT_CVB - 1);
  const int cvc = min(cv, a.CV - 1);
  float sc[VEC], sh[VEC];
  if (mode & PRO_AFFINE) {
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const float4 s4 = psm[(9 * LT_CVB + cx) * WQ + q], t4 = psm[(10 * LT_CVB + cx) * WQ + q];
      sc[q * 4] = s4.x; sc[q * 4 + 1] = s4.y; sc[q * 4 + 2] = s4.z; sc[q * 4 + 3] = s4.w;
      sh[q * 4] = t4.x; sh[q * 4 + 1] = t4.y; sh[q * 4 + 2] = t4.z; sh[q * 4 + 3] = t4.w;
    }
  }
#pragma unroll 1
  for (int u0 = 0; u0 < G::PER; u0 += CH) {
    typename V::raw_t raw[CH];
    bool ok[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int p = (threadIdx.x >> 3) + (u0 + k) * (LT_THREADS / LT_CVB);
      const int r = p / G::IW, c = p - r * G::IW;
      const int hi = h0 - DIL + r, wi = w0 - DIL + c;
      ok[k] = p < G::NPIX && cv < a.CV && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
      const int hic = min(max(hi, 0), a.H - 1), wic = min(max(wi, 0), a.W - 1);
      raw[k] = V::load_raw(X + (((long)n * a.H + hic) * a.W + wic) * a.ldx + cvc * VEC);
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int p = (threadIdx.x >> 3) + (u0 + k) * (LT_THREADS / LT_CVB);
      if (p < G::NPIX) {
        const int r = p / G::IW, c = p - r * G::IW;
        typename V::raw_t v = raw[k];
        if (mode != PRO_NONE) {
          float f[VEC];
          V::unpack_raw(raw[k], f);
          if (mode & PRO_AFFINE) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] = fmaf(f[i], sc[i], sh[i]);
          }
          if (mode & PRO_RELU) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] = fmaxf(f[i], 0.f);
          }
          if (mode & PRO_CLAMP6) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] = fminf(f[i], 6.f);
          }
          v = V::pack_raw(f);
        }
        if (!ok[k]) v = V::zero_raw();
        tile[(r * G::IWP + c) * LT_CVB + cx] = v;
      }
    }
  }
}

// ---- registers -> LDS: activation once per element, zero padding outside the image.
// MODE >= 0: the prologue mode as a compile-time constant (no per-vector branches; MODE 0 stores
// the raw vector); MODE < 0: a.pro_mode at run time.
template <typename T, typename V, typename G, int MODE = -1>
__device__ __forceinline__ void tile_commit_g(const DwTiledArgs& a,
                                              typename V::raw_t* __restrict__ tile,
                                              const typename V::raw_t (&raw)[G::PER],
                                              unsigned okmask, const float4* __restrict__ psm) {
  constexpr int VEC = V::N, WQ = VEC / 4;
  const int mode = MODE >= 0 ? MODE : a.pro_mode;
  const int cx = threadIdx.x & (LT_CVB - 1);
  float sc[VEC], sh[VEC];
  if (mode & PRO_AFFINE) {
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const float4 s4 = psm[(9 * LT_CVB + cx) * WQ + q], t4 = psm[(10 * LT_CVB + cx) * WQ + q];
      sc[q * 4] = s4.x; sc[q * 4 + 1] = s4.y; sc[q * 4 + 2] = s4.z; sc[q * 4 + 3] = s4.w;
      sh[q * 4] = t4.x; sh[q * 4 + 1] = t4.y; sh[q * 4 + 2] = t4.z; sh[q * 4 + 3] = t4.w;
    }
  }
#pragma unroll
  for (int u = 0; u < G::PER; ++u) {
    const int p = (threadIdx.x >> 3) + u * (LT_THREADS / LT_CVB);
    if (p < G::NPIX) {
      const int r = p / G::IW, c = p - r * G::IW;
      typename V::raw_t v = raw[u];
      if (mode != PRO_NONE) {
        float f[VEC];
        V::unpack_raw(raw[u], f);
        if (mode & PRO_AFFINE) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) f[i] = fmaf(f[i], sc[i], sh[i]);
        }
        if (mode & PRO_RELU) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) f[i] = fmaxf(f[i], 0.f);
        }
        if (mode & PRO_CLAMP6) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) f[i] = fminf(f[i], 6.f);
        }
        v = V::pack_raw(f);
      }
      if (!((okmask >> u) & 1u)) v = V::zero_raw();
      tile[(r * G::IWP + c) * LT_CVB + cx] = v;
    }
  }
}

template <typename T, typename V, int DIL, int MODE = -1>
__device__ __forceinline__ void tile_commit(const DwTiledArgs& a,
                                            typename V::raw_t* __restrict__ tile,
                                            const typename V::raw_t (&raw)[TileGeom<DIL>::PER],
                                            unsigned okmask, const float4* __restrict__ psm) {
  tile_commit_g<T, V, TileGeom<DIL>, MODE>(a, tile, raw, okmask, psm);
}

// per-block constants -> LDS: rows 0..8 = the nine taps, rows 9/10 = prologue scale / shift
// (kept out of registers: the persistent loop already carries accumulators and the next tile)
template <typename V>
__device__ __forceinline__ void stage_params(const DwTiledArgs& a, float4* __restrict__ psm,
                                             int cvb0, bool taps) {
  constexpr int VEC = V::N, WQ = VEC / 4;
  const int tid = threadIdx.x;
  if (tid < 11 * LT_CVB * WQ) {
    const int r = tid / (LT_CVB * WQ), q = tid - r * (LT_CVB * WQ);
    const int c = cvb0 * VEC + q * 4;
    float4 v = r == 9 ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < a.C) {
      if (r < 9) {
        if (taps) {
          const int tap = (a.w_layout & 2) ? 8 - r : r;
          if (a.w_layout & 1)
            v = make_float4(a.w[(c + 0) * 9 + tap], a.w[(c + 1) * 9 + tap], a.w[(c + 2) * 9 + tap],
                            a.w[(c + 3) * 9 + tap]);
          else
            v = *reinterpret_cast<const float4*>(a.w + tap * a.C + c);
        }
      } else if (a.pro_mode & PRO_AFFINE) {
        v = *reinterpret_cast<const float4*>((r == 9 ? a.sc : a.sh) + c);
      }
    }
    psm[tid] = v;
  }
}

__device__ __forceinline__ void tile_coords(const DwTiledArgs& a, int t, int& n, int& h0, int& w0) {
  const int tw = t % a.tiles_w;
  t /= a.tiles_w;
  const int th = t % a.tiles_h;
  n = t / a.tiles_h;
  h0 = th * LT_TH;
  w0 = tw * LT_TW;
}

// ------------------------------------------------------------------ forward / stride-1 dgrad
// Logical (channel block, tile lane) of this workgroup.  A block covers only 8 channel vectors
// (64 B of a bf16 pixel in the fused backward, 128 B in the forward) of every pixel it touches,
// so neighbouring CHANNEL blocks of the same tile read the other halves of the same 128-byte
// lines and the same halo rows.  Dispatch order puts consecutive workgroup ids on different
// XCDs (private L2s): with blockIdx.x = channel block every line was fetched once per XCD that
// needed a piece of it (rocprofv3: 149 MB fetched by the fused backward for 59 MB of operands).
// Here every XCD owns a contiguous range of logical ids (xcd_remap) ordered channel-block
// fastest: the blocks that share lines sit on ONE XCD and run in lockstep.
struct LtBlock { int x, y; };
__device__ __forceinline__ LtBlock lt_block() {
  const int flat = blockIdx.x + gridDim.x * blockIdx.y;
  const int L = xcd_remap(flat, gridDim.x * gridDim.y);
  LtBlock b;
  b.y = L / (int)gridDim.x;
  b.x = L - b.y * (int)gridDim.x;
  return b;
}

// One tile of the forward: halo tile -> registers -> (activation) -> LDS -> taps -> store +
// statistics.  G: tile shape; POFF: interior 8 x 16 tiles use the per-block pixel offsets.
template <typename T, typename G, int MODE, bool POFF>
__device__ __forceinline__ void dw_fwd_tile(const DwTiledArgs& a, const T* __restrict__ X,
                                            T* __restrict__ Y, uint4* __restrict__ tile,
                                            const float4* __restrict__ wsm, int n, int h0, int w0,
                                            int hlim, int cv, const int (&poff)[TileGeom<G::DIL>::PER],
                                            float (&ssum)[Vec<T>::N], float (&ssq)[Vec<T>::N]) {
  constexpr int VEC = Vec<T>::N, WQ = VEC / 4, DIL = G::DIL;
  const int tid = threadIdx.x;
  const int cx = tid & (LT_CVB - 1), pix = tid >> 3;
  const int row = pix % G::TH, strip = pix / G::TH;
  if constexpr (POFF) {
    uint4 raw[G::PER];
    unsigned okmask = 0;
    tile_issue<T, Vec<T>, DIL>(a, X, n, h0, w0, cv, raw, okmask, poff);
    __syncthreads();  // every thread is done reading the previous tile
    tile_commit_g<T, Vec<T>, G, MODE>(a, tile, raw, okmask, wsm);
  } else {
    __syncthreads();
    tile_stage_edge<T, Vec<T>, G, MODE>(a, X, tile, wsm, n, h0, w0, cv);
  }
  __syncthreads();

  float acc[4][VEC];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[j][i] = 0.f;
  // kept as a real loop: fully unrolled, the scheduler hoists all 24 LDS reads and spills
#pragma unroll 1
  for (int kh = 0; kh < 3; ++kh) {
    float wv[3][VEC];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        const float4 w4 = wsm[((kh * 3 + kw) * LT_CVB + cx) * WQ + q];
        wv[kw][q * 4] = w4.x; wv[kw][q * 4 + 1] = w4.y;
        wv[kw][q * 4 + 2] = w4.z; wv[kw][q * 4 + 3] = w4.w;
      }
    const uint4* trow = tile + ((row + kh * DIL) * G::IWP + strip * 4) * LT_CVB + cx;
#pragma unroll
    for (int q = 0; q < G::COLS; ++q) {
      float v[VEC];
      Vec<T>::unpack(trow[q * LT_CVB], v);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int j = q - kw * DIL;
        if (j >= 0 && j < 4) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[j][i] = fmaf(v[i], wv[kw][i], acc[j][i]);
        }
      }
    }
  }
  const int ho = h0 + row;
  if (cv < a.CV && ho < hlim) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int wo = w0 + strip * 4 + j;
      if (wo < a.W) {
        stg16(Y + (((long)n * a.H + ho) * a.W + wo) * a.ldy + cv * VEC, Vec<T>::pack(acc[j]));
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          ssum[i] += acc[j][i];
          ssq[i] = fmaf(acc[j][i], acc[j][i], ssq[i]);
        }
      }
    }
  }
}

// remainder tile e (index behind the 8 x 16 tiles) -> image, origin, row limit; true = bottom strip
__device__ __forceinline__ bool rem_coords(const DwTiledArgs& a, int e, int& n, int& h0, int& w0,
                                           int& hlim) {
  if (e < a.N * a.nb) {
    n = e / a.nb;
    h0 = a.hb0;
    w0 = (e - n * a.nb) * 64;
    hlim = a.H;
    return true;
  }
  e -= a.N * a.nb;
  n = e / a.nc;
  h0 = (e - n * a.nc) * 32;
  w0 = a.wc0;
  hlim = a.hc;
  return false;
}

template <typename T, int DIL, int MODE = -1>
__global__ __launch_bounds__(LT_THREADS, sizeof(T) == 2 ? 3 : 4) void dwconv_tiled_kernel(const DwTiledArgs a) {
  using G = TileGeom<DIL>;
  constexpr int VEC = Vec<T>::N;
  extern __shared__ uint4 lt_smem[];
  uint4* tile = lt_smem;
  // (taps / prologue parameters sit behind the LARGEST tile image this launch can stage)
  float4* wsm = reinterpret_cast<float4*>(lt_smem + (a.nb ? TileGeomB<DIL>::TILE_VECS
                                                       : a.nc ? TileGeomC<DIL>::TILE_VECS
                                                              : G::TILE_VECS));
  const int tid = threadIdx.x;
  const LtBlock lb = lt_block();
  const int cvb0 = lb.x * LT_CVB;
  const int cx = tid & (LT_CVB - 1);
  const int cv = cvb0 + cx;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);

  stage_params<Vec<T>>(a, wsm, cvb0, true);
  __syncthreads();
  float ssum[VEC], ssq[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) ssum[i] = ssq[i] = 0.f;

  // (a software pipeline over the tiles — the next tile's loads in flight during the taps — was
  // measured for THIS kernel: no gain on the 24 MB tensors, -10 % on the large ones, 3 blocks/CU
  // already overlap each other; the fused backward, 2 blocks/CU, gains 35 % from it)
  int poff[G::PER];
  tile_offsets<DIL>(a, poff);
  for (int t = lb.y; t < a.ntiles; t += gridDim.y) {
    int n, h0, w0, hlim = a.H;
    if (t < a.ntiles_a) {
      tile_coords(a, t, n, h0, w0);
      dw_fwd_tile<T, G, MODE, true>(a, X, Y, tile, wsm, n, h0, w0, hlim, cv, poff, ssum, ssq);
    } else if (rem_coords(a, t - a.ntiles_a, n, h0, w0, hlim)) {
      dw_fwd_tile<T, TileGeomB<DIL>, MODE, false>(a, X, Y, tile, wsm, n, h0, w0, hlim, cv, poff, ssum, ssq);
    } else {
      dw_fwd_tile<T, TileGeomC<DIL>, MODE, false>(a, X, Y, tile, wsm, n, h0, w0, hlim, cv, poff, ssum, ssq);
    }
  }

  if (a.partial != nullptr) {
    // reduce over the 32 pixel-threads of each channel vector: red[32][CVB][2*VEC] in the tile
    __syncthreads();
    float* red = reinterpret_cast<float*>(tile);
    float* mine = red + ((tid >> 3) * LT_CVB + cx) * 2 * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      mine[i] = ssum[i];
      mine[VEC + i] = ssq[i];
    }
    __syncthreads();
    if (tid < LT_CVB * 2 * VEC) {
      float tot = 0.f;
      for (int r = 0; r < LT_THREADS / LT_CVB; ++r) tot += red[r * LT_CVB * 2 * VEC + tid];
      const int lcx = tid / (2 * VEC), k = tid - lcx * 2 * VEC;
      const int which = k / VEC, ci = k - which * VEC;
      const int c = (cvb0 + lcx) * VEC + ci;
      if (c < a.C) a.partial[((long)lb.y * 2 + which) * a.C + c] = tot;
    }
  }
}

// ------------------------------------------------------------------ forward, stride 2
// The last separable conv of the Xception entry blocks (xception.py:31-33) and MobileNetV2's
// stride-2 depthwise layers ran on the strip kernel (dwconv.hip): every thread fetched its own
// neighbourhood through L1/TA — 431 MB fetched for 268 MB on block1, 160 µs where the tensor
// moves in ≈ 60.  Same scheme as dwconv_tiled_kernel: a block owns 4 x 16 OUTPUT pixels x 8
// channel vectors, stages the 9 x 33 input pixels they read ONCE (activation applied once,
// zero padding after it), taps from LDS.  A thread computes two horizontally adjacent outputs.
struct S2Geom {
  static constexpr int TH = 4, TW = 16;
  static constexpr int IH = 2 * TH + 1, IW = 2 * TW + 1, IWP = IW + 1;
  static constexpr int NPIX = IH * IW;
  static constexpr int PER = (NPIX * LT_CVB + LT_THREADS - 1) / LT_THREADS;
  static constexpr int TILE_VECS = IH * IWP * LT_CVB;
};

template <typename T, int MODE>
__global__ __launch_bounds__(LT_THREADS, 3) void dwconv_tiled_s2_kernel(const DwTiledArgs a) {
  using G = S2Geom;
  constexpr int VEC = Vec<T>::N, WQ = VEC / 4;
  extern __shared__ uint4 lt_smem[];
  uint4* tile = lt_smem;
  float4* wsm = reinterpret_cast<float4*>(lt_smem + G::TILE_VECS);
  const int tid = threadIdx.x;
  const LtBlock lb = lt_block();
  const int cvb0 = lb.x * LT_CVB;
  const int cx = tid & (LT_CVB - 1), row = (tid >> 3) & (G::TH - 1), strip = tid >> 5;
  const int cv = cvb0 + cx, cvc = min(cv, a.CV - 1);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  // a.H / a.W: INPUT size; a.tiles_h / tiles_w / ntiles: tiles of the OUTPUT (Ho x Wo)
  const int Ho = (a.H + 1) / 2, Wo = (a.W + 1) / 2;
  const int mode = MODE >= 0 ? MODE : a.pro_mode;

  stage_params<Vec<T>>(a, wsm, cvb0, true);
  __syncthreads();
  float sc[VEC], sh[VEC];
  if (mode & PRO_AFFINE) {
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const float4 s4 = wsm[(9 * LT_CVB + cx) * WQ + q], t4 = wsm[(10 * LT_CVB + cx) * WQ + q];
      sc[q * 4] = s4.x; sc[q * 4 + 1] = s4.y; sc[q * 4 + 2] = s4.z; sc[q * 4 + 3] = s4.w;
      sh[q * 4] = t4.x; sh[q * 4 + 1] = t4.y; sh[q * 4 + 2] = t4.z; sh[q * 4 + 3] = t4.w;
    }
  }
  float ssum[VEC], ssq[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) ssum[i] = ssq[i] = 0.f;

  for (int t = lb.y; t < a.ntiles; t += gridDim.y) {
    int n, ho0, wo0;
    {
      int tt = t;
      const int tw = tt % a.tiles_w;
      tt /= a.tiles_w;
      const int th = tt % a.tiles_h;
      n = tt / a.tiles_h;
      ho0 = th * G::TH;
      wo0 = tw * G::TW;
    }
    const int h0 = 2 * ho0 - 1, w0 = 2 * wo0 - 1;  // input coordinates of tile element (0, 0)
    // ---- global -> registers (interior tiles: offsets relative to the tile origin)
    uint4 raw[G::PER];
    unsigned okmask = 0;
    const bool inside = h0 >= 0 && h0 + G::IH <= a.H && w0 >= 0 && w0 + G::IW <= a.W;
    if (inside) {
      const int origin = (n * a.H + h0) * a.W + w0;
      const T* __restrict__ base = X + cvc * VEC;
#pragma unroll
      for (int u = 0; u < G::PER; ++u) {
        const int p = (tid >> 3) + u * (LT_THREADS / LT_CVB);
        const int pc = p < G::NPIX ? p : G::NPIX - 1;
        const int r = pc / G::IW, c = pc - r * G::IW;
        okmask |= (p < G::NPIX && cv < a.CV) ? (1u << u) : 0u;
        raw[u] = ldg16(base + (long)(origin + r * a.W + c) * a.ldx);
      }
    } else {
#pragma unroll
      for (int u = 0; u < G::PER; ++u) {
        const int p = (tid >> 3) + u * (LT_THREADS / LT_CVB);
        const int r = p / G::IW, c = p - r * G::IW;
        const int hi = h0 + r, wi = w0 + c;
        const bool ok = p < G::NPIX && cv < a.CV && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
        okmask |= ok ? (1u << u) : 0u;
        const int hic = min(max(hi, 0), a.H - 1), wic = min(max(wi, 0), a.W - 1);
        raw[u] = ldg16(X + (((long)n * a.H + hic) * a.W + wic) * a.ldx + cvc * VEC);
      }
    }
    __syncthreads();  // every thread is done reading the previous tile
    // ---- registers -> LDS: activation once per element, zero padding outside the image
#pragma unroll
    for (int u = 0; u < G::PER; ++u) {
      const int p = (tid >> 3) + u * (LT_THREADS / LT_CVB);
      if (p < G::NPIX) {
        const int r = p / G::IW, c = p - r * G::IW;
        uint4 v = raw[u];
        if (mode != PRO_NONE) {
          float f[VEC];
          Vec<T>::unpack(raw[u], f);
          if (mode & PRO_AFFINE) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] = fmaf(f[i], sc[i], sh[i]);
          }
          if (mode & PRO_RELU) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] = fmaxf(f[i], 0.f);
          }
          if (mode & PRO_CLAMP6) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] = fminf(f[i], 6.f);
          }
          v = Vec<T>::pack(f);
        }
        tile[(r * G::IWP + c) * LT_CVB + cx] = mask_u4(v, (okmask >> u) & 1u);
      }
    }
    __syncthreads();
    // ---- two outputs per thread: columns 2 * strip + {0, 1} of tile row `row`
    float acc[2][VEC];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[j][i] = 0.f;
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
      float wv[3][VEC];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
          const float4 w4 = wsm[((kh * 3 + kw) * LT_CVB + cx) * WQ + q];
          wv[kw][q * 4] = w4.x; wv[kw][q * 4 + 1] = w4.y;
          wv[kw][q * 4 + 2] = w4.z; wv[kw][q * 4 + 3] = w4.w;
        }
      const uint4* trow = tile + ((2 * row + kh) * G::IWP + 4 * strip) * LT_CVB + cx;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        float v[VEC];
        Vec<T>::unpack(trow[q * LT_CVB], v);
        if (q < 3) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[0][i] = fmaf(v[i], wv[q][i], acc[0][i]);
        }
        if (q >= 2) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[1][i] = fmaf(v[i], wv[q - 2][i], acc[1][i]);
        }
      }
    }
    const int ho = ho0 + row;
    if (cv < a.CV && ho < Ho) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int wo = wo0 + 2 * strip + j;
        if (wo < Wo) {
          stg16(Y + (((long)n * Ho + ho) * Wo + wo) * a.ldy + cv * VEC, Vec<T>::pack(acc[j]));
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            ssum[i] += acc[j][i];
            ssq[i] = fmaf(acc[j][i], acc[j][i], ssq[i]);
          }
        }
      }
    }
  }

  if (a.partial != nullptr) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(tile);
    float* mine = red + ((tid >> 3) * LT_CVB + cx) * 2 * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      mine[i] = ssum[i];
      mine[VEC + i] = ssq[i];
    }
    __syncthreads();
    if (tid < LT_CVB * 2 * VEC) {
      float tot = 0.f;
      for (int r = 0; r < LT_THREADS / LT_CVB; ++r) tot += red[r * LT_CVB * 2 * VEC + tid];
      const int lcx = tid / (2 * VEC), k = tid - lcx * 2 * VEC;
      const int which = k / VEC, ci = k - which * VEC;
      const int c = (cvb0 + lcx) * VEC + ci;
      if (c < a.C) a.partial[((long)lb.y * 2 + which) * a.C + c] = tot;
    }
  }
}

// ------------------------------------------------------------------ weight gradient
// dW[kh,kw,c] = sum_p dy[p,c] * act(x)[p + (kh-1)d, (kw-1)d, c]: same tile, the thread's four dy
// vectors come straight from global memory (each is used by one thread only).
template <typename T, int DIL>
__global__ __launch_bounds__(LT_THREADS, 2) void dwconv_wgrad_tiled_kernel(const DwTiledArgs a) {
  using G = TileGeom<DIL>;
  constexpr int VEC = Vec<T>::N;
  extern __shared__ uint4 lt_smem[];
  uint4* tile = lt_smem;
  const int tid = threadIdx.x;
  const LtBlock lb = lt_block();
  const int cvb0 = lb.x * LT_CVB;
  const int cx = tid & (LT_CVB - 1), row = (tid >> 3) & (LT_TH - 1), strip = tid >> 6;
  const int cv = cvb0 + cx;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ DY = reinterpret_cast<const T*>(a.dy);

  float4* psm = reinterpret_cast<float4*>(lt_smem + G::TILE_VECS);
  stage_params<Vec<T>>(a, psm, cvb0, false);
  __syncthreads();
  float acc[9][VEC];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[k][i] = 0.f;

  int poff[G::PER];
  tile_offsets<DIL>(a, poff);
  for (int t = lb.y; t < a.ntiles; t += gridDim.y) {
    int n, h0, w0;
    tile_coords(a, t, n, h0, w0);
    uint4 raw[G::PER];
    unsigned okmask;
    tile_issue<T, Vec<T>, DIL>(a, X, n, h0, w0, cv, raw, okmask, poff);
    // this thread's four dy vectors (zero outside the image / channel range)
    const int ho = h0 + row;
    uint4 graw[4];
    const bool rok = cv < a.CV && ho < a.H;
    const int hoc = min(ho, a.H - 1), cvc = min(cv, a.CV - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int wo = w0 + strip * 4 + j;
      graw[j] = ldg16(DY + (((long)n * a.H + hoc) * a.W + min(wo, a.W - 1)) * a.lddy + cvc * VEC);
      if (!(rok && wo < a.W)) graw[j] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    tile_commit<T, Vec<T>, DIL>(a, tile, raw, okmask, psm);
    __syncthreads();
    float g[4][VEC];
#pragma unroll
    for (int j = 0; j < 4; ++j) Vec<T>::unpack(graw[j], g[j]);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const uint4* trow = tile + ((row + kh * DIL) * G::IWP + strip * 4) * LT_CVB + cx;
#pragma unroll
      for (int q = 0; q < G::COLS; ++q) {
        float v[VEC];
        Vec<T>::unpack(trow[q * LT_CVB], v);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int j = q - kw * DIL;
          if (j >= 0 && j < 4) {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
              acc[kh * 3 + kw][i] = fmaf(v[i], g[j][i], acc[kh * 3 + kw][i]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads of the next row below this point
    }
  }

  // block reduction: across the 8 rows of a wave by lane exchange, across the 4 strips (waves)
  // through LDS: red[4][CVB][9*VEC]
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float v = acc[k][i];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      acc[k][i] = v;
    }
  __syncthreads();
  float* red = reinterpret_cast<float*>(tile);
  if (row == 0) {
    float* mine = red + (strip * LT_CVB + cx) * 9 * VEC;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int i = 0; i < VEC; ++i) mine[k * VEC + i] = acc[k][i];
  }
  __syncthreads();
  for (int e = tid; e < LT_CVB * 9 * VEC; e += LT_THREADS) {
    float tot = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) tot += red[s * LT_CVB * 9 * VEC + e];
    const int lcx = e / (9 * VEC), k = e - lcx * 9 * VEC;
    const int tap = k / VEC, ci = k - tap * VEC;
    const int c = (cvb0 + lcx) * VEC + ci;
    if (c < a.C) a.partial[((long)lb.y * 9 + tap) * a.C + c] = tot;
  }
}


// ------------------------------------------------------------------ fused backward
// One pass over (dy, x) for the whole depthwise backward (stride 1):
//   g'[q]   = relu_mask(x[q]) * sum_k dy[q - d_k] * w[k]      data gradient wrt act(x), masked
//   dW[k]  += act(x[q]) * dy[q - d_k]                          the SAME shifted dy values
//   (sum g', sum g' * x)                                       BatchNorm-backward sums of the
//                                                              producer's BN (if any)
// The dy tile (+halo) is staged in LDS exactly like the forward input tile; LDS row r = kh*3+kw
// of the staged taps holds w[8 - r] (flipped), and the value read at tile offset (kh, kw) pairs
// with tap 8 - r of the weight gradient.  Replaces dgrad + wgrad + bn_bwd_reduce (three passes
// over two tensors each).
// RES: the second gradient of a forked activation (an Xception block input feeds the residual
// sum AND the first separable conv, xception.py:40-42) is added in the store path — the
// element-wise add autograd would launch for it (2 reads + 1 write of the tensor) is gone.
// x vectors of one 4-pixel strip (one 64-bit pixel offset per tile and thread, + j * ld: the four
// vectors used to rebuild ((n*H + h)*W + w)*ld each — with the residual loads and the stores ~150
// VALU of address arithmetic per tile next to ~250 of tap work; only the last strip of a
// right-border tile, a wave-uniform case, needs the column clamp)
template <typename T, typename V>
__device__ __forceinline__ void dw_bwd_issue_x(const DwTiledArgs& a, const T* __restrict__ X, int n,
                                               int h0, int w0, int row, int strip, int cv,
                                               typename V::raw_t (&xr)[4]) {
  constexpr int VEC = V::N;
  const int hoc = min(h0 + row, a.H - 1), cvc = min(cv, a.CV - 1);
  const int wb = w0 + strip * 4;
  const T* __restrict__ px = X + ((long)(n * a.H + hoc) * a.W + min(wb, a.W - 1)) * a.ldx + cvc * VEC;
  if (wb + 3 < a.W) {
#pragma unroll
    for (int j = 0; j < 4; ++j) xr[j] = V::load_raw(px + j * (int)a.ldx);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      xr[j] = V::load_raw(px + (min(wb + j, a.W - 1) - min(wb, a.W - 1)) * (int)a.ldx);
  }
}

// The arithmetic of one tile of the fused backward: the dy halo tile is in LDS (tile), this
// thread's four x vectors in xraw.  G: tile shape (8 x 16, or a remainder shape).
template <typename T, typename G, bool RES>
__device__ __forceinline__ void dw_bwd_compute(
    const DwTiledArgs& a, const typename HVec<T>::raw_t* __restrict__ tile,
    const float4* __restrict__ psm, T* __restrict__ GO, int n, int h0, int w0, int hlim, int row,
    int strip, int cx, int cv, const typename HVec<T>::raw_t (&xraw)[4],
    const float (&sc)[HVec<T>::N], const float (&sh)[HVec<T>::N], float (&accw)[9][HVec<T>::N],
    float (&s1)[HVec<T>::N], float (&s2)[HVec<T>::N]) {
  using V = HVec<T>;
  using raw_t = typename V::raw_t;
  constexpr int VEC = V::N, WQ = VEC / 4, DIL = G::DIL;
  const int ho = h0 + row;
  const bool rok = cv < a.CV && ho < hlim;

  float xa[4][VEC];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    V::unpack_raw(xraw[j], xa[j]);
    if (a.pro_mode & PRO_AFFINE) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) xa[j][i] = fmaf(xa[j][i], sc[i], sh[i]);
    }
    if (a.pro_mode & PRO_RELU) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) xa[j][i] = fmaxf(xa[j][i], 0.f);
    }
    if (a.pro_mode & PRO_CLAMP6) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) xa[j][i] = fminf(xa[j][i], 6.f);
    }
    const bool ok = rok && (w0 + strip * 4 + j) < a.W;
    if (!ok) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) xa[j][i] = 0.f;
    }
  }
  float accg[4][VEC];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < VEC; ++i) accg[j][i] = 0.f;
  raw_t rres[4];
  if (RES) {  // in flight during the tap loop
    const T* __restrict__ R = reinterpret_cast<const T*>(a.res);
    const int hoc = min(ho, a.H - 1), cvc = min(cv, a.CV - 1);
    const int wb = w0 + strip * 4, wbc = min(wb, a.W - 1);
    const T* __restrict__ pr = R + ((long)(n * a.H + hoc) * a.W + wbc) * a.ldr + cvc * VEC;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      rres[j] = V::load_raw(pr + (min(wb + j, a.W - 1) - wbc) * (int)a.ldr);
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    float wv[3][VEC];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        const float4 w4 = psm[((kh * 3 + kw) * LT_CVB + cx) * WQ + q];
        wv[kw][q * 4] = w4.x; wv[kw][q * 4 + 1] = w4.y;
        wv[kw][q * 4 + 2] = w4.z; wv[kw][q * 4 + 3] = w4.w;
      }
    const raw_t* trow = tile + ((row + kh * DIL) * G::IWP + strip * 4) * LT_CVB + cx;
#pragma unroll
    for (int q = 0; q < G::COLS; ++q) {
      float v[VEC];
      V::unpack_raw(trow[q * LT_CVB], v);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int j = q - kw * DIL;
        if (j >= 0 && j < 4) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            accg[j][i] = fmaf(v[i], wv[kw][i], accg[j][i]);
            accw[kh * 3 + kw][i] = fmaf(v[i], xa[j][i], accw[kh * 3 + kw][i]);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // mask, store, BatchNorm-backward sums
  T* __restrict__ pgo = GO + ((long)(n * a.H + min(ho, a.H - 1)) * a.W + (w0 + strip * 4)) * a.ldy + cv * VEC;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int wo = w0 + strip * 4 + j;
    if (rok && wo < a.W) {
      float xr[VEC];
      V::unpack_raw(xraw[j], xr);
      if (a.pro_mode & PRO_RELU) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const bool on = xa[j][i] > 0.f && (!(a.pro_mode & PRO_CLAMP6) || xa[j][i] < 6.f);
          accg[j][i] = on ? accg[j][i] : 0.f;
        }
      }
      if (RES) {
        float rr[VEC], o[VEC];
        V::unpack_raw(rres[j], rr);
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = accg[j][i] + rr[i];
        V::store(pgo + j * (int)a.ldy, o);
      } else {
        V::store(pgo + j * (int)a.ldy, accg[j]);
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        s1[i] += accg[j][i];
        s2[i] = fmaf(accg[j][i], xr[i], s2[i]);
      }
    }
  }
}

// one REMAINDER tile of the fused backward (no software pipeline: a block meets at most a few)
template <typename T, typename G, bool RES>
__device__ __forceinline__ void dw_bwd_edge_tile(
    const DwTiledArgs& a, const DwTiledArgs& dyargs,
    typename HVec<T>::raw_t* __restrict__ tile, const float4* __restrict__ psm, int n, int h0,
    int w0, int hlim, int cx, int cv, const float (&sc)[HVec<T>::N], const float (&sh)[HVec<T>::N],
    float (&accw)[9][HVec<T>::N], float (&s1)[HVec<T>::N], float (&s2)[HVec<T>::N]) {
  using V = HVec<T>;
  using raw_t = typename V::raw_t;
  const int pix = threadIdx.x >> 3, row = pix % G::TH, strip = pix / G::TH;
  raw_t xraw[4];
  dw_bwd_issue_x<T, V>(a, reinterpret_cast<const T*>(a.x), n, h0, w0, row, strip, cv, xraw);
  __syncthreads();
  tile_stage_edge<T, V, G, PRO_NONE>(dyargs, reinterpret_cast<const T*>(a.dy), tile, psm, n, h0, w0,
                                     cv);
  __syncthreads();
  dw_bwd_compute<T, G, RES>(a, tile, psm, reinterpret_cast<T*>(a.y), n, h0, w0, hlim, row, strip, cx,
                            cv, xraw, sc, sh, accw, s1, s2);
}

// REM: this launch has remainder tiles (a separate instance: with their code compiled in, the
// 8 x 16 loop of the classic tiling spilled 12 bytes and ran 3-5 % slower on the large maps)
template <typename T, int DIL, bool RES = false, bool REM = false>
__global__ __launch_bounds__(LT_THREADS, 2) void dwconv_bwd_tiled_kernel(const DwTiledArgs a) {
  // 4 channels per thread in both element types (8-byte bf16 vectors): this kernel carries nine
  // tap accumulators per channel on top of the data-gradient accumulators
  using G = TileGeom<DIL>;
  using V = HVec<T>;
  using raw_t = typename V::raw_t;
  constexpr int VEC = V::N, WQ = VEC / 4;
  extern __shared__ uint4 lt_smem[];
  raw_t* tile = reinterpret_cast<raw_t*>(lt_smem);
  // (parameter rows behind the largest tile image of the launch, tiled_lds)
  float4* psm = reinterpret_cast<float4*>(
      lt_smem + (!REM ? G::TILE_VECS : a.nb ? TileGeomB<DIL>::TILE_VECS
                                            : a.nc ? TileGeomC<DIL>::TILE_VECS : G::TILE_VECS));
  const int tid = threadIdx.x;
  const LtBlock lb = lt_block();
  const int cvb0 = lb.x * LT_CVB;
  const int cx = tid & (LT_CVB - 1), row = (tid >> 3) & (LT_TH - 1), strip = tid >> 6;
  const int cv = cvb0 + cx;
  const T* __restrict__ DY = reinterpret_cast<const T*>(a.dy);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  T* __restrict__ GO = reinterpret_cast<T*>(a.y);

  // taps (flipped) + the prologue of x; the dy tile itself is staged without activation
  stage_params<V>(a, psm, cvb0, true);
  __syncthreads();
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int q = 0; q < WQ; ++q) {
    const float4 s4 = psm[(9 * LT_CVB + cx) * WQ + q], t4 = psm[(10 * LT_CVB + cx) * WQ + q];
    sc[q * 4] = s4.x; sc[q * 4 + 1] = s4.y; sc[q * 4 + 2] = s4.z; sc[q * 4 + 3] = s4.w;
    sh[q * 4] = t4.x; sh[q * 4 + 1] = t4.y; sh[q * 4 + 2] = t4.z; sh[q * 4 + 3] = t4.w;
  }
  DwTiledArgs plain = a;  // dy is staged raw
  plain.pro_mode = PRO_NONE;

  float accw[9][VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) accw[k][i] = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;

  // software pipeline over the block's 8 x 16 tiles (see dwconv_tiled_kernel): the next tile's dy
  // halo tile and x vectors are in flight while the current tile is computed
  DwTiledArgs dyargs = plain;
  dyargs.ldx = a.lddy;
  constexpr bool PIPE = DIL == 1;  // dilation 2 (wider halo tile) would spill with the prefetch
  int t = lb.y, nn = 0, nh0 = 0, nw0 = 0;
  int poff[G::PER];
  tile_offsets<DIL>(a, poff);
  raw_t raw[G::PER], xnext[4];
  unsigned okmask = 0;
  if (PIPE && t < a.ntiles_a) {
    tile_coords(a, t, nn, nh0, nw0);
    tile_issue<T, V, DIL>(dyargs, DY, nn, nh0, nw0, cv, raw, okmask, poff);
    dw_bwd_issue_x<T, V>(a, X, nn, nh0, nw0, row, strip, cv, xnext);
  }
  while (t < a.ntiles_a) {
    if (!PIPE) {
      tile_coords(a, t, nn, nh0, nw0);
      tile_issue<T, V, DIL>(dyargs, DY, nn, nh0, nw0, cv, raw, okmask, poff);
      dw_bwd_issue_x<T, V>(a, X, nn, nh0, nw0, row, strip, cv, xnext);
    }
    __syncthreads();
    tile_commit<T, V, DIL, PRO_NONE>(plain, tile, raw, okmask, psm);
    const int n = nn, h0 = nh0, w0 = nw0;
    raw_t xraw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xraw[j] = xnext[j];
    t += gridDim.y;
    if (PIPE && t < a.ntiles_a) {
      tile_coords(a, t, nn, nh0, nw0);
      tile_issue<T, V, DIL>(dyargs, DY, nn, nh0, nw0, cv, raw, okmask, poff);
      dw_bwd_issue_x<T, V>(a, X, nn, nh0, nw0, row, strip, cv, xnext);
    }
    __syncthreads();
    dw_bwd_compute<T, G, RES>(a, tile, psm, GO, n, h0, w0, a.H, row, strip, cx, cv, xraw, sc, sh,
                              accw, s1, s2);
  }
  // remainder tiles (t continues behind the 8 x 16 tiles with the same stride)
  if constexpr (REM)
  for (; t < a.ntiles; t += gridDim.y) {
    int n, h0, w0, hlim;
    if (rem_coords(a, t - a.ntiles_a, n, h0, w0, hlim))
      dw_bwd_edge_tile<T, TileGeomB<DIL>, RES>(a, dyargs, tile, psm, n, h0, w0, hlim, cx, cv,
                                               sc, sh, accw, s1, s2);
    else
      dw_bwd_edge_tile<T, TileGeomC<DIL>, RES>(a, dyargs, tile, psm, n, h0, w0, hlim, cx, cv,
                                               sc, sh, accw, s1, s2);
  }

  // ---- block reductions (rows of a wave by lane exchange, strips through LDS)
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float v = accw[k][i];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      accw[k][i] = v;
    }
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    s1[i] += __shfl_xor(s1[i], 8, 64);  s1[i] += __shfl_xor(s1[i], 16, 64);
    s1[i] += __shfl_xor(s1[i], 32, 64);
    s2[i] += __shfl_xor(s2[i], 8, 64);  s2[i] += __shfl_xor(s2[i], 16, 64);
    s2[i] += __shfl_xor(s2[i], 32, 64);
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(lt_smem);  // [4 strips][CVB][11 * VEC]
  if (row == 0) {
    float* mine = red + (strip * LT_CVB + cx) * 11 * VEC;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int i = 0; i < VEC; ++i) mine[k * VEC + i] = accw[k][i];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      mine[9 * VEC + i] = s1[i];
      mine[10 * VEC + i] = s2[i];
    }
  }
  __syncthreads();
  for (int e = tid; e < LT_CVB * 11 * VEC; e += LT_THREADS) {
    float tot = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) tot += red[s * LT_CVB * 11 * VEC + e];
    const int lcx = e / (11 * VEC), k = e - lcx * 11 * VEC;
    const int r = k / VEC, ci = k - r * VEC;
    const int c = (cvb0 + lcx) * VEC + ci;
    if (c < a.C) {
      if (r < 9) a.partial[((long)lb.y * 9 + (8 - r)) * a.C + c] = tot;  // LDS row r = tap 8-r
      else if (a.partial_bn != nullptr) a.partial_bn[((long)lb.y * 2 + (r - 9)) * a.C + c] = tot;
    }
  }
}

// partial [R][9][C] -> dW [C][9] (torch's [C,1,3,3]): fixed-order fp64 column sums + transpose in
// one launch (was: two-level colsum + a torch transpose copy)
__global__ __launch_bounds__(256) void dw_wgrad_finalize_kernel(const float* __restrict__ part,
                                                                int R, int C,
                                                                float* __restrict__ out) {
  // block = 8 columns x 32 row groups: short serial loops and 9C/8 blocks in flight
  __shared__ double red[32][9];
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int col = blockIdx.x * 8 + cx;  // column of the [9*C] row: tap * C + c
  const int L = 9 * C;
  double acc = 0.0;
  if (col < L)
    for (int r = ry; r < R; r += 32) acc += (double)part[(long)r * L + col];
  red[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && col < L) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][cx];
    const int tap = col / C, c = col - tap * C;
    out[c * 9 + tap] = (float)t;
  }
}

int launch_dw_wgrad_finalize(const float* partial, int R, int C, float* out, hipStream_t st) {
  hipLaunchKernelGGL(dw_wgrad_finalize_kernel, dim3((9 * C + 7) / 8), dim3(256), 0, st, partial,
                     R, C, out);
  return check_launch("dw_wgrad_finalize");
}

// ------------------------------------------------------------------ host side
bool dw_tiled_supported(int stride, int dil) { return stride == 1 && (dil == 1 || dil == 2); }

// rem: the r05 tiling with remainder tiles (forward / fused backward); the weight-gradient kernel
// keeps the classic one
// rem_pct: minimum share (percent) of the classic tile count the remainder tiles must save
// (the fused backward: 6 — below that its pipelined 8 x 16 loop on the classic tiling wins)
static void tiled_geom(DwTiledArgs& a, int dtype, int N, int H, int W, int C, bool rem = false,
                       int rem_pct = 0) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  a.N = N; a.H = H; a.W = W; a.C = C; a.CV = C / vec;
  const int full_h = H / LT_TH, full_w = W / LT_TW, rb = H % LT_TH, rr = W % LT_TW;
  bool use_b = rem && rb > 0 && rb <= 2 && full_h > 0;
  bool use_c = rem && rr > 0 && rr <= 4 && full_w > 0;
  if (use_b || use_c) {
    const long classic = (long)((H + LT_TH - 1) / LT_TH) * ((W + LT_TW - 1) / LT_TW);
    const int hc = use_b ? full_h * LT_TH : H;
    const long now = (long)(use_b ? full_h : (H + LT_TH - 1) / LT_TH) *
                         (use_c ? full_w : (W + LT_TW - 1) / LT_TW) +
                     (use_b ? (W + 63) / 64 : 0) + (use_c ? (hc + 31) / 32 : 0);
    if ((classic - now) * 100 < (long)rem_pct * classic) use_b = use_c = false;
  }
  a.tiles_h = use_b ? full_h : (H + LT_TH - 1) / LT_TH;
  a.tiles_w = use_c ? full_w : (W + LT_TW - 1) / LT_TW;
  a.hb0 = full_h * LT_TH;
  a.wc0 = full_w * LT_TW;
  a.hc = use_b ? a.hb0 : H;
  a.nb = use_b ? (W + 63) / 64 : 0;
  a.nc = use_c ? (a.hc + 31) / 32 : 0;
  a.ntiles_a = N * a.tiles_h * a.tiles_w;
  a.ntiles = a.ntiles_a + N * (a.nb + a.nc);
}

// Persistent blocks: about one resident set of blocks for the whole launch (3 per CU forward, 2 per
// CU fused backward), each walking several tiles so that the tile pipeline has something to
// overlap; the count also bounds the number of partial rows.
int dw_tiled_grid_y(int dtype, int C, int N, int H, int W, int kind) {
  // kind 0: forward / dgrad (one tile per block where possible: 3 blocks/CU overlap each other),
  // 1: fused backward (two resident blocks per CU, several tiles each for the tile pipeline),
  // 2: weight gradient (few partial rows: its block reduction is the expensive part)
  DwTiledArgs a;
  tiled_geom(a, dtype, N, H, W, C, kind != 2, kind == 1 ? DW_BWD_REM_PCT : 0);
  const int cv = kind == 1 ? C / 4 : a.CV;  // the fused backward works on 4-channel vectors
  const int gx = (cv + LT_CVB - 1) / LT_CVB;
  // measured (tools/lab/op_time.py, [2,65,129,728] bf16): forward 768 / 1024 / 1536 / 2048
  // blocks = 23.0 / 21.6 / 28.0 / 22.5 us; fused backward 256 / 384 / 512 / 768 / 1024 =
  // 43.9 / 44.5 / 36.3 / 38.8 / 39.6 us
  constexpr int fwd_cap = 2048, bwd_cap = 512;
  long cap = (kind == 0 ? fwd_cap : kind == 1 ? bwd_cap : 768) / gx;
  if (cap < 1) cap = 1;
  long gy = a.ntiles;
  if (gy > cap) gy = cap;
  return (int)gy;
}

template <int DIL> static size_t tiled_lds(int dtype, bool with_weights, int nb = 0, int nc = 0) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  // (the kernels place the parameter rows behind the largest tile image of the launch)
  size_t b = (size_t)(nb ? TileGeomB<DIL>::TILE_VECS : nc ? TileGeomC<DIL>::TILE_VECS
                                                          : TileGeom<DIL>::TILE_VECS) * 16;
  const size_t red_fwd = (size_t)LT_THREADS * 2 * vec * sizeof(float);
  const size_t red_wg = (size_t)4 * LT_CVB * 11 * vec * sizeof(float);
  if (b < red_fwd) b = red_fwd;
  if (b < red_wg) b = red_wg;
  (void)with_weights;
  b += (size_t)11 * LT_CVB * vec * sizeof(float);  // taps + prologue scale / shift
  return b;
}

// ---- stride 2 (pad 1, dilation 1): the launcher takes the INPUT size H x W, tiles cover the
// output ((H + 1) / 2 x (W + 1) / 2); the grid query takes the output size
int dw_tiled_s2_grid_y(int dtype, int C, int N, int Ho, int Wo) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  const long ntiles = (long)N * ((Ho + S2Geom::TH - 1) / S2Geom::TH) * ((Wo + S2Geom::TW - 1) / S2Geom::TW);
  const int gx = (C / vec + LT_CVB - 1) / LT_CVB;
  long cap = 2048 / gx;
  if (cap < 1) cap = 1;
  return (int)(ntiles < cap ? ntiles : cap);
}

int launch_dw_tiled_s2(int dtype, const void* x, long ldx, int N, int H, int W, int C,
                       const float* w, int w_layout, int pro_mode, const float* sc,
                       const float* sh, void* y, long ldy, float* stat_partial, int grid_y,
                       hipStream_t st) {
  DwTiledArgs a;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  a.N = N; a.H = H; a.W = W; a.C = C; a.CV = C / vec;
  a.tiles_h = (Ho + S2Geom::TH - 1) / S2Geom::TH;
  a.tiles_w = (Wo + S2Geom::TW - 1) / S2Geom::TW;
  a.ntiles = N * a.tiles_h * a.tiles_w;
  a.w_layout = w_layout;
  a.x = x; a.w = w; a.y = y; a.dy = nullptr; a.sc = sc; a.sh = sh; a.partial = stat_partial;
  a.partial_bn = nullptr; a.res = nullptr; a.ldr = 0;
  a.ldx = ldx; a.ldy = ldy; a.lddy = 0; a.pro_mode = pro_mode;
  const dim3 grid((a.CV + LT_CVB - 1) / LT_CVB, grid_y);
  size_t lds = (size_t)S2Geom::TILE_VECS * 16;
  const size_t red = (size_t)LT_THREADS * 2 * vec * sizeof(float);
  if (lds < red) lds = red;
  lds += (size_t)11 * LT_CVB * vec * sizeof(float);
#define SEG_S2(TT, MM) \
  hipLaunchKernelGGL((dwconv_tiled_s2_kernel<TT, MM>), grid, dim3(LT_THREADS), lds, st, a)
  if (dtype == DT_BF16) {
    switch (pro_mode) {
      case PRO_RELU: SEG_S2(bf16_t, PRO_RELU); break;
      case PRO_AFFINE_RELU: SEG_S2(bf16_t, PRO_AFFINE_RELU); break;
      default: SEG_S2(bf16_t, -1); break;
    }
  } else {
    SEG_S2(float, -1);
  }
#undef SEG_S2
  return check_launch("dwconv3x3 (tiled, stride 2)");
}

int launch_dw_tiled(int dtype, const void* x, long ldx, int N, int H, int W, int C,
                    const float* w, int w_layout, int dil, int pro_mode, const float* sc,
                    const float* sh, void* y, long ldy, float* stat_partial, int grid_y,
                    hipStream_t st) {
  DwTiledArgs a;
  tiled_geom(a, dtype, N, H, W, C, true);
  a.w_layout = w_layout;
  a.x = x; a.w = w; a.y = y; a.dy = nullptr; a.sc = sc; a.sh = sh; a.partial = stat_partial;
  a.partial_bn = nullptr; a.res = nullptr; a.ldr = 0;
  a.ldx = ldx; a.ldy = ldy; a.lddy = 0; a.pro_mode = pro_mode;
  const dim3 grid((a.CV + LT_CVB - 1) / LT_CVB, grid_y);
#define SEG_LT(TT, DD, MM) \
  hipLaunchKernelGGL((dwconv_tiled_kernel<TT, DD, MM>), grid, dim3(LT_THREADS), \
                     tiled_lds<DD>(dtype, true, a.nb, a.nc), st, a)
  if (dtype == DT_BF16 && dil == 1) {
    // the prologue modes of the networks as compile-time constants (no per-vector branches)
    switch (pro_mode) {
      case PRO_NONE: SEG_LT(bf16_t, 1, PRO_NONE); break;
      case PRO_RELU: SEG_LT(bf16_t, 1, PRO_RELU); break;
      case PRO_AFFINE: SEG_LT(bf16_t, 1, PRO_AFFINE); break;
      case PRO_AFFINE_RELU: SEG_LT(bf16_t, 1, PRO_AFFINE_RELU); break;
      default: SEG_LT(bf16_t, 1, -1); break;
    }
  } else if (dtype == DT_BF16) {
    SEG_LT(bf16_t, 2, -1);
  } else {
    if (dil == 1) SEG_LT(float, 1, -1); else SEG_LT(float, 2, -1);
  }
#undef SEG_LT
  return check_launch("dwconv3x3 (tiled)");
}

int launch_dw_wgrad_tiled(int dtype, const void* x, long ldx, int N, int H, int W, int C,
                          const void* dy, long lddy, int dil, int pro_mode, const float* sc,
                          const float* sh, float* partial, int grid_y, hipStream_t st) {
  DwTiledArgs a;
  tiled_geom(a, dtype, N, H, W, C);
  a.w_layout = 0;
  a.x = x; a.w = nullptr; a.y = nullptr; a.dy = dy; a.sc = sc; a.sh = sh; a.partial = partial;
  a.partial_bn = nullptr; a.res = nullptr; a.ldr = 0;
  a.ldx = ldx; a.ldy = 0; a.lddy = lddy; a.pro_mode = pro_mode;
  const dim3 grid((a.CV + LT_CVB - 1) / LT_CVB, grid_y);
#define SEG_LT(TT, DD) \
  hipLaunchKernelGGL((dwconv_wgrad_tiled_kernel<TT, DD>), grid, dim3(LT_THREADS), \
                     tiled_lds<DD>(dtype, false), st, a)
  if (dtype == DT_BF16) { if (dil == 1) SEG_LT(bf16_t, 1); else SEG_LT(bf16_t, 2); }
  else { if (dil == 1) SEG_LT(float, 1); else SEG_LT(float, 2); }
#undef SEG_LT
  return check_launch("dwconv3x3_wgrad (tiled)");
}


int launch_dw_bwd_tiled(int dtype, const void* dy, long lddy, const void* x, long ldx, int N, int H,
                        int W, int C, const float* w, int w_layout, int dil, int pro_mode,
                        const float* sc, const float* sh, void* g, long ldg, float* partial_w,
                        float* partial_bn, int grid_y, hipStream_t st, const void* res, long ldr) {
  DwTiledArgs a;
  tiled_geom(a, dtype, N, H, W, C, true, DW_BWD_REM_PCT);
  a.CV = C / 4;               // HVec: 4 channels per thread in both element types
  a.w_layout = w_layout ^ 2;  // taps staged flipped (bit 1 toggles the caller's orientation)
  a.x = x; a.w = w; a.y = g; a.dy = dy; a.sc = sc; a.sh = sh;
  a.partial = partial_w; a.partial_bn = partial_bn;
  a.res = res; a.ldr = ldr;
  a.ldx = ldx; a.ldy = ldg; a.lddy = lddy; a.pro_mode = pro_mode;
  const dim3 grid((a.CV + LT_CVB - 1) / LT_CVB, grid_y);
#define SEG_LT(TT, DD, RR)                                                                       \
  do {                                                                                           \
    if (a.nb || a.nc)                                                                            \
      hipLaunchKernelGGL((dwconv_bwd_tiled_kernel<TT, DD, RR, true>), grid, dim3(LT_THREADS),    \
                         tiled_lds<DD>(dtype, true, a.nb, a.nc), st, a);                         \
    else                                                                                         \
      hipLaunchKernelGGL((dwconv_bwd_tiled_kernel<TT, DD, RR, false>), grid, dim3(LT_THREADS),   \
                         tiled_lds<DD>(dtype, true), st, a);                                     \
  } while (0)
  if (res != nullptr) {  // (dilation 1 only: dw_bwd_tiled_res_supported)
    if (dtype == DT_BF16) SEG_LT(bf16_t, 1, true); else SEG_LT(float, 1, true);
  } else if (dtype == DT_BF16) { if (dil == 1) SEG_LT(bf16_t, 1, false); else SEG_LT(bf16_t, 2, false); }
  else { if (dil == 1) SEG_LT(float, 1, false); else SEG_LT(float, 2, false); }
#undef SEG_LT
  return check_launch("dwconv3x3_bwd_fused (tiled)");
}

}  // namespace seg
