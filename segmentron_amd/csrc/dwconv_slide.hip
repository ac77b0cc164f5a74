// Register-sliding depthwise 3x3 (stride 1, dilation 1): forward (+ BatchNorm statistics) and the
// fused backward (masked data gradient + weight-gradient partials + BatchNorm-backward sums), r06.
// Reference call sites: segmentron/modules/basic.py:38-40,152-153 (SeparableConv2d's depthwise
// conv), xception.py:27-42.
//
// Why a third generation.  The LDS-tiled kernels (dwconv_tiled.hip) issue 8.5 M / 12.7 M wave
// instructions for the 12.2 M outputs of the [2,65,129,728] middle-flow map and spend 2/3 of
// their wave-cycles parked (profiles/r03_pmc_depthwise.md: wait_any 38 / 32 %, issue stalls
// 26 / 21 %): global -> registers -> activation -> repack -> LDS -> barrier -> unpack per tap,
// two barriers per tile, 2-3 waves per SIMD.  Their VALU *cycles* are a sixth of the launch — the
// kernels are bound by dependent latencies and instruction count, not by HBM and not by the ALUs.
// Here nothing goes through LDS and there is no barrier in the loop:
//   * a thread owns ONE image column x 4 channels and slides down a strip of rows with a 3 x 3
//     window of activated fp32 vectors in registers; per output row it loads the three vectors of
//     the NEXT input row (w-1, w, w+1: the neighbours' loads hit L1), applies the producer's
//     BatchNorm + ReLU once per loaded vector, and issues the nine taps (18 v_pk_fma_f32);
//   * the next row's loads are requested before the current row is computed (one row in flight
//     per wave, 4 waves per SIMD in the forward);
//   * weights (9 x 4 fp32) and the prologue parameters live in registers for the whole strip;
//   * statistics / tap accumulators stay in registers and are reduced once per block.
// Instruction count per output vector of four channels: ~70 forward / ~100 backward, against
// ~185 / ~280 for the tiled kernels (hipcc -S).  The activated operand is no longer rounded to the
// storage dtype before the taps (the tiled kernels parked it in LDS as bf16): one rounding less.
//
// Block = 256 threads = 16 channel quads (64 channels: one 128-byte line of a bf16 pixel) x 16
// columns; grid = channel blocks x column blocks x row strips x images, channel block fastest
// inside an XCD's contiguous range (xcd_remap).  One partial row per (image, strip, column block).
#include <type_traits>
#include "common.h"
#include "dwconv_slide.h"

namespace seg {

constexpr int SL_THREADS = 256, SL_CQ = 16, SL_WL = 16;
// rows in flight per thread / waves per SIMD (tools/lab sweeps, profiles/r06_dw_slide.md: depth 2 .. 6
// and 2 .. 4 waves are within 5 % of each other once the loop is branch-free; 2 / 4 / 2 shipped)
constexpr int SL_DEPTH_F = 2, SL_OCC_F = 4, SL_DEPTH_B = 2;
#define SL_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)

struct DwSlideArgs {
  const void* x;        // fwd: input; bwd: the forward input (raw tensor + prologue)
  const void* dy;       // bwd: gradient wrt the depthwise output
  const void* res;      // bwd: tensor added to the masked data gradient in the store (or null)
  void* y;              // fwd: output; bwd: data gradient
  const float* w;       // forward taps: [9][C] tap-major (w_layout bit 0 clear) or torch's [C][9];
  int w_layout;         // bit 1 (fwd only): use tap 8-k for tap k (stride-1 data gradient)
  const float* sc; const float* sh;
  float* partial;       // fwd: [rows][2][C] statistics or null; bwd: [rows][9][C]
  float* partial_bn;    // bwd: [rows][2][C] or null
  long ldx, lddy, ldy, ldr;
  int N, H, W, C, pro_mode, rs, nstrips, nwblk, ncblk;
};

struct SlBlock { int n, strip, wblk, cblk, prow; };
__device__ __forceinline__ SlBlock sl_block(const DwSlideArgs& a) {
  const int nblk = a.ncblk * a.nwblk * a.nstrips * a.N;
  int L = xcd_remap(blockIdx.x, nblk);
  SlBlock b;
  b.cblk = L % a.ncblk;
  L /= a.ncblk;
  b.prow = L;
  b.wblk = L % a.nwblk;
  L /= a.nwblk;
  b.strip = L % a.nstrips;
  b.n = L / a.nstrips;
  return b;
}

// ---- four channels of one pixel as two packed fp32 pairs (v_pk_fma_f32 / v_pk_add_f32: one issue
// slot per two lanes-elements — hipcc does not form them from scalar fmaf under -ffp-contract=off)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
struct Q4 { f32x2 lo, hi; };
__device__ __forceinline__ Q4 q4_zero() { Q4 q; q.lo = q.hi = (f32x2){0.f, 0.f}; return q; }
__device__ __forceinline__ void q4_fma(Q4& acc, const Q4& a, const Q4& b) {
  acc.lo = __builtin_elementwise_fma(a.lo, b.lo, acc.lo);
  acc.hi = __builtin_elementwise_fma(a.hi, b.hi, acc.hi);
}
template <typename T> struct SlIO;
template <> struct SlIO<bf16_t> {
  typedef uint2 raw_t;
  static constexpr int BYTES = 8;
  __device__ static __forceinline__ Q4 unpack(const raw_t& r) {
    Q4 q;
    q.lo = (f32x2){__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xFFFF0000u)};
    q.hi = (f32x2){__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xFFFF0000u)};
    return q;
  }
  // max(x, 0) on the packed storage words: bf16 is sign-magnitude, as int16 every negative
  // value (and -0) is < 0 (two v_pk_max_i16 instead of four v_max_f32)
  __device__ static __forceinline__ raw_t relu_raw(const raw_t& r) {
    const s16x2 z = {0, 0};
    raw_t o;
    o.x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, r.x), z));
    o.y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, r.y), z));
    return o;
  }
  static constexpr bool RAW_RELU = true;
  __device__ static __forceinline__ raw_t pack(const Q4& q) {
    return make_uint2(pack_bf16x2(q.lo.x, q.lo.y), pack_bf16x2(q.hi.x, q.hi.y));
  }
};
template <> struct SlIO<float> {
  typedef uint4 raw_t;
  static constexpr int BYTES = 16;
  __device__ static __forceinline__ Q4 unpack(const raw_t& r) {
    Q4 q;
    q.lo = (f32x2){__uint_as_float(r.x), __uint_as_float(r.y)};
    q.hi = (f32x2){__uint_as_float(r.z), __uint_as_float(r.w)};
    return q;
  }
  __device__ static __forceinline__ raw_t relu_raw(const raw_t& r) { return r; }
  static constexpr bool RAW_RELU = false;
  __device__ static __forceinline__ raw_t pack(const Q4& q) {
    return make_uint4(__float_as_uint(q.lo.x), __float_as_uint(q.lo.y), __float_as_uint(q.hi.x),
                      __float_as_uint(q.hi.y));
  }
};
// uniform base (SGPR pair) + per-lane 32-bit byte offset: the load / store needs no address VALU
template <typename R>
__device__ __forceinline__ R sl_ld(const unsigned char* __restrict__ base, unsigned off) {
  return *reinterpret_cast<const R*>(base + off);
}
template <typename R>
__device__ __forceinline__ void sl_st(unsigned char* __restrict__ base, unsigned off, const R& v) {
  *reinterpret_cast<R*>(base + off) = v;
}

// taps of this thread's four channels; the nine float4 of torch's [C][9] layout are one
// contiguous 144-byte run (36 c bytes from the start: 16-byte aligned for c % 4 == 0)
__device__ __forceinline__ void sl_load_taps(const DwSlideArgs& a, int c, Q4 (&wt)[9]) {
  float f[36];
  if (a.w_layout & 1) {
    const float4* p = reinterpret_cast<const float4*>(a.w + (long)c * 9);
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const float4 v = p[j];
      f[4 * j] = v.x; f[4 * j + 1] = v.y; f[4 * j + 2] = v.z; f[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int tap = (a.w_layout & 2) ? 8 - k : k;  // (runtime select between two registers)
      wt[k].lo = (f32x2){(a.w_layout & 2) ? f[8 - k] : f[k], (a.w_layout & 2) ? f[9 + 8 - k] : f[9 + k]};
      wt[k].hi = (f32x2){(a.w_layout & 2) ? f[18 + 8 - k] : f[18 + k],
                         (a.w_layout & 2) ? f[27 + 8 - k] : f[27 + k]};
      (void)tap;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int tap = (a.w_layout & 2) ? 8 - k : k;
      const float4 v = *reinterpret_cast<const float4*>(a.w + (long)tap * a.C + c);
      wt[k].lo = (f32x2){v.x, v.y};
      wt[k].hi = (f32x2){v.z, v.w};
    }
  }
}

// producer's BatchNorm (+ReLU / ReLU6) on one vector; MODE: compile-time prologue bits.
// RELU_DONE: the ReLU was already taken on the packed storage words.
template <int MODE, bool RELU_DONE>
__device__ __forceinline__ void sl_act(Q4& q, const Q4& sc, const Q4& sh) {
  const f32x2 z = {0.f, 0.f}, six = {6.f, 6.f};
  if (MODE & PRO_AFFINE) {
    q.lo = __builtin_elementwise_fma(q.lo, sc.lo, sh.lo);
    q.hi = __builtin_elementwise_fma(q.hi, sc.hi, sh.hi);
  }
  if ((MODE & PRO_RELU) && !RELU_DONE) {
    q.lo = __builtin_elementwise_max(q.lo, z);
    q.hi = __builtin_elementwise_max(q.hi, z);
  }
  if (MODE & PRO_CLAMP6) {
    q.lo = __builtin_elementwise_min(q.lo, six);
    q.hi = __builtin_elementwise_min(q.hi, six);
  }
}

// Geometry shared by both kernels.  Threads past the last column / channel quad MIRROR the last
// valid one (same addresses, same values: their stores rewrite identical bytes) instead of being
// masked: every step then issues the same memory instructions in every wave, no s_cbranch_execz
// sits between a load and its use, and hipcc counts s_waitcnt vmcnt(n) exactly (with per-lane
// branches around the loads / the store it waited vmcnt(0) before every use).  Zero padding costs
// nothing per step either: a column outside the image is a ZERO WEIGHT column of this thread
// (its taps with that kw never see a valid pixel), rows outside the image exist only in the first
// and the last step of an image (select in the strip's prologue / tail).  Mirrored threads are
// taken out of the statistics / tap sums once, at the end.
struct SlThread {
  int c, w, r0, r1, rlast;
  bool live, lm, rm;
  unsigned off_m, off_0, off_p;  // byte offsets of the three columns inside an image row
  // r06b: only the wave's two outer columns LOAD a neighbour (left of its first column, right of
  // its last one, one half-active instruction); the others take the neighbour's centre vector
  // from the lane 16 below / above (lane = column * 16 + channel quad, four columns per wave) by
  // ds_bpermute.  Three full loads per row were ~3 us of the 15.7 / 20.9 us launches (TA-bound:
  // profiles/r06_dw_slide.md, centre-only ablation 12.5 / 18.2 us).
  bool edge, first, last;
  unsigned off_e;
  int bp_up, bp_dn;  // ds_bpermute byte addresses of lane - 16 / lane + 16
};
template <int ESIZE>
__device__ __forceinline__ SlThread sl_thread(const DwSlideArgs& a, const SlBlock& b, long ld) {
  const int tid = threadIdx.x, cq = tid & (SL_CQ - 1), wl = tid >> 4;
  SlThread t;
  const int cfull = (b.cblk * SL_CQ + cq) * 4, wfull = b.wblk * SL_WL + wl;
  t.c = min(cfull, a.C - 4);
  t.w = min(wfull, a.W - 1);
  t.live = wfull < a.W && cfull < a.C;
  t.lm = t.w >= 1;
  t.rm = t.w + 1 < a.W;
  t.r0 = b.strip * a.rs;
  t.r1 = min(a.H, t.r0 + a.rs);
  t.rlast = min(t.r1, a.H - 1);  // last row the strip reads (bottom halo)
  t.off_m = (unsigned)((max(t.w - 1, 0) * ld + t.c) * ESIZE);
  t.off_0 = (unsigned)((t.w * ld + t.c) * ESIZE);
  t.off_p = (unsigned)((min(t.w + 1, a.W - 1) * ld + t.c) * ESIZE);
  const int lane = tid & 63, col = lane >> 4;
  t.first = col == 0;
  t.last = col == 3;
  t.edge = t.first || t.last;
  t.off_e = t.first ? t.off_m : (t.last ? t.off_p : 0xFFFFFFFFu);  // inner columns: out of range
  t.bp_up = ((lane - 16) & 63) * 4;
  t.bp_dn = ((lane + 16) & 63) * 4;
  return t;
}

// the outer columns' extra load as a BUFFER load over the image row: the inner columns pass an
// offset past the end of the buffer — they read nothing and get zeros, without a branch around the
// instruction (a per-lane `if` put an s_cbranch_execz between loads and uses: hipcc then stops
// counting vmcnt and the register allocation of the ring doubles)
typedef unsigned int sl_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int sl_u32x4 __attribute__((ext_vector_type(4)));
template <typename R> __device__ __forceinline__ R sl_ld_edge(const unsigned char* row, unsigned nbytes, unsigned off);
template <> __device__ __forceinline__ uint2 sl_ld_edge<uint2>(const unsigned char* row, unsigned nbytes, unsigned off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, (int)nbytes, 0x00020000);
  const sl_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
  return make_uint2(v.x, v.y);
}
template <> __device__ __forceinline__ uint4 sl_ld_edge<uint4>(const unsigned char* row, unsigned nbytes, unsigned off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, (int)nbytes, 0x00020000);
  const sl_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
// ... and the store as a buffer store: threads past the last column / channel quad pass an offset
// out of range and write nothing (with the lane exchange a mirrored thread no longer computes its
// twin's value — its neighbours are other mirrors — so it must not store at all)
template <typename R> __device__ __forceinline__ void sl_st_row(unsigned char* row, unsigned nbytes, unsigned off, const R& v);
template <> __device__ __forceinline__ void sl_st_row<uint2>(unsigned char* row, unsigned nbytes, unsigned off, const uint2& v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, (int)nbytes, 0x00020000);
  const sl_u32x2 d = {v.x, v.y};
  __builtin_amdgcn_raw_buffer_store_b64(d, r, (int)off, 0, 0);
}
template <> __device__ __forceinline__ void sl_st_row<uint4>(unsigned char* row, unsigned nbytes, unsigned off, const uint4& v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, (int)nbytes, 0x00020000);
  const sl_u32x4 d = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)off, 0, 0);
}
// the three vectors of a row from (own centre, the outer columns' extra load)
template <typename R> __device__ __forceinline__ R sl_bperm(int addr, const R& v);
template <> __device__ __forceinline__ uint2 sl_bperm<uint2>(int addr, const uint2& v) {
  return make_uint2((unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.x),
                    (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.y));
}
template <> __device__ __forceinline__ uint4 sl_bperm<uint4>(int addr, const uint4& v) {
  return make_uint4((unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.x),
                    (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.y),
                    (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.z),
                    (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v.w));
}
__device__ __forceinline__ uint2 sl_sel(bool c, const uint2& a, const uint2& b) {
  return make_uint2(c ? a.x : b.x, c ? a.y : b.y);
}
__device__ __forceinline__ uint4 sl_sel(bool c, const uint4& a, const uint4& b) {
  return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
template <typename R>
__device__ __forceinline__ void sl_neighbours(const SlThread& t, const R (&raw)[3], R& left, R& right) {
  const R up = sl_bperm<R>(t.bp_up, raw[0]), dn = sl_bperm<R>(t.bp_dn, raw[0]);
  left = sl_sel(t.first, raw[1], up);
  right = sl_sel(t.last, raw[1], dn);
}

// ------------------------------------------------------------------------------------ forward
template <typename T, int MODE, bool XCHG>
__global__ __launch_bounds__(SL_THREADS, SL_OCC_F) void dwconv_slide_fwd_kernel(const DwSlideArgs a) {
  using IO = SlIO<T>;
  using raw_t = typename IO::raw_t;
  constexpr int ES = sizeof(T);
  constexpr bool RAWRELU = IO::RAW_RELU && MODE == PRO_RELU;
  const int tid = threadIdx.x, cq = tid & (SL_CQ - 1);
  const SlBlock b = sl_block(a);
  const SlThread t = sl_thread<ES>(a, b, a.ldx);
  const int r0 = t.r0, r1 = t.r1;

  Q4 wt[9], sc, sh;
  sc = sh = q4_zero();
  const unsigned char* __restrict__ Xb = reinterpret_cast<const unsigned char*>(a.x) +
                                         (long)b.n * a.H * a.W * a.ldx * ES;
  unsigned char* __restrict__ Yb = reinterpret_cast<unsigned char*>(a.y) +
                                   (long)b.n * a.H * a.W * a.ldy * ES;
  const long xpitch = (long)a.W * a.ldx * ES, ypitch = (long)a.W * a.ldy * ES;
  const unsigned rowbytes = (unsigned)xpitch;
  const unsigned yoff = t.live ? (unsigned)((t.w * a.ldy + t.c) * ES) : 0xFFFFFFFFu;

  // (rows outside [0, rlast] re-read a row of the strip: unconditional, the value is unused/masked)
  auto issue = [&](int r, raw_t (&raw)[3]) {
    const unsigned char* __restrict__ row = Xb + (long)min(max(r, 0), t.rlast) * xpitch;
    if (XCHG) {
      raw[0] = sl_ld<raw_t>(row, t.off_0);
      raw[1] = sl_ld_edge<raw_t>(row, rowbytes, t.off_e);
    } else {
      raw[0] = sl_ld<raw_t>(row, t.off_0);
      raw[1] = sl_ld<raw_t>(row, t.off_m);
      raw[2] = sl_ld<raw_t>(row, t.off_p);
    }
  };
  auto commit = [&](int r, const raw_t (&raw)[3], Q4 (&row)[3], auto check) {
    constexpr bool CHECK = decltype(check)::value;
    const bool rv = !CHECK || (r >= 0 && r < a.H);
    raw_t r3[3];
    r3[1] = raw[0];
    if (XCHG) {
      sl_neighbours<raw_t>(t, raw, r3[0], r3[2]);
    } else {
      r3[0] = raw[1];
      r3[2] = raw[2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      row[k] = IO::unpack(RAWRELU ? IO::relu_raw(r3[k]) : r3[k]);
      sl_act<MODE, RAWRELU>(row[k], sc, sh);
      if (CHECK && !rv) row[k] = q4_zero();  // zero padding AFTER the activation
    }
  };
  Q4 ssum = q4_zero(), ssq = q4_zero();
  auto compute = [&](int ro, const Q4 (&ra)[3], const Q4 (&rb)[3], const Q4 (&rc)[3]) {
    Q4 acc = q4_zero();
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) q4_fma(acc, ra[kw], wt[kw]);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) q4_fma(acc, rb[kw], wt[3 + kw]);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) q4_fma(acc, rc[kw], wt[6 + kw]);
    sl_st_row<raw_t>(Yb + (long)ro * ypitch, (unsigned)ypitch, yoff, IO::pack(acc));
    ssum.lo += acc.lo; ssum.hi += acc.hi;
    q4_fma(ssq, acc, acc);
  };

  // D input rows in flight per thread (ring of raw vectors).  Row index j = r0 - 1 + j; window
  // role j % 3, ring slot j % D.
  constexpr int D = SL_DEPTH_F, U = (D % 3 == 0) ? D : 3 * D;
  Q4 win[3][3];
  raw_t ring[D][3];
#pragma unroll
  for (int j = 0; j < D; ++j) issue(r0 - 1 + j, ring[j]);
  // (taps and prologue parameters are requested BEHIND the first rows: one memory round trip at
  // the head of every strip instead of two)
  sl_load_taps(a, t.c, wt);
  if (MODE & PRO_AFFINE) {
    const float4 s4 = *reinterpret_cast<const float4*>(a.sc + t.c);
    const float4 t4 = *reinterpret_cast<const float4*>(a.sh + t.c);
    sc.lo = (f32x2){s4.x, s4.y}; sc.hi = (f32x2){s4.z, s4.w};
    sh.lo = (f32x2){t4.x, t4.y}; sh.hi = (f32x2){t4.z, t4.w};
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {  // columns outside the image: zero taps
    if (!t.lm) wt[kh * 3] = q4_zero();
    if (!t.rm) wt[kh * 3 + 2] = q4_zero();
  }
  commit(r0 - 1, ring[0], win[0], std::true_type{});
  issue(r0 - 1 + D, ring[0]);
  commit(r0, ring[1 % D], win[1], std::false_type{});
  issue(r0 + D, ring[1 % D]);
  int ro = r0;
  for (; ro + U < r1; ro += U) {  // full groups above the strip's last row: no branch, no select
#pragma unroll
    for (int k = 0; k < U; ++k) {
      // the row below the output row arrives (requested D steps ago), its slot is re-issued,
      // then the nine taps
      commit(ro + k + 1, ring[(k + 2) % D], win[(k + 2) % 3], std::false_type{});
      issue(ro + k + 1 + D, ring[(k + 2) % D]);
      SL_SCHED_BARRIER();
      compute(ro + k, win[k % 3], win[(k + 1) % 3], win[(k + 2) % 3]);
      SL_SCHED_BARRIER();
    }
  }
#pragma unroll
  for (int k = 0; k < U; ++k) {  // the last <= U rows of the strip (the image's last row: padding)
    if (ro + k < r1) {
      commit(ro + k + 1, ring[(k + 2) % D], win[(k + 2) % 3], std::true_type{});
      issue(ro + k + 1 + D, ring[(k + 2) % D]);
      compute(ro + k, win[k % 3], win[(k + 1) % 3], win[(k + 2) % 3]);
    }
  }

  if (a.partial != nullptr) {
    // columns of a wave by lane exchange (lane = wl_lo * 16 + cq), waves through LDS
    __shared__ float red[4][SL_CQ][8];
    float v[8] = {ssum.lo.x, ssum.lo.y, ssum.hi.x, ssum.hi.y, ssq.lo.x, ssq.lo.y, ssq.hi.x, ssq.hi.y};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = t.live ? v[i] : 0.f;  // mirrored threads
      v[i] += __shfl_xor(v[i], 16, 64);
      v[i] += __shfl_xor(v[i], 32, 64);
    }
    if ((tid & 63) < SL_CQ) {
#pragma unroll
      for (int i = 0; i < 8; ++i) red[tid >> 6][cq][i] = v[i];
    }
    __syncthreads();
    if (tid < SL_CQ * 8) {
      const int q = tid >> 3, k = tid & 7;
      const float tot = red[0][q][k] + red[1][q][k] + red[2][q][k] + red[3][q][k];
      const int ch = (b.cblk * SL_CQ + q) * 4 + (k & 3);
      if (ch < a.C) a.partial[((long)b.prow * 2 + (k >> 2)) * a.C + ch] = tot;
    }
  }
}

// ------------------------------------------------------------------------------ fused backward
//   g'[q]   = relu_mask(x[q]) * sum_k dy[q - d_k] * w[k]   (+ res[q])
//   dW[k]  += act(x[q]) * dy[q - d_k]                       the same shifted dy values
//   (sum g', sum g' * x_raw)                                of the masked gradient before `res`
// The window holds dy (no activation); window position (a, b) pairs with tap 8 - (3a + b).
template <typename T, int MODE, bool RES, bool XCHG>
__global__ __launch_bounds__(SL_THREADS, 2) void dwconv_slide_bwd_kernel(const DwSlideArgs a) {
  using IO = SlIO<T>;
  using raw_t = typename IO::raw_t;
  constexpr int ES = sizeof(T);
  const int tid = threadIdx.x, cq = tid & (SL_CQ - 1);
  const SlBlock b = sl_block(a);
  const SlThread t = sl_thread<ES>(a, b, a.lddy);
  const int r0 = t.r0, r1 = t.r1;

  Q4 wt[9], sc, sh;
  sc = sh = q4_zero();
  const long img = (long)b.n * a.H * a.W;
  const unsigned char* __restrict__ Db = reinterpret_cast<const unsigned char*>(a.dy) + img * a.lddy * ES;
  const unsigned char* __restrict__ Xb = reinterpret_cast<const unsigned char*>(a.x) + img * a.ldx * ES;
  const unsigned char* __restrict__ Rb =
      RES ? reinterpret_cast<const unsigned char*>(a.res) + img * a.ldr * ES : nullptr;
  unsigned char* __restrict__ Gb = reinterpret_cast<unsigned char*>(a.y) + img * a.ldy * ES;
  const long dpitch = (long)a.W * a.lddy * ES, xpitch = (long)a.W * a.ldx * ES;
  const long rpitch = (long)a.W * a.ldr * ES, gpitch = (long)a.W * a.ldy * ES;
  const unsigned rowbytes = (unsigned)dpitch;
  const unsigned xoff = (unsigned)((t.w * a.ldx + t.c) * ES), roff = (unsigned)((t.w * a.ldr + t.c) * ES);
  const unsigned goff = t.live ? (unsigned)((t.w * a.ldy + t.c) * ES) : 0xFFFFFFFFu;

  auto issue = [&](int r, raw_t (&raw)[3]) {
    const unsigned char* __restrict__ row = Db + (long)min(max(r, 0), t.rlast) * dpitch;
    if (XCHG) {
      raw[0] = sl_ld<raw_t>(row, t.off_0);
      raw[1] = sl_ld_edge<raw_t>(row, rowbytes, t.off_e);
    } else {
      raw[0] = sl_ld<raw_t>(row, t.off_0);
      raw[1] = sl_ld<raw_t>(row, t.off_m);
      raw[2] = sl_ld<raw_t>(row, t.off_p);
    }
  };
  auto issue_x = [&](int r, raw_t& xr, raw_t& rr) {  // centre pixel of output row r
    const int rcl = min(r, r1 - 1);
    xr = sl_ld<raw_t>(Xb + (long)rcl * xpitch, xoff);
    if (RES) rr = sl_ld<raw_t>(Rb + (long)rcl * rpitch, roff);
  };
  auto commit = [&](int r, const raw_t (&raw)[3], Q4 (&row)[3], auto check) {
    constexpr bool CHECK = decltype(check)::value;
    const bool rv = !CHECK || (r >= 0 && r < a.H);
    raw_t r3[3];
    r3[1] = raw[0];
    if (XCHG) {
      sl_neighbours<raw_t>(t, raw, r3[0], r3[2]);
    } else {
      r3[0] = raw[1];
      r3[2] = raw[2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      row[k] = IO::unpack(r3[k]);
      if (CHECK && !rv) row[k] = q4_zero();
    }
  };
  Q4 accw[9], s1 = q4_zero(), s2 = q4_zero();
#pragma unroll
  for (int k = 0; k < 9; ++k) accw[k] = q4_zero();

  auto compute = [&](int ro, const raw_t& xraw, const raw_t& rraw, const Q4 (&ra)[3],
                     const Q4 (&rb)[3], const Q4 (&rcw)[3]) {
    const Q4 xr = IO::unpack(xraw);
    Q4 xa = xr;
    sl_act<MODE, false>(xa, sc, sh);
    Q4 g = q4_zero();
#pragma unroll
    for (int bq = 0; bq < 3; ++bq) { q4_fma(g, ra[bq], wt[8 - bq]); q4_fma(accw[8 - bq], ra[bq], xa); }
#pragma unroll
    for (int bq = 0; bq < 3; ++bq) { q4_fma(g, rb[bq], wt[5 - bq]); q4_fma(accw[5 - bq], rb[bq], xa); }
#pragma unroll
    for (int bq = 0; bq < 3; ++bq) { q4_fma(g, rcw[bq], wt[2 - bq]); q4_fma(accw[2 - bq], rcw[bq], xa); }
    if (MODE & PRO_RELU) {
      auto on = [&](float v) { return v > 0.f && (!(MODE & PRO_CLAMP6) || v < 6.f); };
      g.lo.x = on(xa.lo.x) ? g.lo.x : 0.f; g.lo.y = on(xa.lo.y) ? g.lo.y : 0.f;
      g.hi.x = on(xa.hi.x) ? g.hi.x : 0.f; g.hi.y = on(xa.hi.y) ? g.hi.y : 0.f;
    }
    if (RES) {
      const Q4 rr = IO::unpack(rraw);
      Q4 o;
      o.lo = g.lo + rr.lo;
      o.hi = g.hi + rr.hi;
      sl_st_row<raw_t>(Gb + (long)ro * gpitch, (unsigned)gpitch, goff, IO::pack(o));
    } else {
      sl_st_row<raw_t>(Gb + (long)ro * gpitch, (unsigned)gpitch, goff, IO::pack(g));
    }
    s1.lo += g.lo; s1.hi += g.hi;
    q4_fma(s2, g, xr);
  };

  constexpr int D = SL_DEPTH_B, U = (D % 3 == 0) ? D : 3 * D;
  Q4 win[3][3];
  raw_t ring[D][3], xring[D], rring[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    issue(r0 - 1 + j, ring[j]);
    rring[j] = xring[j] = sl_ld<raw_t>(Xb + (long)min(r0 + j, r1 - 1) * xpitch, xoff);
    if (RES) rring[j] = sl_ld<raw_t>(Rb + (long)min(r0 + j, r1 - 1) * rpitch, roff);
  }
  sl_load_taps(a, t.c, wt);
  if (MODE & PRO_AFFINE) {
    const float4 s4 = *reinterpret_cast<const float4*>(a.sc + t.c);
    const float4 t4 = *reinterpret_cast<const float4*>(a.sh + t.c);
    sc.lo = (f32x2){s4.x, s4.y}; sc.hi = (f32x2){s4.z, s4.w};
    sh.lo = (f32x2){t4.x, t4.y}; sh.hi = (f32x2){t4.z, t4.w};
  }
  // window column b pairs with kernel column 2 - b: a dy column outside the image is a zero tap
  // column for the data gradient (its weight-gradient sums are dropped at the end)
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    if (!t.lm) wt[kh * 3 + 2] = q4_zero();
    if (!t.rm) wt[kh * 3] = q4_zero();
  }
  commit(r0 - 1, ring[0], win[0], std::true_type{});
  issue(r0 - 1 + D, ring[0]);
  commit(r0, ring[1 % D], win[1], std::false_type{});
  issue(r0 + D, ring[1 % D]);
  int ro = r0;
  for (; ro + U < r1; ro += U) {
#pragma unroll
    for (int k = 0; k < U; ++k) {
      commit(ro + k + 1, ring[(k + 2) % D], win[(k + 2) % 3], std::false_type{});
      issue(ro + k + 1 + D, ring[(k + 2) % D]);
      const raw_t xraw = xring[k % D], rraw = rring[k % D];
      issue_x(ro + k + D, xring[k % D], rring[k % D]);
      SL_SCHED_BARRIER();
      compute(ro + k, xraw, rraw, win[k % 3], win[(k + 1) % 3], win[(k + 2) % 3]);
      SL_SCHED_BARRIER();
    }
  }
#pragma unroll
  for (int k = 0; k < U; ++k) {
    if (ro + k < r1) {
      commit(ro + k + 1, ring[(k + 2) % D], win[(k + 2) % 3], std::true_type{});
      issue(ro + k + 1 + D, ring[(k + 2) % D]);
      const raw_t xraw = xring[k % D], rraw = rring[k % D];
      issue_x(ro + k + D, xring[k % D], rring[k % D]);
      compute(ro + k, xraw, rraw, win[k % 3], win[(k + 1) % 3], win[(k + 2) % 3]);
    }
  }

  // ---- block reduction: 44 values per thread over the block's 16 columns.  Mirrored threads add
  // nothing; a dy column outside the image never met a valid pixel: its taps (kw = 2 - b) drop out.
  float v[44];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int kw = k % 3;
    const bool keep = t.live && (kw != 2 || t.lm) && (kw != 0 || t.rm);
    v[k * 4] = keep ? accw[k].lo.x : 0.f; v[k * 4 + 1] = keep ? accw[k].lo.y : 0.f;
    v[k * 4 + 2] = keep ? accw[k].hi.x : 0.f; v[k * 4 + 3] = keep ? accw[k].hi.y : 0.f;
  }
  v[36] = s1.lo.x; v[37] = s1.lo.y; v[38] = s1.hi.x; v[39] = s1.hi.y;
  v[40] = s2.lo.x; v[41] = s2.lo.y; v[42] = s2.hi.x; v[43] = s2.hi.y;
#pragma unroll
  for (int i = 36; i < 44; ++i) v[i] = t.live ? v[i] : 0.f;
#pragma unroll
  for (int i = 0; i < 44; ++i) {
    v[i] += __shfl_xor(v[i], 16, 64);
    v[i] += __shfl_xor(v[i], 32, 64);
  }
  __shared__ float red[4][SL_CQ][44];
  if ((tid & 63) < SL_CQ) {
#pragma unroll
    for (int i = 0; i < 44; ++i) red[tid >> 6][cq][i] = v[i];
  }
  __syncthreads();
  for (int e = tid; e < SL_CQ * 44; e += SL_THREADS) {
    const int q = e / 44, k = e - q * 44;
    const float tot = red[0][q][k] + red[1][q][k] + red[2][q][k] + red[3][q][k];
    const int ch = (b.cblk * SL_CQ + q) * 4 + (k & 3), r = k >> 2;
    if (ch < a.C) {
      if (r < 9) a.partial[((long)b.prow * 9 + r) * a.C + ch] = tot;
      else if (a.partial_bn != nullptr) a.partial_bn[((long)b.prow * 2 + (r - 9)) * a.C + ch] = tot;
    }
  }
}

// ---------------------------------------------------------------------------------------- host
bool dw_slide_supported(int stride, int dil, int C) { return stride == 1 && dil == 1 && C % 4 == 0; }

// Strips of ~43 rows: a strip pays a fixed price (tap / parameter loads, the first D rows' round
// trip, two halo rows, the block reduction) that 13-row strips did not amortise on the 65 x 129
// middle-flow map, while one strip per image leaves too few waves (profiles/r06_dw_slide.md:
// [2,65,129,728] forward / backward 18.2 / 24.8 us with 1 strip, 15.7 / 20.9 with 2, 16.9 / 25.7
// with 5; 129 rows: 3 strips, 257: 6, 513: 7-8 best) — and few enough partial rows for the
// one-launch finalize kernels (R <= 1024).
static void slide_geom(DwSlideArgs& a, int N, int H, int W, int C) {
  a.ncblk = (C / 4 + SL_CQ - 1) / SL_CQ;
  a.nwblk = (W + SL_WL - 1) / SL_WL;
  long ns = (H + 21) / 43;
  // ... but at least ~400 blocks where the map allows strips of >= 8 rows (batch-1 inference:
  // one strip of a 64 x 128 map is 120 blocks — DeepLabv3+/MobileNetV2 940 -> 882 img/s)
  const long per_strip = (long)N * a.nwblk * a.ncblk;
  const long want = (400 + per_strip - 1) / per_strip;
  if (ns < want) ns = want;
  if (ns > H / 8) ns = H / 8;
  const long cap = 1024 / ((long)N * a.nwblk);
  if (ns > cap) ns = cap;
  if (ns < 1) ns = 1;
  a.rs = (int)((H + ns - 1) / ns);
  a.nstrips = (H + a.rs - 1) / a.rs;
}

int dw_slide_rows(int C, int N, int H, int W) {
  DwSlideArgs a;
  slide_geom(a, N, H, W, C);
  return N * a.nstrips * a.nwblk;
}

// Neighbour columns by lane exchange (one full + one half-active load per row) or by three loads:
// the exchange adds ~16 instructions per row step (4 ds_bpermute, 4 selects, the buffer
// descriptors) and wins where the loads are the limit — every map of >= 30 MB: [2,513,1025,128]
// forward 154 -> 122 us, backward 270 -> 175; [2,257,513,256] 72 -> 55 / 153 -> 95 — and loses
// on the 24 MB middle-flow maps (15.7 -> 16.5 / 20.9 -> 21.8 us: issue-bound at 1.7 waves per SIMD).
static bool slide_exchange(const DwSlideArgs& a, int esize) {
  return (long)a.N * a.H * a.W * a.C * esize >= (30L << 20);
}
template <typename T>
static void slide_launch_fwd(const DwSlideArgs& a, dim3 grid, hipStream_t st) {
  const bool xc = slide_exchange(a, (int)sizeof(T));
  switch (a.pro_mode) {
#define SL_CASE(M)                                                                                  \
  case M:                                                                                           \
    if (xc) hipLaunchKernelGGL((dwconv_slide_fwd_kernel<T, M, true>), grid, dim3(SL_THREADS), 0, st, a);  \
    else hipLaunchKernelGGL((dwconv_slide_fwd_kernel<T, M, false>), grid, dim3(SL_THREADS), 0, st, a);    \
    break;
    SL_CASE(0) SL_CASE(1) SL_CASE(2) SL_CASE(3) SL_CASE(5) SL_CASE(7)
#undef SL_CASE
    default: break;
  }
}
template <typename T, bool RES>
static void slide_launch_bwd(const DwSlideArgs& a, dim3 grid, hipStream_t st) {
  const bool xc = slide_exchange(a, (int)sizeof(T));
  switch (a.pro_mode) {
#define SL_CASE(M)                                                                                       \
  case M:                                                                                                \
    if (xc) hipLaunchKernelGGL((dwconv_slide_bwd_kernel<T, M, RES, true>), grid, dim3(SL_THREADS), 0, st, a);  \
    else hipLaunchKernelGGL((dwconv_slide_bwd_kernel<T, M, RES, false>), grid, dim3(SL_THREADS), 0, st, a);    \
    break;
    SL_CASE(0) SL_CASE(1) SL_CASE(2) SL_CASE(3) SL_CASE(5) SL_CASE(7)
#undef SL_CASE
    default: break;
  }
}
static bool slide_mode_ok(int m) { return m == 0 || m == 1 || m == 2 || m == 3 || m == 5 || m == 7; }

int launch_dw_slide_fwd(int dtype, const void* x, long ldx, int N, int H, int W, int C,
                        const float* w, int w_layout, int pro_mode, const float* sc,
                        const float* sh, void* y, long ldy, float* stat_partial, int rows,
                        hipStream_t st) {
  DwSlideArgs a;
  a.x = x; a.dy = nullptr; a.res = nullptr; a.y = y; a.w = w; a.w_layout = w_layout;
  a.sc = sc; a.sh = sh; a.partial = stat_partial; a.partial_bn = nullptr;
  a.ldx = ldx; a.lddy = 0; a.ldy = ldy; a.ldr = 0;
  a.N = N; a.H = H; a.W = W; a.C = C; a.pro_mode = pro_mode;
  slide_geom(a, N, H, W, C);
  SEG_REQUIRE(slide_mode_ok(pro_mode), "dwconv (slide): unsupported prologue mode %d", pro_mode);
  SEG_REQUIRE(stat_partial == nullptr || rows == N * a.nstrips * a.nwblk,
              "dwconv (slide): %d partial rows, the launch writes %d (seg_dwconv_grid_y)", rows,
              N * a.nstrips * a.nwblk);
  const dim3 grid((unsigned)((long)a.ncblk * a.nwblk * a.nstrips * N));
  if (dtype == DT_BF16) slide_launch_fwd<bf16_t>(a, grid, st);
  else slide_launch_fwd<float>(a, grid, st);
  return check_launch("dwconv3x3 (slide)");
}

int launch_dw_slide_bwd(int dtype, const void* dy, long lddy, const void* x, long ldx, int N, int H,
                        int W, int C, const float* w, int w_layout, int pro_mode, const float* sc,
                        const float* sh, void* g, long ldg, float* partial_w, float* partial_bn,
                        int rows, hipStream_t st, const void* res, long ldr) {
  DwSlideArgs a;
  a.x = x; a.dy = dy; a.res = res; a.y = g; a.w = w; a.w_layout = w_layout & 1;
  a.sc = sc; a.sh = sh; a.partial = partial_w; a.partial_bn = partial_bn;
  a.ldx = ldx; a.lddy = lddy; a.ldy = ldg; a.ldr = ldr;
  a.N = N; a.H = H; a.W = W; a.C = C; a.pro_mode = pro_mode;
  slide_geom(a, N, H, W, C);
  SEG_REQUIRE(slide_mode_ok(pro_mode), "dwconv bwd (slide): unsupported prologue mode %d", pro_mode);
  SEG_REQUIRE(rows == N * a.nstrips * a.nwblk,
              "dwconv bwd (slide): %d partial rows, the launch writes %d (seg_dwconv_grid_y)", rows,
              N * a.nstrips * a.nwblk);
  const dim3 grid((unsigned)((long)a.ncblk * a.nwblk * a.nstrips * N));
  if (dtype == DT_BF16) {
    if (res) slide_launch_bwd<bf16_t, true>(a, grid, st);
    else slide_launch_bwd<bf16_t, false>(a, grid, st);
  } else {
    if (res) slide_launch_bwd<float, true>(a, grid, st);
    else slide_launch_bwd<float, false>(a, grid, st);
  }
  return check_launch("dwconv3x3_bwd_fused (slide)");
}

}  // namespace seg
