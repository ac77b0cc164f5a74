// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Activations are NHWC ("channels_last" memory), element type T in {float, bf16}; per-channel
// BatchNorm parameters / statistics are always fp32; accumulation is always fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/segmentron_hip.h"  // the public C-ABI: definitions are checked against it

namespace seg {

typedef uint16_t bf16_t;  // raw bfloat16 storage

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { DT_F32 = 0, DT_BF16 = 1 };
// prologue applied to a conv / elementwise input:  v = x*scale[c] + shift[c]  (if AFFINE), then relu
enum { PRO_NONE = 0, PRO_RELU = 1, PRO_AFFINE = 2, PRO_AFFINE_RELU = 3, PRO_CLAMP6 = 4 };
// PRO_CLAMP6 (with PRO_RELU): ReLU6 = min(max(v,0),6)  (segmentron/modules/basic.py:71)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// float -> bf16, round-to-nearest-even.  Native __bf16 conversions lower to the gfx950
// v_cvt_pk_bf16_f32 instruction (two elements per VALU op) — the hand-rolled integer rounding
// sequence cost ~5 VALU ops per element in every store path.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t f = {lo, hi};
  union { bf16x2_t v; uint32_t u; } c;
  c.v = __builtin_convertvector(f, bf16x2_t);
  return c.u;
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  union { __bf16 b; bf16_t u; } c;
  c.b = static_cast<__bf16>(f);
  return c.u;
}

// 16-byte vector view of T: VEC elements
template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  typedef uint4 raw_t;
  __device__ static __forceinline__ raw_t load_raw(const float* p) {
    return *reinterpret_cast<const uint4*>(p);
  }
  __device__ static __forceinline__ void unpack_raw(const raw_t& v, float* f) { unpack(v, f); }
  __device__ static __forceinline__ raw_t pack_raw(const float* f) { return pack(f); }
  __device__ static __forceinline__ raw_t zero_raw() { return make_uint4(0u, 0u, 0u, 0u); }
  __device__ static __forceinline__ void store(float* p, const float* f) {
    *reinterpret_cast<uint4*>(p) = pack(f);
  }
  __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
    f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                      __float_as_uint(f[3]));
  }
  __device__ static __forceinline__ float load1(const float* p) { return *p; }
  __device__ static __forceinline__ void store1(float* p, float v) { *p = v; }
};
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  typedef uint4 raw_t;
  __device__ static __forceinline__ raw_t load_raw(const bf16_t* p) {
    return *reinterpret_cast<const uint4*>(p);
  }
  __device__ static __forceinline__ void unpack_raw(const raw_t& v, float* f) { unpack(v, f); }
  __device__ static __forceinline__ raw_t pack_raw(const float* f) { return pack(f); }
  __device__ static __forceinline__ raw_t zero_raw() { return make_uint4(0u, 0u, 0u, 0u); }
  __device__ static __forceinline__ void store(bf16_t* p, const float* f) {
    *reinterpret_cast<uint4*>(p) = pack(f);
  }
  __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xFFFF0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xFFFF0000u);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                      pack_bf16x2(f[6], f[7]));
  }
  __device__ static __forceinline__ float load1(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void store1(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 4-element vector view (16 B of fp32, 8 B of bf16) for register-heavy kernels
template <typename T> struct HVec;
template <> struct HVec<float> {
  static constexpr int N = 4;
  typedef uint4 raw_t;
  __device__ static __forceinline__ raw_t load_raw(const float* p) {
    return *reinterpret_cast<const uint4*>(p);
  }
  __device__ static __forceinline__ void unpack_raw(const raw_t& v, float* f) {
    Vec<float>::unpack(v, f);
  }
  __device__ static __forceinline__ raw_t pack_raw(const float* f) { return Vec<float>::pack(f); }
  __device__ static __forceinline__ raw_t zero_raw() { return make_uint4(0u, 0u, 0u, 0u); }
  __device__ static __forceinline__ void load(const float* p, float* f) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    Vec<float>::unpack(v, f);
  }
  __device__ static __forceinline__ void store(float* p, const float* f) {
    *reinterpret_cast<uint4*>(p) = Vec<float>::pack(f);
  }
};
template <> struct HVec<bf16_t> {
  static constexpr int N = 4;
  typedef uint2 raw_t;
  __device__ static __forceinline__ raw_t load_raw(const bf16_t* p) {
    return *reinterpret_cast<const uint2*>(p);
  }
  __device__ static __forceinline__ void unpack_raw(const raw_t& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
  }
  __device__ static __forceinline__ raw_t pack_raw(const float* f) {
    return make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
  }
  __device__ static __forceinline__ raw_t zero_raw() { return make_uint2(0u, 0u); }
  __device__ static __forceinline__ void load(const bf16_t* p, float* f) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float* f) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
  }
};

__device__ __forceinline__ uint4 ldg16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void stg16(void* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }

// N (multiple of 4) consecutive fp32 per-channel parameters starting at channel c (c % 4 == 0,
// rows of parameter tables are 16-byte aligned): 16-byte loads.  Element-wise `p[c + i]` under a
// runtime mode test compiles to N branchy dword loads with a full s_waitcnt between them — eight
// serialized L2 round trips (~8 us) at the head of every element-wise kernel.
template <int N>
__device__ __forceinline__ void load_params(const float* __restrict__ p, int c, float* out) {
  static_assert(N % 4 == 0, "parameter vectors are loaded 16 bytes at a time");
#pragma unroll
  for (int i = 0; i < N; i += 4) {
    const float4 v = *reinterpret_cast<const float4*>(p + c + i);
    out[i] = v.x; out[i + 1] = v.y; out[i + 2] = v.z; out[i + 3] = v.w;
  }
}

// Apply the fused-BatchNorm prologue to VEC consecutive channels starting at c.
template <int N>
__device__ __forceinline__ void apply_prologue(float* f, int mode, const float* __restrict__ scale,
                                               const float* __restrict__ shift, int c) {
  if (mode & PRO_AFFINE) {
    float s[N], t[N];
    load_params<N>(scale, c, s);
    load_params<N>(shift, c, t);
#pragma unroll
    for (int i = 0; i < N; ++i) f[i] = fmaf(f[i], s[i], t[i]);
  }
  if (mode & PRO_RELU) {
#pragma unroll
    for (int i = 0; i < N; ++i) f[i] = fmaxf(f[i], 0.f);
  }
  if (mode & PRO_CLAMP6) {
#pragma unroll
    for (int i = 0; i < N; ++i) f[i] = fminf(f[i], 6.f);
  }
}

// Same, with the channel parameters already in registers (loaded once per K slab).
template <int N>
__device__ __forceinline__ void apply_prologue_regs(float* f, int mode, const float (&s)[N],
                                                    const float (&t)[N]) {
  if (mode & PRO_AFFINE) {
#pragma unroll
    for (int i = 0; i < N; ++i) f[i] = fmaf(f[i], s[i], t[i]);
  }
  if (mode & PRO_RELU) {
#pragma unroll
    for (int i = 0; i < N; ++i) f[i] = fmaxf(f[i], 0.f);
  }
  if (mode & PRO_CLAMP6) {
#pragma unroll
    for (int i = 0; i < N; ++i) f[i] = fminf(f[i], 6.f);
  }
}

// v if keep else 0, component-wise (a ?: on the uint4 STRUCT becomes a select between two stack
// addresses and sends both operands through scratch memory)
__device__ __forceinline__ uint4 mask_u4(const uint4& v, bool keep) {
  const unsigned m = keep ? 0xFFFFFFFFu : 0u;
  return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}

// MI355X has 8 XCDs with private L2s and the dispatcher places block b on XCD b % 8
// (speed-only assumption).  Remap so each XCD works on a contiguous range of logical tiles
// (bijective for any block count).
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = b & 7, idx = b >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace seg

// ---- host side error plumbing (C-ABI: int status + thread-local message) -------------------
namespace seg {
void set_error(const char* fmt, ...);
int check_launch(const char* what);
}  // namespace seg

#define SEG_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      seg::set_error(__VA_ARGS__);        \
      return 1;                           \
    }                                     \
  } while (0)
