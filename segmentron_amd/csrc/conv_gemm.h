// MFMA tile core shared by the implicit-GEMM convolution kernels (forward/dgrad and wgrad).
//
// Block tile 128(M) x 128(N), K staged in slabs of 128 BYTES per row (64 bf16 / 32 f32), 4 waves
// arranged 2x2, each wave owns a 64x64 sub-tile = 2x2 MFMA 32x32 tiles (64 fp32 accumulators).
// Both operands live in LDS as [row][k] with k contiguous and a 16-byte pad per row (row stride
// 144 B: 9 sixteen-byte slots, coprime with the 16 slots of a bank row -> conflict-free
// ds_read_b128 fragment reads and ds_write_b128 staging writes).
//
// One "vector step" consumes one 16-byte vector of A and of B per lane:
//   bf16 : v_mfma_f32_32x32x16_bf16 (K=16: lanes 0-31 hold k 0..7, lanes 32-63 k 8..15)
//   f32  : 4 x v_mfma_f32_32x32x2_f32 (exact fp32; lanes 0-31 take element i of vector 2s,
//          lanes 32-63 element i of vector 2s+1 — any k->lane assignment is valid as long as
//          A and B use the same one, which they do)
// C/D layout (both): col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
#pragma once
#include "common.h"

namespace seg {

constexpr int BM = 128, BN = 128;
constexpr int ROW_BYTES = 128;           // K-slab bytes per row
constexpr int ROW_STRIDE = ROW_BYTES + 16;
constexpr int TILE_BYTES = 128 * ROW_STRIDE;  // one operand tile in LDS (18432 B)
constexpr int GEMM_THREADS = 256;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ void step(const uint4& a, const uint4& b, f32x16& c) {
    union { uint4 u; bf16x8 v; } ua, ub;
    ua.u = a; ub.u = b;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  __device__ static __forceinline__ void step(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// Multiply the staged slab: acc[i][j] += A(wave rows, i) * B(wave cols, j)^T over the slab.
template <typename T>
__device__ __forceinline__ void mma_slab(const unsigned char* __restrict__ sA,
                                         const unsigned char* __restrict__ sB, int wm, int wn,
                                         int lane, f32x16 (&acc)[2][2]) {
  const int r = lane & 31, h = lane >> 5;
  const unsigned char* pa = sA + (wm * 64 + r) * ROW_STRIDE + h * 16;
  const unsigned char* pb = sB + (wn * 64 + r) * ROW_STRIDE + h * 16;
#pragma unroll
  for (int s = 0; s < ROW_BYTES / 32; ++s) {
    uint4 a0 = *reinterpret_cast<const uint4*>(pa + s * 32);
    uint4 a1 = *reinterpret_cast<const uint4*>(pa + 32 * ROW_STRIDE + s * 32);
    uint4 b0 = *reinterpret_cast<const uint4*>(pb + s * 32);
    uint4 b1 = *reinterpret_cast<const uint4*>(pb + 32 * ROW_STRIDE + s * 32);
    Mma<T>::step(a0, b0, acc[0][0]);
    Mma<T>::step(a0, b1, acc[0][1]);
    Mma<T>::step(a1, b0, acc[1][0]);
    Mma<T>::step(a1, b1, acc[1][1]);
  }
}

}  // namespace seg
