// Row softmax (forward / backward) of the attention matrices of DANet's position- and
// channel-attention modules (segmentron/modules/module.py:100-162: `nn.Softmax(dim=-1)` between
// two torch.bmm).  The two bmm run on the MFMA convolution kernels (a 1x1 convolution IS the
// NT GEMM, its weight gradient the TN GEMM); these kernels are the step in between:
//   A[r, j] = softmax_j(sign * E[r, j]),  j < L      (columns L..Lp-1 are written as zeros: the
//                                                      GEMMs need a row pitch of whole vectors)
//   dE[r, j] = sign * A[r, j] * (dA[r, j] - sum_j A[r, j] dA[r, j])
// sign = -1 serves CAM's `softmax(max(energy) - energy)` (module.py:152-154): softmax is shift
// invariant, so it equals softmax(-energy) and the row maximum needs no gradient.
// One block per row, three passes over a row that stays in L2; fp32 arithmetic; fixed-order
// block reductions (deterministic).  Element types are runtime codes (DT_F32 / DT_BF16): the
// energies come out of the GEMMs in the compute dtype, CAM's C x C energies as fp32 partial sums.
#include "common.h"

namespace seg {

constexpr int SM_THREADS = 256;

__device__ __forceinline__ float sm_ld(const void* p, int dt, long i) {
  return dt == DT_BF16 ? bf16_to_f32(reinterpret_cast<const bf16_t*>(p)[i])
                       : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void sm_st(void* p, int dt, long i, float v) {
  if (dt == DT_BF16) reinterpret_cast<bf16_t*>(p)[i] = f32_to_bf16(v);
  else reinterpret_cast<float*>(p)[i] = v;
}

template <bool MAX>
__device__ __forceinline__ float sm_block_reduce(float v, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float w = __shfl_xor(v, o, 64);
    v = MAX ? fmaxf(v, w) : v + w;
  }
  __syncthreads();  // (red may still be read from the previous reduction)
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int k = 1; k < SM_THREADS / 64; ++k) r = MAX ? fmaxf(r, red[k]) : r + red[k];
  return r;
}

__global__ __launch_bounds__(SM_THREADS) void row_softmax_kernel(
    const void* __restrict__ e, int dt_in, long lde, void* __restrict__ a, int dt_out, long lda,
    int L, int Lp, float sign) {
  __shared__ float red[SM_THREADS / 64];
  const long r = blockIdx.x;
  const long e0 = r * lde, a0 = r * lda;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < L; j += SM_THREADS) m = fmaxf(m, sign * sm_ld(e, dt_in, e0 + j));
  m = sm_block_reduce<true>(m, red);
  float s = 0.f;
  for (int j = threadIdx.x; j < L; j += SM_THREADS) s += expf(sign * sm_ld(e, dt_in, e0 + j) - m);
  s = sm_block_reduce<false>(s, red);
  const float inv = 1.f / s;
  for (int j = threadIdx.x; j < Lp; j += SM_THREADS)
    sm_st(a, dt_out, a0 + j, j < L ? expf(sign * sm_ld(e, dt_in, e0 + j) - m) * inv : 0.f);
}

__global__ __launch_bounds__(SM_THREADS) void row_softmax_bwd_kernel(
    const void* __restrict__ a, int dt_a, long lda, const void* __restrict__ g, int dt_g, long ldg,
    void* __restrict__ de, int dt_out, long ldde, int L, int Lp, float sign) {
  __shared__ float red[SM_THREADS / 64];
  const long r = blockIdx.x;
  float s = 0.f;
  for (int j = threadIdx.x; j < L; j += SM_THREADS)
    s = fmaf(sm_ld(a, dt_a, r * lda + j), sm_ld(g, dt_g, r * ldg + j), s);
  s = sm_block_reduce<false>(s, red);
  for (int j = threadIdx.x; j < Lp; j += SM_THREADS)
    sm_st(de, dt_out, r * ldde + j,
          j < L ? sign * sm_ld(a, dt_a, r * lda + j) * (sm_ld(g, dt_g, r * ldg + j) - s) : 0.f);
}

}  // namespace seg

static bool sm_dt_ok(int dt) { return dt == seg::DT_F32 || dt == seg::DT_BF16; }

extern "C" int seg_row_softmax(const void* e, int dt_in, long lde, void* a, int dt_out, long lda,
                               long R, int L, int Lp, float sign, void* stream) {
  using namespace seg;
  SEG_REQUIRE(sm_dt_ok(dt_in) && sm_dt_ok(dt_out), "row_softmax: bad dtype");
  SEG_REQUIRE(R >= 1 && R < (1L << 31) && L >= 1 && Lp >= L && lde >= L && lda >= Lp,
              "row_softmax: bad R / L / pitch");
  hipLaunchKernelGGL(row_softmax_kernel, dim3((unsigned)R), dim3(SM_THREADS), 0,
                     (hipStream_t)stream, e, dt_in, lde, a, dt_out, lda, L, Lp, sign);
  return check_launch("row_softmax");
}

extern "C" int seg_row_softmax_bwd(const void* a, int dt_a, long lda, const void* g, int dt_g,
                                   long ldg, void* de, int dt_out, long ldde, long R, int L,
                                   int Lp, float sign, void* stream) {
  using namespace seg;
  SEG_REQUIRE(sm_dt_ok(dt_a) && sm_dt_ok(dt_g) && sm_dt_ok(dt_out), "row_softmax_bwd: bad dtype");
  SEG_REQUIRE(R >= 1 && R < (1L << 31) && L >= 1 && Lp >= L && lda >= L && ldg >= L && ldde >= Lp,
              "row_softmax_bwd: bad R / L / pitch");
  hipLaunchKernelGGL(row_softmax_bwd_kernel, dim3((unsigned)R), dim3(SM_THREADS), 0,
                     (hipStream_t)stream, a, dt_a, lda, g, dt_g, ldg, de, dt_out, ldde, L, Lp, sign);
  return check_launch("row_softmax_bwd");
}
