// 1x1 / stride-1 implicit GEMM, second generation: block tile 256 pixels x 128 channels, four
// waves as 2 (pixel halves) x 2 (channel halves), each wave 128 px x 64 ch = 4 x 2 MFMA 32x32
// tiles (128 fp32 accumulators).  Same call sites as conv_gemm_fwd.hip (pointwise / shortcut /
// ASPP / classifier convs and every 1x1 data gradient: 96 % of the DeepLabv3+ FLOPs).
//
// Why: with 64x64 per wave (conv_gemm_fwd_kernel) one k-step reads 2+2 operand fragments from LDS
// for 4 MFMAs = 128 B/clk/CU at MFMA peak — exactly the LDS bandwidth, so the matrix pipe could
// never be more than ~half busy once the staging writes are added.  128x64 per wave reads 4+2
// fragments for 8 MFMAs (96 B/clk/CU at peak).
//
// Operand roles are swapped with respect to the first kernel: the WEIGHT rows are the MFMA "A"
// operand and the PIXEL rows the "B" operand, so a lane's 16 accumulators of a tile are 4 groups
// of 4 CONSECUTIVE CHANNELS of ONE pixel (C layout: col = lane & 31 = pixel, row = channel).  The
// epilogue therefore writes 8-byte (bf16) / 16-byte (fp32) channel groups into the LDS patch
// instead of 2-byte scalars (the old epilogue spent ~19 VALU/LDS instructions per MFMA), reads
// the patch back as 16-byte NHWC vectors for coalesced stores, and takes the BatchNorm partial
// sums on that read: a thread always sees the same channel vector, so the sums stay lane-local.
// The statistics are those of the values as stored (bf16-rounded on the bf16 path), which is
// what the consumer's normalisation is applied to.
#include "conv_gemm.h"
#include "conv_gemm_args.h"
#include <cstdlib>

namespace seg {

constexpr int PX_BM = 256, PX_BN = 128;
constexpr int PX_SA = PX_BM * ROW_STRIDE;  // pixel operand tile (36864 B)
constexpr int PX_SB = PX_BN * ROW_STRIDE;  // weight operand tile (18432 B)

// (Tried and removed, r01: a two-LDS-stage / two-register-stage variant with one barrier per slab
// and one block per CU — hipcc spread the 128 accumulators over VGPRs AND AGPRs with ~800
// v_accvgpr moves per K loop; 36.7 -> 64.7 us on 728->728.  See profiles/r01_pmc_px256.md.)
template <typename T>
__global__ __launch_bounds__(GEMM_THREADS, 2) void conv_gemm_px256_kernel(const ConvGemmArgs a) {
  constexpr int VEC = Vec<T>::N;
  constexpr int BK = ROW_BYTES / (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) unsigned char smem[PX_SA + PX_SB];
  unsigned char* const sA = smem;
  unsigned char* const sB = smem + PX_SA;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wp = wave >> 1, wc = wave & 1;
  const int L = xcd_remap(blockIdx.x, a.tiles_m * a.tiles_n);
  const int tile_m = L / a.tiles_n, tile_n = L - tile_m * a.tiles_n;
  const int m0 = tile_m * PX_BM, n0 = tile_n * PX_BN;

  // ---- staging assignment: thread -> 16-byte vector column vc of rows rb + 32*j
  const int vc = tid & 7, rb = tid >> 3;
  // operand rows addressed in 16-byte units from the tensor base (32-bit: tensors < 64 GB)
  const uint4* __restrict__ X = reinterpret_cast<const uint4*>(a.x);
  const uint4* __restrict__ W = reinterpret_cast<const uint4*>(a.w);
  const long ldxv = a.ldx / VEC, ldwv = a.K / VEC;
  unsigned aoff[8], boff[4];
  unsigned row_ok = 0, col_ok = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int p = m0 + rb + 32 * j;
    aoff[j] = (unsigned)((long)(p < a.M ? p : 0) * ldxv) + vc;
    if (p < a.M) row_ok |= 1u << j;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int o = n0 + rb + 32 * j;
    boff[j] = (unsigned)((long)(o < a.O ? o : 0) * ldwv) + vc;
    if (o < a.O) col_ok |= 1u << j;
  }

  struct Regs { uint4 a[8]; uint4 b[4]; bool kok; };
  auto load_slab = [&](int kt, Regs& r) {
    r.kok = kt * BK + vc * VEC < a.K;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      r.a[j] = make_uint4(0, 0, 0, 0);
      if (r.kok && ((row_ok >> j) & 1u)) r.a[j] = X[aoff[j] + kt * (BK / VEC)];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r.b[j] = make_uint4(0, 0, 0, 0);
      if (r.kok && ((col_ok >> j) & 1u)) r.b[j] = W[boff[j] + kt * (BK / VEC)];
    }
  };
  auto stage = [&](int kt, Regs& r, unsigned char* dA, unsigned char* dB) {
    // registers -> LDS; the producer's BatchNorm(+ReLU) rides on the pixel operand (rows beyond
    // M and columns beyond K stay exactly zero)
    if (a.pro_mode != PRO_NONE && r.kok) {
      const int c = kt * BK + vc * VEC;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if ((row_ok >> j) & 1u) {
          float f[VEC];
          Vec<T>::unpack(r.a[j], f);
          apply_prologue<VEC>(f, a.pro_mode, a.pro_scale, a.pro_shift, c);
          r.a[j] = Vec<T>::pack(f);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      *reinterpret_cast<uint4*>(dA + (rb + 32 * j) * ROW_STRIDE + vc * 16) = r.a[j];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<uint4*>(dB + (rb + 32 * j) * ROW_STRIDE + vc * 16) = r.b[j];
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int jc = 0; jc < 2; ++jc)
#pragma unroll
    for (int ip = 0; ip < 4; ++ip)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[jc][ip][e] = 0.f;

  const int r32 = lane & 31, hh = lane >> 5;
  const int fragP = (wp * 128 + r32) * ROW_STRIDE + hh * 16;  // byte offsets inside a stage
  const int fragC = (wc * 64 + r32) * ROW_STRIDE + hh * 16;
  // Fragment reads software-pipelined by hand: the six fragments of k-step s+1 are requested
  // before the eight MFMAs of step s are issued, into their own registers.  (Left to itself the
  // compiler re-used ONE register quad for consecutive pixel fragments: ds_read -> wait -> 2 MFMAs
  // -> ds_read ..., i.e. a full LDS round trip in front of every MFMA pair.)
  auto mma = [&](const unsigned char* bA, const unsigned char* bB) {
    uint4 cf[2][2], pf[2][4];
    auto frags = [&](int s, int set) {
      cf[set][0] = *reinterpret_cast<const uint4*>(bB + fragC + s * 32);
      cf[set][1] = *reinterpret_cast<const uint4*>(bB + fragC + 32 * ROW_STRIDE + s * 32);
#pragma unroll
      for (int ip = 0; ip < 4; ++ip)
        pf[set][ip] = *reinterpret_cast<const uint4*>(bA + fragP + ip * 32 * ROW_STRIDE + s * 32);
    };
    frags(0, 0);
#pragma unroll
    for (int s = 0; s < ROW_BYTES / 32; ++s) {
      if (s + 1 < ROW_BYTES / 32) frags(s + 1, (s + 1) & 1);
      // keep those reads ABOVE this step's MFMAs (bf16; the fp32 instantiation has no spare
      // registers for the second fragment set and is left to the compiler)
      if (sizeof(T) == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ip = 0; ip < 4; ++ip) {
        Mma<T>::step(cf[s & 1][0], pf[s & 1][ip], acc[0][ip]);
        Mma<T>::step(cf[s & 1][1], pf[s & 1][ip], acc[1][ip]);
      }
    }
  };
  const int nk = (a.K + BK - 1) / BK;
  {
    Regs r;
    load_slab(0, r);
    for (int kt = 0; kt < nk; ++kt) {
      stage(kt, r, sA, sB);
      __syncthreads();
      if (kt + 1 < nk) load_slab(kt + 1, r);  // global loads in flight under the MFMAs
      mma(sA, sB);
      __syncthreads();
    }
  }

  // ---- epilogue, per wave and 32-pixel tile: channel groups -> LDS patch [32 px][64 ch] ->
  // 16-byte NHWC vectors (+ statistics, + folded-BatchNorm correction) -> global
  constexpr int EP_STRIDE = 64 * (int)sizeof(T) + 16;
  constexpr int VPR = 64 / VEC;  // vectors per patch row
  unsigned char* ep = smem + wave * 32 * EP_STRIDE;
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const int v = lane & (VPR - 1);       // this lane's vector column in every patch row
  const int o = n0 + wc * 64 + v * VEC; // ... = these output channels
  float ssum[VEC], ssq[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) ssum[k] = ssq[k] = 0.f;
  float c0v[VEC], c1v[VEC];
  const bool epc = a.ep_x != nullptr && o + VEC <= a.O;
  if (epc) {
    load_params<VEC>(a.ep_c0, o, c0v);
    load_params<VEC>(a.ep_c1, o, c1v);
  }
#pragma unroll
  for (int ip = 0; ip < 4; ++ip) {
#pragma unroll
    for (int jc = 0; jc < 2; ++jc) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = jc * 32 + 8 * g + 4 * hh;  // first of 4 consecutive channels
        float f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = acc[jc][ip][4 * g + k];
        if (a.bias != nullptr) {
          const int ob = n0 + wc * 64 + ch;
#pragma unroll
          for (int k = 0; k < 4; ++k) f[k] += (ob + k < a.O) ? a.bias[ob + k] : 0.f;
        }
        HVec<T>::store(reinterpret_cast<T*>(ep + r32 * EP_STRIDE) + ch, f);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (32 * VPR) / 64; ++q) {
      const int r = (q * 64 + lane) / VPR;
      const int p = m0 + wp * 128 + ip * 32 + r;
      uint4 val = *reinterpret_cast<const uint4*>(ep + r * EP_STRIDE + v * 16);
      if (a.stat_partial != nullptr) {  // rows beyond M are exact zeros
        float f[VEC];
        Vec<T>::unpack(val, f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          ssum[k] += f[k];
          ssq[k] = fmaf(f[k], f[k], ssq[k]);
        }
      }
      if (p < a.M && o < a.O) {
        T* dst = Y + (long)p * a.ldy + o;
        if (epc) {
          float f[VEC], xv[VEC];
          Vec<T>::unpack(val, f);
          Vec<T>::unpack(ldg16(reinterpret_cast<const T*>(a.ep_x) + (long)p * a.ldep + o), xv);
#pragma unroll
          for (int k = 0; k < VEC; ++k) f[k] = f[k] - c0v[k] - c1v[k] * xv[k];
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
    // lanes sharing a vector column differ in lane bits >= log2(VPR): fold them, then the two
    // pixel halves through LDS: red[wp][2][128]
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
#pragma unroll
      for (int m = VPR; m < 64; m <<= 1) {
        ssum[k] += __shfl_xor(ssum[k], m, 64);
        ssq[k] += __shfl_xor(ssq[k], m, 64);
      }
    }
    float* red = reinterpret_cast<float*>(smem);
    if (lane < VPR) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        red[(wp * 2 + 0) * 128 + wc * 64 + v * VEC + k] = ssum[k];
        red[(wp * 2 + 1) * 128 + wc * 64 + v * VEC + k] = ssq[k];
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int oc = n0 + tid;
      if (oc < a.O) {
        float* dst = a.stat_partial + (long)tile_m * 2 * a.O;
        dst[oc] = red[0 * 128 + tid] + red[2 * 128 + tid];
        dst[a.O + oc] = red[1 * 128 + tid] + red[3 * 128 + tid];
      }
    }
  }
}

int px256_tiles_m(long M) { return (int)((M + PX_BM - 1) / PX_BM); }

int launch_conv_gemm_px256(int dtype, ConvGemmArgs a, hipStream_t stream) {
  a.tiles_m = px256_tiles_m(a.M);
  a.tiles_n = (a.O + PX_BN - 1) / PX_BN;
  const dim3 grid(a.tiles_m * a.tiles_n), block(GEMM_THREADS);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((conv_gemm_px256_kernel<bf16_t>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((conv_gemm_px256_kernel<float>), grid, block, 0, stream, a);
  return check_launch("conv_gemm_fwd (px256)");
}

}  // namespace seg
