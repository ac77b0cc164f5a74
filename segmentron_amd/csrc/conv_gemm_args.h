// Argument block shared by the implicit-GEMM forward/dgrad kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace seg {

struct ConvGemmArgs {
  const void* x;
  const void* w;
  void* y;
  const float* pro_scale;
  const float* pro_shift;
  const float* bias;
  float* stat_partial;  // [tiles_m][2][O] or null
  // optional epilogue correction (data gradient through a folded BatchNorm, fold.hip):
  //   y[p][o] = acc - ep_c0[o] - ep_c1[o] * ep_x[p][o]      (ep_x addressed like y)
  const void* ep_x;
  const float* ep_c0;
  const float* ep_c1;
  long ldx, ldy, ldep;
  int N, Hi, Wi, C, Ho, Wo, O;
  int KH, KW, stride, pad, dil;
  int pro_mode;
  int tconv;  // 1: transposed-stride gather (data gradient of a strided KxK convolution)
  int M, K;
  int out_H, out_W, out_s;  // output row scatter geometry (out_s == 1 -> dense rows)
  int tiles_m, tiles_n;
};

int px256_tiles_m(long M);
int launch_conv_gemm_px256(int dtype, ConvGemmArgs a, hipStream_t stream);
// direct-to-LDS 256x256 kernel (conv_gemm_glds.hip): bf16, 1x1 stride 1, no prologue
bool conv_gemm_glds_usable(int dtype, const ConvGemmArgs& a);
int launch_conv_gemm_glds(ConvGemmArgs a, hipStream_t stream);
// the same pipeline as an implicit GEMM for stride-1 KxK convolutions (C % 32 == 0, no prologue)
bool conv_gemm_glds_kxk_usable(int dtype, const ConvGemmArgs& a);
int launch_conv_gemm_glds_kxk(ConvGemmArgs a, hipStream_t stream);

// the four-wave generation of the same pipeline (conv_gemm_glds4.hip, 1x1 only); rows = 224 per
// tile, tiles_m / tiles_n already set for it
int launch_conv_gemm_glds4(const ConvGemmArgs& a, int rows, hipStream_t stream);

// r05: 256- or 192-row tiles (r06: or 224, 1x1 only), whichever fills the 256 CUs in fewer /
// shorter rounds; a statistics row describes glds_rows_per_tile(M, O, kxk) pixels
int glds_rows_per_tile(long M, int O, bool kxk);
int glds_tiles_m(long M, int O, bool kxk);

// direct 3x3 stride-1 kernel for few channels at large spatial sizes (conv3x3_direct.hip)
bool conv3x3_direct_usable(int dtype, const ConvGemmArgs& a);
int conv3x3_direct_blocks(int N, int H, int W, int C = 32);
int launch_conv3x3_direct(const ConvGemmArgs& a, hipStream_t stream);

}  // namespace seg
