// Argument block shared by the weight-gradient GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace seg {

struct WgradArgs {
  const void* x;
  const void* dy;
  float* partial;  // [splits][O][K]
  const float* pro_scale;
  const float* pro_shift;
  long ldx, lddy;
  int N, Hi, Wi, C, Ho, Wo, O;
  int KH, KW, stride, pad, dil;
  int pro_mode;
  int M, K;
  int tiles_o, tiles_k, splits;
  int chunk;  // pixels per split (multiple of the slab depth)
};

// direct-to-LDS 128x128 kernel with LDS transpose reads (conv_gemm_wgrad_glds.hip): bf16, 1x1
// stride 1, no prologue
bool conv_wgrad_glds_usable(int dtype, const WgradArgs& a);
int conv_wgrad_glds_splits(long M, int O, int K);
int launch_conv_wgrad_glds(WgradArgs a, hipStream_t stream);

// direct halo-tile kernel for 3x3 stride-1 stems with 32 input channels (conv3x3_direct.hip)
bool conv3x3_wgrad_direct_usable(int dtype, int C, int O, int KH, int KW, int stride, int pad,
                                 int dil, long M, long ldx, long lddy);
int conv3x3_direct_blocks(int N, int H, int W, int C = 32);
int launch_conv3x3_wgrad_direct(const void* x, long ldx, const void* dy, long lddy, int N, int H,
                                int W, int O, int pro_mode, const float* pro_scale,
                                const float* pro_shift, float* partial, hipStream_t stream);

}  // namespace seg
