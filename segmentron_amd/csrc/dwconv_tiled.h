// LDS-tiled depthwise 3x3 (stride 1, dilation 1/2) — see dwconv_tiled.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace seg {
bool dw_tiled_supported(int stride, int dil);
int dw_tiled_grid_y(int dtype, int C, int N, int H, int W, int kind);
int launch_dw_tiled(int dtype, const void* x, long ldx, int N, int H, int W, int C,
                    const float* w, int w_layout, int dil, int pro_mode, const float* sc,
                    const float* sh, void* y, long ldy, float* stat_partial, int grid_y,
                    hipStream_t st);
// stride 2, pad 1, dilation 1 (H x W: input size)
int dw_tiled_s2_grid_y(int dtype, int C, int N, int Ho, int Wo);
int launch_dw_tiled_s2(int dtype, const void* x, long ldx, int N, int H, int W, int C,
                       const float* w, int w_layout, int pro_mode, const float* sc,
                       const float* sh, void* y, long ldy, float* stat_partial, int grid_y,
                       hipStream_t st);
int launch_dw_bwd_tiled(int dtype, const void* dy, long lddy, const void* x, long ldx, int N, int H,
                        int W, int C, const float* w, int w_layout, int dil, int pro_mode,
                        const float* sc, const float* sh, void* g, long ldg, float* partial_w,
                        float* partial_bn, int grid_y, hipStream_t st,
                        const void* res = nullptr, long ldr = 0);
int launch_dw_wgrad_finalize(const float* partial, int R, int C, float* out, hipStream_t st);
int launch_dw_wgrad_tiled(int dtype, const void* x, long ldx, int N, int H, int W, int C,
                          const void* dy, long lddy, int dil, int pro_mode, const float* sc,
                          const float* sh, float* partial, int grid_y, hipStream_t st);
// row-segment kernels for stride 1 with wide dilation (dwconv_row.hip)
bool dw_row_supported(int stride, int dil);
int dw_row_grid_y(int dtype, int C, int N, int H, int W, int dil);
int launch_dw_row_fwd(int dtype, const void* x, long ldx, int N, int H, int W, int C,
                      const float* w9c, int dil, int pro_mode, const float* sc, const float* sh,
                      void* y, long ldy, float* stat_partial, int grid_y, hipStream_t st);
int launch_dw_row_bwd(int dtype, const void* dy, long lddy, const void* x, long ldx, int N, int H,
                      int W, int C, const float* w9c, int dil, int pro_mode, const float* sc,
                      const float* sh, void* g, long ldg, float* partial_w, float* partial_bn,
                      int grid_y, hipStream_t st);
}  // namespace seg
