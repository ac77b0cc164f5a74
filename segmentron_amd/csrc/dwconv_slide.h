// Register-sliding depthwise 3x3 (stride 1, dilation 1) — see dwconv_slide.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace seg {
bool dw_slide_supported(int stride, int dil, int C);
// partial rows one launch writes (forward statistics / backward partials): image x strip x column block
int dw_slide_rows(int C, int N, int H, int W);
int launch_dw_slide_fwd(int dtype, const void* x, long ldx, int N, int H, int W, int C,
                        const float* w, int w_layout, int pro_mode, const float* sc,
                        const float* sh, void* y, long ldy, float* stat_partial, int rows,
                        hipStream_t st);
int launch_dw_slide_bwd(int dtype, const void* dy, long lddy, const void* x, long ldx, int N, int H,
                        int W, int C, const float* w, int w_layout, int pro_mode, const float* sc,
                        const float* sh, void* g, long ldg, float* partial_w, float* partial_bn,
                        int rows, hipStream_t st, const void* res = nullptr, long ldr = 0);
}  // namespace seg
