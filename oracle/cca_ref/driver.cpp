// TEST INFRASTRUCTURE ONLY (oracle/): host harness that runs the REFERENCE's own criss-cross
// attention kernels — the six `__global__` templates of
// /root/reference/segmentron/modules/csrc/criss_cross_attention/ca_cuda.cu:8-177 — on the CPU.
//
// Nothing of the reference is copied into this repository: oracle/cca_ref/build.sh extracts the
// kernel section of ca_cuda.cu (everything above `namespace segmentron {`, minus the ATen / THC
// #includes) from the reference checkout into oracle/_ref/ca_kernels.inc at build time and
// compiles this file around it.  `__global__` is defined away and blockIdx / blockDim /
// threadIdx become plain globals that `launch()` sweeps over the grid the reference's host
// functions use (ca_cuda.cu:196-209, 229-252, 269-283, 301-328: 32 x 32 threads,
// grid = (ceil(w/32), ceil(h/32), h + w | c)), one "thread" after another.  Every output
// element is accumulated by exactly one thread, so the sequential sweep reproduces the CUDA
// result operation for operation.
#include <cstring>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
static dim3 blockIdx, blockDim, threadIdx;
#define __global__

#include "ca_kernels.inc"

template <typename F>
static void launch(dim3 grid, dim3 block, F&& body) {
  blockDim = block;
  for (blockIdx.z = 0; blockIdx.z < grid.z; ++blockIdx.z)
    for (blockIdx.y = 0; blockIdx.y < grid.y; ++blockIdx.y)
      for (blockIdx.x = 0; blockIdx.x < grid.x; ++blockIdx.x)
        for (threadIdx.y = 0; threadIdx.y < block.y; ++threadIdx.y)
          for (threadIdx.x = 0; threadIdx.x < block.x; ++threadIdx.x) body();
}

static dim3 grid_of(int h, int w, int d3) { return dim3((w + 31) / 32, (h + 31) / 32, d3); }

#define CCA_REF_API(T, SFX)                                                                       \
  extern "C" void cca_ref_forward_##SFX(const T* t, const T* f, T* weight, int n, int c, int h,   \
                                        int w) {                                                  \
    std::memset(weight, 0, sizeof(T) * (size_t)n * (h + w - 1) * h * w);                          \
    launch(grid_of(h, w, h + w), dim3(32, 32),                                                    \
           [&] { ca_forward_kernel<T>(t, f, weight, n, c, h, w); });                              \
  }                                                                                               \
  extern "C" void cca_ref_backward_##SFX(const T* dw, const T* t, const T* f, T* dt, T* df,       \
                                         int n, int c, int h, int w) {                            \
    std::memset(dt, 0, sizeof(T) * (size_t)n * c * h * w);                                        \
    std::memset(df, 0, sizeof(T) * (size_t)n * c * h * w);                                        \
    launch(grid_of(h, w, c), dim3(32, 32),                                                        \
           [&] { ca_backward_kernel_t<T>(dw, t, f, dt, n, c, h, w); });                           \
    launch(grid_of(h, w, c), dim3(32, 32),                                                        \
           [&] { ca_backward_kernel_f<T>(dw, t, f, df, n, c, h, w); });                           \
  }                                                                                               \
  extern "C" void cca_ref_map_forward_##SFX(const T* weight, const T* g, T* out, int n, int c,    \
                                            int h, int w) {                                       \
    std::memset(out, 0, sizeof(T) * (size_t)n * c * h * w);                                       \
    launch(grid_of(h, w, c), dim3(32, 32),                                                        \
           [&] { ca_map_forward_kernel<T>(weight, g, out, n, c, h, w); });                        \
  }                                                                                               \
  extern "C" void cca_ref_map_backward_##SFX(const T* dout, const T* weight, const T* g, T* dw,   \
                                             T* dg, int n, int c, int h, int w) {                 \
    std::memset(dw, 0, sizeof(T) * (size_t)n * (h + w - 1) * h * w);                              \
    std::memset(dg, 0, sizeof(T) * (size_t)n * c * h * w);                                        \
    launch(grid_of(h, w, h + w), dim3(32, 32),                                                    \
           [&] { ca_map_backward_kernel_w<T>(dout, weight, g, dw, n, c, h, w); });                \
    launch(grid_of(h, w, c), dim3(32, 32),                                                        \
           [&] { ca_map_backward_kernel_g<T>(dout, weight, g, dg, n, c, h, w); });                \
  }

CCA_REF_API(float, f32)
CCA_REF_API(double, f64)
