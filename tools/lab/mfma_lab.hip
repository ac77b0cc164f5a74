// Lab: sustained v_mfma_f32_32x32x16_bf16 rate (no memory traffic) -> effective clock / peak.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 4096;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 4 * iters * 8;
    const double flop = mfma * 32.0 * 32 * 16 * 2;
    printf("%d wave(s)/SIMD: %.3f ms, %.0f TFLOP/s, %.1f ns per MFMA per wave-slot\n", waves_per_simd, ms,
           flop / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * 8.0 * waves_per_simd));
  }
  return 0;
}
