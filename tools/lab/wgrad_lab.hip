// Stand-alone A/B harness for the weight-gradient GEMM kernels through the C-ABI
// (seg_conv_gemm_wgrad): first-generation register-transpose kernel vs the direct-to-LDS
// transpose-read kernel.  dW[o][c] = sum_p dy[p][o] * x[p][c] checked against a naive device
// reference after summing the split partials in order; timing with hipEvents.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/segmentron_hip.h"
typedef unsigned short bf16_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static inline bf16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }

__global__ void ref_wgrad(const bf16_t* x, long ldx, const bf16_t* dy, long lddy, int M, int C, int O, float* dw) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)O * C) return;
  const int o = (int)(i / C), c = (int)(i % C);
  float acc = 0.f;
  for (int p = 0; p < M; ++p)
    acc = fmaf(__uint_as_float(((unsigned)dy[(long)p * lddy + o]) << 16),
               __uint_as_float(((unsigned)x[(long)p * ldx + c]) << 16), acc);
  dw[i] = acc;
}
__global__ void sum_splits(const float* part, int S, long n, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int s = 0; s < S; ++s) a += part[(long)s * n + i];
  out[i] = a;
}
struct Shape { int M, C, O; const char* name; };
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const Shape shapes[] = {
      {16770, 728, 728, "middle flow 728x728 @2x65x129"},
      {16770, 1536, 2048, "exit 1536->2048"},
      {16770, 1024, 1536, "exit 1024->1536"},
      {16770, 728, 1024, "exit 728->1024"},
      {66306, 256, 728, "entry 256->728 @2x129x257"},
      {263682, 128, 256, "entry 128->256 @2x257x513"},
      {4290, 728, 728, "middle flow @2x33x65"},
      {5003, 200, 392, "ragged"},
  };
  size_t maxX = 0, maxD = 0, maxW = 0;
  for (const Shape& s : shapes) { maxX = std::max(maxX, (size_t)s.M * (s.C + 8)); maxD = std::max(maxD, (size_t)s.M * (s.O + 16)); maxW = std::max(maxW, (size_t)s.O * s.C); }
  bf16_t *dx, *ddy; float *dpart, *dref, *dsum;
  CK(hipMalloc(&dx, maxX * 2)); CK(hipMalloc(&ddy, maxD * 2));
  CK(hipMalloc(&dpart, (size_t)300 << 20)); CK(hipMalloc(&dref, maxW * 4)); CK(hipMalloc(&dsum, maxW * 4));
  std::vector<bf16_t> hx(maxX), hd(maxD);
  srand(2);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : hx) v = f2bf(rnd());
  for (auto& v : hd) v = f2bf(rnd());
  CK(hipMemcpy(dx, hx.data(), maxX * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(ddy, hd.data(), maxD * 2, hipMemcpyHostToDevice));
  std::vector<float> href(maxW), hsum(maxW);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int bad = 0;
  for (const Shape& s : shapes) {
    const long ldx = s.C + 8, lddy = s.O + 16;  // channel-slice pitches
    const long n = (long)s.O * s.C;
    hipLaunchKernelGGL(ref_wgrad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, ldx, ddy, lddy, s.M, s.C, s.O, dref);
    CK(hipMemcpy(href.data(), dref, n * 4, hipMemcpyDeviceToHost));
    double scale = 0; for (long i = 0; i < n; ++i) scale = std::max(scale, (double)fabsf(href[i]));
    float ms[2] = {0, 0}; int splits[2] = {0, 0};
    for (int which = 0; which < 2; ++which) {  // 0 = first generation, 1 = glds
      seg_conv_gemm_wgrad_config(which ? 0 : 2);
      const int S = seg_conv_gemm_wgrad_splits(1, 1, 1, s.M, s.C, s.O, 1, 1, 1, 0, 1, 0);
      splits[which] = S;
      if ((size_t)S * n * 4 > ((size_t)300 << 20)) { printf("partials do not fit\n"); bad++; continue; }
      CK(hipMemset(dpart, 0xFF, (size_t)S * n * 4));
      auto run = [&] { return seg_conv_gemm_wgrad(1, dx, ldx, 1, 1, s.M, s.C, ddy, lddy, 1, s.M, s.O, 1, 1, 1, 0, 1, 0, nullptr, nullptr, dpart, S, nullptr); };
      int rc = run();
      CK(hipDeviceSynchronize());
      if (rc) { printf("launch failed: %s\n", seg_last_error()); bad++; continue; }
      hipLaunchKernelGGL(sum_splits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dpart, S, n, dsum);
      CK(hipMemcpy(hsum.data(), dsum, n * 4, hipMemcpyDeviceToHost));
      double err = 0; long nanc = 0;
      for (long i = 0; i < n; ++i) { if (!(hsum[i] == hsum[i])) { nanc++; continue; } err = std::max(err, (double)fabs(hsum[i] - href[i])); }
      const bool ok = nanc == 0 && err <= 2e-3 * scale;
      if (!ok) bad++;
      printf("%-32s %-5s splits %2d max|err|/scale %.2e nan %ld %s\n", s.name, which ? "glds" : "gen1", S, err / scale, nanc, ok ? "ok" : "FAIL");
      for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 2; ++i) run();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) run();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (rep > 0) ms[which] += t / iters / 2;
      }
    }
    const double fl = 2.0 * s.M * s.C * s.O;
    printf("  TIME %-32s gen1 %7.1f us %5.0f TF (%2d splits) | glds %7.1f us %5.0f TF (%2d splits) | x%.2f\n", s.name,
           ms[0] * 1e3, fl / ms[0] / 1e9, splits[0], ms[1] * 1e3, fl / ms[1] / 1e9, splits[1], ms[0] / ms[1]);
  }
  printf(bad ? "LAB FAILED (%d)\n" : "LAB OK\n", bad);
  return bad ? 1 : 0;
}
