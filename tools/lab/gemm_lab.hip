// Stand-alone A/B harness for the 1x1-convolution GEMM kernels (no Python / torch: a gpurun call
// costs seconds instead of the 1-2 minutes of a first `import torch`).  Links the production
// objects, so what is measured and checked is exactly what libsegmentron_hip.so runs.
//
//   make -C segmentron_amd/csrc && hipcc --offload-arch=gfx950 -O2 -std=c++17 \
//       tools/lab/gemm_lab.hip segmentron_amd/csrc/{core,conv_gemm_px256,conv_gemm_glds}.o \
//       -o tools/lab/gemm_lab
//   tools/lab/gemm_lab [iters]
//
// For every shape: both kernels against a naive fp32-accumulating device reference on random
// bf16 data (max error normalised by the output scale; statistics rows; bias; folded-BN
// epilogue correction; channel-slice output), then time (hipEvent, `iters` launches each,
// interleaved A/B/A/B to share clock state) and print TFLOP/s.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../segmentron_amd/csrc/conv_gemm_args.h"

using namespace seg;
typedef unsigned short bf16_t;
namespace seg { int launch_conv_gemm_glds_variant(ConvGemmArgs a, hipStream_t stream, int variant); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static inline bf16_t f2bf(float f) {
  unsigned u; memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
static inline float bf2f(bf16_t b) { unsigned u = ((unsigned)b) << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void ref_gemm(const bf16_t* x, long ldx, const bf16_t* w, int M, int K, int O,
                         const float* bias, float* y) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * O) return;
  const int p = (int)(i / O), o = (int)(i % O);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const float a = __uint_as_float(((unsigned)x[(long)p * ldx + k]) << 16);
    const float b = __uint_as_float(((unsigned)w[(long)o * K + k]) << 16);
    acc = fmaf(a, b, acc);
  }
  y[i] = acc + (bias ? bias[o] : 0.f);
}

struct Shape { int M, K, O; const char* name; };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const Shape shapes[] = {
      {16770, 728, 728, "middle flow 728->728 @2x65x129"},
      {16770, 1536, 2048, "exit 1536->2048"},
      {16770, 1024, 1536, "exit 1024->1536"},
      {16770, 1536, 1536, "exit 1536->1536"},
      {16770, 728, 1024, "exit 728->1024"},
      {16770, 2048, 1536, "dgrad 2048->1536"},
      {4290, 728, 728, "middle flow @2x33x65 (513x1025 input)"},
      {66306, 256, 728, "entry 256->728 @2x129x257"},
      {5001, 200, 392, "ragged M/K (O % 8 == 0)"},
  };
  size_t maxX = 0, maxW = 0, maxY = 0;
  for (const Shape& s : shapes) {
    maxX = std::max(maxX, (size_t)s.M * (s.K + 16));
    maxW = std::max(maxW, (size_t)s.O * s.K);
    maxY = std::max(maxY, (size_t)s.M * (s.O + 24));
  }
  bf16_t *dx, *dw, *dy, *dep;
  float *dref, *dbias, *dc0, *dc1, *dstat;
  CK(hipMalloc(&dx, maxX * 2)); CK(hipMalloc(&dw, maxW * 2)); CK(hipMalloc(&dy, maxY * 2));
  CK(hipMalloc(&dep, maxY * 2)); CK(hipMalloc(&dref, maxY * 4));
  CK(hipMalloc(&dbias, 4096 * 4)); CK(hipMalloc(&dc0, 4096 * 4)); CK(hipMalloc(&dc1, 4096 * 4));
  CK(hipMalloc(&dstat, 512 * 2 * 4096 * 4));
  std::vector<bf16_t> hx(maxX), hw(maxW), hep(maxY), hy(maxY);
  std::vector<float> href(maxY), hb(4096), hc0(4096), hc1(4096), hstat(512 * 2 * 4096);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : hx) v = f2bf(rnd());
  for (auto& v : hw) v = f2bf(rnd() * 0.05f);
  for (auto& v : hep) v = f2bf(rnd());
  for (int i = 0; i < 4096; ++i) { hb[i] = rnd(); hc0[i] = rnd() * 0.1f; hc1[i] = rnd() * 0.1f; }
  CK(hipMemcpy(dx, hx.data(), maxX * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), maxW * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dep, hep.data(), maxY * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dbias, hb.data(), 4096 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dc0, hc0.data(), 4096 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dc1, hc1.data(), 4096 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int bad = 0;
  // LAB_SHAPES="0,3": only these shape indices; LAB_NOTEST=1: timing only; LAB_WHICH=0|1: time
  // only px256 / glds (for rocprofv3 --pmc runs)
  const char* only = getenv("LAB_SHAPES");
  const bool notest = getenv("LAB_NOTEST") != nullptr;
  const int only_which = getenv("LAB_WHICH") ? atoi(getenv("LAB_WHICH")) : -1;
  int sidx = -1;
  for (const Shape& s : shapes) {
    ++sidx;
    if (only) {
      char key[8]; snprintf(key, sizeof key, "%d", sidx);
      bool hit = false;
      for (const char* p = only; *p; ) { if (atoi(p) == sidx) hit = true; while (*p && *p != ',') ++p; if (*p) ++p; }
      if (!hit) continue;
    }
    // feature set 0: plain + statistics; 1: bias + slice output (pitch O+24, input pitch K+16);
    // 2: folded-BN epilogue correction
    const int alt = getenv("GL_ALT") ? atoi(getenv("GL_ALT")) : 1;  // variant in the 'v1' column
    for (int feat = 0; feat < 3; ++feat) {
      const long ldx = feat == 1 ? s.K + 16 : s.K, ldy = feat == 1 ? s.O + 24 : s.O;
      if (feat == 2 && s.O % 8) continue;
      if (notest && feat != 0) continue;
      ConvGemmArgs a;
      memset(&a, 0, sizeof(a));
      a.x = dx; a.w = dw; a.y = dy; a.ldx = ldx; a.ldy = ldy;
      a.N = 1; a.Hi = 1; a.Wi = s.M; a.Ho = 1; a.Wo = s.M; a.C = s.K; a.O = s.O;
      a.KH = a.KW = 1; a.stride = 1; a.dil = 1; a.M = s.M; a.K = s.K;
      a.out_H = 1; a.out_W = s.M; a.out_s = 1;
      a.bias = nullptr;  // (convolutions with a bias stay on the px256 kernel)
      a.stat_partial = feat == 0 ? dstat : nullptr;
      if (feat == 2) { a.ep_x = dep; a.ldep = s.O; a.ep_c0 = dc0; a.ep_c1 = dc1; }
      hipLaunchKernelGGL(ref_gemm, dim3((unsigned)(((long)s.M * s.O + 255) / 256)), dim3(256), 0, 0,
                         dx, ldx, dw, s.M, s.K, s.O, a.bias, dref);
      CK(hipMemcpy(href.data(), dref, (size_t)s.M * s.O * 4, hipMemcpyDeviceToHost));
      double scale = 0;
      for (size_t i = 0; i < (size_t)s.M * s.O; ++i) scale = std::max(scale, (double)fabsf(href[i]));
      for (int which = 0; which < 2 && !notest; ++which) {  // 0 = px256, 1 = glds
        if (which == 1 && !conv_gemm_glds_usable(1, a)) { printf("glds not usable?\n"); continue; }
        CK(hipMemset(dy, 0xFF, (size_t)s.M * ldy * 2));  // NaN pattern: unwritten outputs show up
        CK(hipMemset(dstat, 0, 512 * 2 * 4096 * 4));
        int rc = which ? launch_conv_gemm_glds_variant(a, 0, alt > 1 ? alt : 3) : launch_conv_gemm_px256(1, a, 0);
        CK(hipDeviceSynchronize());
        if (rc) { printf("launch failed rc=%d\n", rc); bad++; continue; }
        CK(hipMemcpy(hy.data(), dy, (size_t)s.M * ldy * 2, hipMemcpyDeviceToHost));
        double err = 0; long nanc = 0, slice_touched = 0;
        for (int p = 0; p < s.M; ++p) {
          for (int o = 0; o < s.O; ++o) {
            double r = href[(size_t)p * s.O + o];
            if (feat == 2) r = r - hc0[o] - (double)hc1[o] * bf2f(hep[(size_t)p * s.O + o]);
            const float g = bf2f(hy[(size_t)p * ldy + o]);
            if (!(g == g)) { nanc++; continue; }
            err = std::max(err, fabs(g - r));
          }
          for (long o = s.O; o < ldy; ++o) slice_touched += hy[(size_t)p * ldy + o] != 0xFFFF;
        }
        double serr = 0;
        if (feat == 0) {
          const int rows = (s.M + 255) / 256;
          CK(hipMemcpy(hstat.data(), dstat, (size_t)rows * 2 * s.O * 4, hipMemcpyDeviceToHost));
          for (int o = 0; o < s.O; o += 37) {
            double s1 = 0, s2 = 0, r1 = 0, r2 = 0;
            for (int t = 0; t < rows; ++t) { s1 += hstat[((size_t)t * 2) * s.O + o]; s2 += hstat[((size_t)t * 2 + 1) * s.O + o]; }
            for (int p = 0; p < s.M; ++p) { const double v = bf2f(hy[(size_t)p * ldy + o]); r1 += v; r2 += v * v; }
            serr = std::max(serr, std::max(fabs(s1 - r1) / (fabs(r1) + s.M * 1e-3 * scale), fabs(s2 - r2) / (r2 + 1e-9)));
          }
        }
        const bool ok = nanc == 0 && slice_touched == 0 && err <= 8e-3 * scale && serr < 1e-4;
        if (!ok) bad++;
        printf("%-40s feat %d %-5s max|err|/scale %.2e stat %.1e nan %ld oob %ld %s\n", s.name, feat,
               which ? "ring" : "px256", err / scale, serr, nanc, slice_touched, ok ? "ok" : "FAIL");
      }
      if (feat != 0) continue;
      // ---- timing, interleaved: 0 = px256, 1 = glds variant 0, 2 = glds variant 1 (production)
      // which: 0 = px256, 1 = glds variant 1 (two 64-k stages), 2 = glds variant 3 (ring)
      auto run = [&](int which) {
        return which == 0 ? launch_conv_gemm_px256(1, a, 0)
                          : launch_conv_gemm_glds_variant(a, 0, which == 1 ? alt : 3);
      };
      float ms[3] = {0, 0, 0};
      for (int rep = 0; rep < 3; ++rep)
        for (int which = 0; which < 3; ++which) {
          if (only_which >= 0 && which != only_which) continue;
          for (int i = 0; i < 2; ++i) run(which);
          CK(hipEventRecord(e0, 0));
          for (int i = 0; i < iters; ++i) run(which);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float t; CK(hipEventElapsedTime(&t, e0, e1));
          if (rep > 0) ms[which] += t / iters / 2;
        }
      if (getenv("GL_ABLATE")) {  // glds kernel: full / no epilogue / one K tile only
        const int modes[8] = {1, 101, 102, 104, 104, 106, 107, 108};
        float t3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int rep = 0; rep < 3; ++rep)
          for (int m = 0; m < 8; ++m) {
            a.dil = modes[m];
            for (int i = 0; i < 2; ++i) launch_conv_gemm_glds(a, 0);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) launch_conv_gemm_glds(a, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (rep > 0) t3[m] += t / iters / 2;
          }
        a.dil = 1;
        // ring variants: 3 full, 4 no in-loop DMA, 5 no MFMA (garbage results, timing only)
        float tv[3] = {0, 0, 0};
        for (int rep = 0; rep < 3; ++rep)
          for (int m = 0; m < 3; ++m) {
            for (int i = 0; i < 2; ++i) launch_conv_gemm_glds_variant(a, 0, 3 + m);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) launch_conv_gemm_glds_variant(a, 0, 3 + m);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (rep > 0) tv[m] += t / iters / 2;
          }
        printf("  ABLATE %-38s stores: none %6.1f us | vmcnt(0) after each %6.1f | vmcnt(2) %6.1f | vmcnt(0) per 32-px group %6.1f us\n", s.name, t3[3] * 1e3, t3[5] * 1e3, t3[6] * 1e3, t3[7] * 1e3);
        printf("  ABLATE %-38s v1: full %6.1f us | no epilogue %6.1f us | one K tile + epilogue %6.1f us || ring: full %6.1f | no in-loop DMA %6.1f | no MFMA %6.1f us\n",
               s.name, t3[0] * 1e3, t3[1] * 1e3, t3[2] * 1e3, tv[0] * 1e3, tv[1] * 1e3, tv[2] * 1e3);
      }
      const double fl = 2.0 * s.M * s.K * s.O;
      printf("  TIME %-38s px256 %6.1f us %5.0f TF | glds v1 %6.1f us %5.0f TF | glds ring %6.1f us %5.0f TF | x%.2f\n",
             s.name, ms[0] * 1e3, fl / ms[0] / 1e9, ms[1] * 1e3, fl / ms[1] / 1e9, ms[2] * 1e3,
             fl / ms[2] / 1e9, ms[0] / ms[2]);
    }
  }
  printf(bad ? "LAB FAILED (%d)\n" : "LAB OK\n", bad);
  return bad ? 1 : 0;
}
