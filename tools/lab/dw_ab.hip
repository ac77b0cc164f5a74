// Lab: A/B of two builds of the depthwise entry points over the C3 shapes, without torch.
//   hipcc -O3 --offload-arch=gfx950 -o dw_ab dw_ab.hip -ldl
//   ./dw_ab <baseline libsegmentron_hip.so> <candidate libsegmentron_hip.so>
// For every stride-1 / dilation-1 depthwise shape of DeepLabv3+/xception65 @1025x2049 (batch 2, bf16)
// and the two prologue modes the network uses: forward + BatchNorm partials, fused backward
// (masked data gradient + weight-gradient partials + BatchNorm-backward partials) through BOTH
// libraries on the same inputs — outputs compared bit for bit (a kernel change that keeps the
// arithmetic order must give 0 differing elements; anything else is reported as max |diff|), and
// both timed with HIP events (30 launches after 3 warm-ups).  A gpurun call of this costs seconds:
// the loop for kernel work on dwconv_tiled.hip (keep the baseline .so of HEAD next to the
// candidate: `cp segmentron_amd/libsegmentron_hip.so /tmp/base.so` before editing).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef int (*dw_fn)(int, int, const void*, long, int, int, int, int, const float*, int, int, int, int,
                     const float*, const float*, void*, long, int, int, float*, int, void*);
typedef int (*grid_fn)(int, int, int, int, int, int, int, int);
typedef int (*bwd_fn)(int, const void*, long, const void*, long, int, int, int, int, const float*, int,
                      int, int, const float*, const float*, void*, long, float*, float*, int, void*);
typedef const char* (*err_fn)();

struct Lib { void* h; dw_fn dw; grid_fn grid; bwd_fn bwd; err_fn err; const char* name; };

static int load(Lib& l, const char* path, const char* name) {
  l.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!l.h) { printf("dlopen %s: %s\n", path, dlerror()); return 1; }
  l.dw = (dw_fn)dlsym(l.h, "seg_dwconv3x3");
  l.grid = (grid_fn)dlsym(l.h, "seg_dwconv_grid_y");
  l.bwd = (bwd_fn)dlsym(l.h, "seg_dwconv3x3_bwd_fused");
  l.err = (err_fn)dlsym(l.h, "seg_last_error");
  l.name = name;
  if (!l.dw || !l.grid || !l.bwd || !l.err) { printf("%s: missing symbol\n", path); return 1; }
  return 0;
}

__global__ void fill_bf16(uint32_t* p, long n, uint32_t seed) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    uint32_t h = ((uint32_t)i + seed) * 2654435761u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    // two bf16 in [-2, 2): sign + exponent 0x3f/0x40 region, 7 random mantissa bits each
    const uint32_t lo = 0x3f00u | (h & 0x80ffu), hi = 0x3f00u | ((h >> 16) & 0x80ffu);
    p[i] = lo | (hi << 16);
  }
}

static long count_diff(const std::vector<uint16_t>& a, const std::vector<uint16_t>& b, double& maxd) {
  long n = 0; maxd = 0;
  for (size_t i = 0; i < a.size(); ++i) if (a[i] != b[i]) {
    ++n; union { uint32_t u; float f; } p, q; p.u = (uint32_t)a[i] << 16; q.u = (uint32_t)b[i] << 16;
    maxd = fmax(maxd, fabs((double)p.f - q.f));
  }
  return n;
}
static long count_diff_f(const std::vector<float>& a, const std::vector<float>& b, double& maxd) {
  long n = 0; maxd = 0;
  for (size_t i = 0; i < a.size(); ++i) if (memcmp(&a[i], &b[i], 4)) { ++n; maxd = fmax(maxd, fabs((double)a[i] - b[i])); }
  return n;
}

int main(int argc, char** argv) {
  if (argc < 3) { printf("usage: dw_ab <baseline.so> <candidate.so>\n"); return 2; }
  Lib L[2];
  if (load(L[0], argv[1], "baseline") || load(L[1], argv[2], "candidate")) return 1;
  // [N, H, W, C] of the stride-1 dilation-1 depthwise layers of C3 x launches per step
  const int shapes[][5] = {{2, 513, 1025, 64, 1},  {2, 513, 1025, 128, 1}, {2, 257, 513, 128, 1},
                           {2, 257, 513, 256, 3},  {2, 129, 257, 256, 1},  {2, 129, 257, 728, 1},
                           {2, 65, 129, 728, 50},  {2, 65, 129, 1024, 1},  {2, 257, 513, 304, 1}};
  const int PRO_RELU = 1, PRO_AFFINE_RELU = 3, DT_BF16 = 1;
  double tot[2][2] = {{0, 0}, {0, 0}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* only = getenv("DW_AB_SHAPES");  // e.g. "6" or "1,6": indices into shapes[]
  int shape_idx = -1;
  for (auto& sh : shapes) {
    ++shape_idx;
    if (only) {
      bool hit = false;
      for (const char* q = only; *q; ++q)
        if (*q >= '0' && *q <= '9' && (*q - '0') == shape_idx && (q == only || q[-1] == ',') && (q[1] == 0 || q[1] == ',')) hit = true;
      if (!hit) continue;
    }
    const int N = sh[0], H = sh[1], W = sh[2], C = sh[3], per_step = sh[4];
    const long elems = (long)N * H * W * C;
    void *x, *dy, *y[2], *g[2]; float *w, *sc, *shf;
    CK(hipMalloc(&x, elems * 2)); CK(hipMalloc(&dy, elems * 2));
    for (int k = 0; k < 2; ++k) { CK(hipMalloc(&y[k], elems * 2)); CK(hipMalloc(&g[k], elems * 2)); }
    CK(hipMalloc(&w, 9 * C * 4)); CK(hipMalloc(&sc, C * 4)); CK(hipMalloc(&shf, C * 4));
    hipLaunchKernelGGL(fill_bf16, dim3((elems / 2 + 255) / 256), dim3(256), 0, 0, (uint32_t*)x, elems / 2, 1u);
    hipLaunchKernelGGL(fill_bf16, dim3((elems / 2 + 255) / 256), dim3(256), 0, 0, (uint32_t*)dy, elems / 2, 77u);
    std::vector<float> hw(9 * C), hs(C), ht(C);
    for (int i = 0; i < 9 * C; ++i) hw[i] = 0.05f * (float)((i * 7919) % 13 - 6);
    for (int i = 0; i < C; ++i) { hs[i] = 0.75f + 0.01f * (i % 50); ht[i] = 0.02f * ((i % 11) - 5); }
    CK(hipMemcpy(w, hw.data(), 9 * C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sc, hs.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(shf, ht.data(), C * 4, hipMemcpyHostToDevice));
    printf("[%d, %d, %d, %d] bf16  (%.1f MB per tensor, x%d per step)\n", N, H, W, C, elems * 2 / 1e6, per_step);
    for (int mode : {PRO_AFFINE_RELU, PRO_RELU}) {
      float us[2][2]; long gy[2][2]; float *sp[2], *pw[2], *pb[2];
      for (int k = 0; k < 2; ++k) {
        gy[k][0] = L[k].grid(DT_BF16, C, N, H, W, 1, 1, 0);
        gy[k][1] = L[k].grid(DT_BF16, C, N, H, W, 1, 1, 1);
        CK(hipMalloc(&sp[k], gy[k][0] * 2 * C * 4)); CK(hipMalloc(&pw[k], gy[k][1] * 9 * C * 4));
        CK(hipMalloc(&pb[k], gy[k][1] * 2 * C * 4));
        auto fwd = [&] { return L[k].dw(DT_BF16, 0, x, C, N, H, W, C, w, 0, 1, 1, mode, sc, shf, y[k], C, H, W, sp[k], (int)gy[k][0], nullptr); };
        auto bwd = [&] { return L[k].bwd(DT_BF16, dy, C, x, C, N, H, W, C, w, 0, 1, mode, sc, shf, g[k], C, pw[k], pb[k], (int)gy[k][1], nullptr); };
        for (int which = 0; which < 2; ++which) {
          for (int i = 0; i < 3; ++i) if (which ? bwd() : fwd()) { printf("  %s: %s\n", L[k].name, L[k].err()); return 1; }
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0));
          for (int i = 0; i < 30; ++i) which ? bwd() : fwd();
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          us[k][which] = ms * 1e3f / 30;
          tot[k][which] += us[k][which] * per_step * (mode == PRO_AFFINE_RELU ? 2.0 / 3 : 1.0 / 3);
        }
      }
      // outputs
      std::vector<uint16_t> a(elems), b(elems); double md;
      CK(hipMemcpy(a.data(), y[0], elems * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), y[1], elems * 2, hipMemcpyDeviceToHost));
      const long dyf = count_diff(a, b, md); const double mdf = md;
      CK(hipMemcpy(a.data(), g[0], elems * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), g[1], elems * 2, hipMemcpyDeviceToHost));
      const long dgb = count_diff(a, b, md); const double mdb = md;
      // reduced partials (the partial-row counts of the two builds may differ): column sums in fp64
      auto colsum = [&](float* dev, long R, int L_) { std::vector<float> h(R * L_); (void)hipMemcpy(h.data(), dev, R * L_ * 4, hipMemcpyDeviceToHost);
        std::vector<float> o(L_); for (int c = 0; c < L_; ++c) { double s = 0; for (long r = 0; r < R; ++r) s += h[r * L_ + c]; o[c] = (float)s; } return o; };
      double m1, m2, m3;
      const long d1 = count_diff_f(colsum(sp[0], gy[0][0], 2 * C), colsum(sp[1], gy[1][0], 2 * C), m1);
      const long d2 = count_diff_f(colsum(pw[0], gy[0][1], 9 * C), colsum(pw[1], gy[1][1], 9 * C), m2);
      const long d3 = count_diff_f(colsum(pb[0], gy[0][1], 2 * C), colsum(pb[1], gy[1][1], 2 * C), m3);
      const double gb = 2.0 * elems * 2 / 1e3;  // forward: one read + one write (bytes / 1e3 -> GB/s with us)
      printf("  mode %d  fwd %7.2f -> %7.2f us (%5.0f -> %5.0f GB/s)   bwd %7.2f -> %7.2f us   | y diff %ld (max %.3g)  g diff %ld (max %.3g)"
             "  stats %ld (%.3g)  dW %ld (%.3g)  bn sums %ld (%.3g)\n", mode, us[0][0], us[1][0], gb / us[0][0], gb / us[1][0],
             us[0][1], us[1][1], dyf, mdf, dgb, mdb, d1, m1, d2, m2, d3, m3);
      for (int k = 0; k < 2; ++k) { (void)hipFree(sp[k]); (void)hipFree(pw[k]); (void)hipFree(pb[k]); }
    }
    (void)hipFree(x); (void)hipFree(dy); (void)hipFree(w); (void)hipFree(sc); (void)hipFree(shf);
    for (int k = 0; k < 2; ++k) { (void)hipFree(y[k]); (void)hipFree(g[k]); }
  }
  // optional sweep (argv[3..] = forced partial-row counts): the candidate's forward and fused
  // backward on [2,65,129,728] with the persistent-block count of the caller's choice
  if (argc > 3) {
    const int N = 2, H = 65, W = 129, C = 728; const long elems = (long)N * H * W * C;
    void *x, *dy, *y, *g; float *w, *sc, *shf, *sp, *pw, *pb;
    CK(hipMalloc(&x, elems * 2)); CK(hipMalloc(&dy, elems * 2)); CK(hipMalloc(&y, elems * 2)); CK(hipMalloc(&g, elems * 2));
    CK(hipMalloc(&w, 9 * C * 4)); CK(hipMalloc(&sc, C * 4)); CK(hipMalloc(&shf, C * 4));
    CK(hipMalloc(&sp, 4096l * 2 * C * 4)); CK(hipMalloc(&pw, 4096l * 9 * C * 4)); CK(hipMalloc(&pb, 4096l * 2 * C * 4));
    hipLaunchKernelGGL(fill_bf16, dim3((elems / 2 + 255) / 256), dim3(256), 0, 0, (uint32_t*)x, elems / 2, 1u);
    hipLaunchKernelGGL(fill_bf16, dim3((elems / 2 + 255) / 256), dim3(256), 0, 0, (uint32_t*)dy, elems / 2, 77u);
    CK(hipMemset(w, 0, 9 * C * 4)); CK(hipMemset(sc, 0, C * 4)); CK(hipMemset(shf, 0, C * 4));
    printf("grid sweep on [2,65,129,728] (candidate; its own choice: fwd %d, bwd %d rows)\n",
           L[1].grid(DT_BF16, C, N, H, W, 1, 1, 0), L[1].grid(DT_BF16, C, N, H, W, 1, 1, 1));
    for (int ai = 3; ai < argc; ++ai) {
      const int gy = atoi(argv[ai]);
      if (gy < 1 || gy > 4096) continue;
      float us2[2];
      for (int which = 0; which < 2; ++which) {
        auto run = [&] { return which ? L[1].bwd(DT_BF16, dy, C, x, C, N, H, W, C, w, 0, 1, PRO_AFFINE_RELU, sc, shf, g, C, pw, pb, gy, nullptr)
                                      : L[1].dw(DT_BF16, 0, x, C, N, H, W, C, w, 0, 1, 1, PRO_AFFINE_RELU, sc, shf, y, C, H, W, sp, gy, nullptr); };
        for (int i = 0; i < 3; ++i) if (run()) { printf("  %s\n", L[1].err()); return 1; }
        CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
        for (int i = 0; i < 30; ++i) run();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); us2[which] = ms * 1e3f / 30;
      }
      printf("  rows %4d: fwd %7.2f us  fused bwd %7.2f us\n", gy, us2[0], us2[1]);
    }
  }
  printf("per-step estimate (launch counts of C3, 2/3 of the launches with the affine prologue): forward %.2f -> %.2f ms, fused backward %.2f -> %.2f ms\n",
         tot[0][0] / 1e3, tot[1][0] / 1e3, tot[0][1] / 1e3, tot[1][1] / 1e3);
  return 0;
}
