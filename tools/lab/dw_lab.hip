// Lab: the first LDS-tiled depthwise 3x3 prototype (stride 1, dil 1) vs the production entry point
// (which has since become the tiled kernel itself: kept as the record of the experiment).
// hipcc -O3 --offload-arch=gfx950 -DOCC=4 -o dw_lab dw_lab.hip -L../../segmentron_amd -lsegmentron_hip -Wl,-rpath,'$ORIGIN/../../segmentron_amd'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
extern "C" int seg_dwconv3x3(int dtype, int mode, const void* x, long ldx, int N, int Hi, int Wi, int C,
                             const float* w9c, int w_layout, int stride, int dil, int pro_mode, const float* ps,
                             const float* pt, void* y, long ldy, int Ho, int Wo, float* stat_partial,
                             int grid_y, void* stream);
extern "C" int seg_dwconv_grid_y(int dtype, int C, int N, int Ho, int Wo, int stride, int dil, int kind);
extern "C" const char* seg_last_error();

__device__ __forceinline__ void unpack(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xFFFF0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xFFFF0000u);
}
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk(float lo, float hi) {
  f32x2_t f = {lo, hi}; union { bf16x2_t v; uint32_t u; } c; c.v = __builtin_convertvector(f, bf16x2_t); return c.u;
}
__device__ __forceinline__ uint4 pack(const float* f) {
  return make_uint4(pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7]));
}

#ifndef OCC
#define OCC 4
#endif
constexpr int TH = 8, TW = 16, CVB = 8;            // output tile 8 x 16 pixels x 8 channel vectors
constexpr int IH = TH + 2, IW = TW + 2, IWP = IW + 1;  // input tile + one pad pixel per row (bank spread)

struct Args {
  const uint16_t* x; const float* w; uint16_t* y; const float* sc; const float* sh; float* stat;
  int N, H, W, C, CV, mode, tiles_h, tiles_w;
};

__global__ __launch_bounds__(256, OCC) void dw_lds(const Args a) {
  __shared__ uint4 tile[IH * IWP * CVB];   // activated, packed bf16
  __shared__ float4 wsm[9 * CVB * 2];
  const int tid = threadIdx.x;
  const int cvb0 = blockIdx.x * CVB;
  int t = blockIdx.y;
  const int tw = t % a.tiles_w; t /= a.tiles_w;
  const int th = t % a.tiles_h; const int n = t / a.tiles_h;
  const int h0 = th * TH, w0 = tw * TW;
  // ---- weights -> LDS
  if (tid < 9 * CVB * 2) {
    const int tap = tid / (CVB * 2), q = tid % (CVB * 2);
    const int c = cvb0 * 8 + q * 4;
    wsm[tid] = c < a.C ? *(const float4*)(a.w + tap * a.C + c) : make_float4(0, 0, 0, 0);
  }
  // ---- input tile -> LDS (activation applied once per element)
  {
    const int cx = tid & (CVB - 1);
    const int cv = cvb0 + cx;
    float sc[8], sh[8];
    if ((a.mode & 2) && cv < a.CV) {
      for (int i = 0; i < 8; i += 4) {
        float4 v = *(const float4*)(a.sc + cv * 8 + i); sc[i] = v.x; sc[i+1] = v.y; sc[i+2] = v.z; sc[i+3] = v.w;
        v = *(const float4*)(a.sh + cv * 8 + i); sh[i] = v.x; sh[i+1] = v.y; sh[i+2] = v.z; sh[i+3] = v.w;
      }
    } else { for (int i = 0; i < 8; ++i) { sc[i] = 1.f; sh[i] = 0.f; } }
    constexpr int NPIX = IH * IW;                  // 180
    constexpr int PER = (NPIX * CVB + 255) / 256;  // 6
    uint4 raw[PER]; bool ok[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int p = (tid >> 3) + u * 32;           // pixel index within the input tile
      const int r = p / IW, c = p - r * IW;
      const int hi = h0 - 1 + r, wi = w0 - 1 + c;
      ok[u] = p < NPIX && cv < a.CV && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
      const int hic = min(max(hi, 0), a.H - 1), wic = min(max(wi, 0), a.W - 1);
      const int cvc = min(cv, a.CV - 1);
      raw[u] = *(const uint4*)(a.x + (((long)n * a.H + hic) * a.W + wic) * a.C + cvc * 8);
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int p = (tid >> 3) + u * 32;
      if (p < NPIX) {
        const int r = p / IW, c = p - r * IW;
        float f[8]; unpack(raw[u], f);
        if (a.mode & 2) for (int i = 0; i < 8; ++i) f[i] = fmaf(f[i], sc[i], sh[i]);
        if (a.mode & 1) for (int i = 0; i < 8; ++i) f[i] = fmaxf(f[i], 0.f);
        uint4 v = pack(f);
        if (!ok[u]) v = make_uint4(0, 0, 0, 0);
        tile[(r * IWP + c) * CVB + cx] = v;
      }
    }
  }
  __syncthreads();
  // ---- compute: thread = (cx, row 0..7, strip 0..3); 4 outputs along W
  const int cx = tid & 7, row = (tid >> 3) & 7, strip = tid >> 6;
  const int cv = cvb0 + cx;
  float acc[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
#pragma unroll 1
  for (int kh = 0; kh < 3; ++kh) {
    float wv[3][8];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float4 wa = wsm[(kh * 3 + kw) * CVB * 2 + cx * 2], wb = wsm[(kh * 3 + kw) * CVB * 2 + cx * 2 + 1];
      wv[kw][0] = wa.x; wv[kw][1] = wa.y; wv[kw][2] = wa.z; wv[kw][3] = wa.w;
      wv[kw][4] = wb.x; wv[kw][5] = wb.y; wv[kw][6] = wb.z; wv[kw][7] = wb.w;
    }
    const uint4* trow = tile + ((row + kh) * IWP + strip * 4) * CVB + cx;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      float v[8];
      unpack(trow[q * CVB], v);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int j = q - kw;
        if (j >= 0 && j < 4) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(v[i], wv[kw][i], acc[j][i]);
        }
      }
    }
  }
  const int ho = h0 + row;
  float ssum[8], ssq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ssum[i] = ssq[i] = 0.f;
  if (cv < a.CV && ho < a.H) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int wo = w0 + strip * 4 + j;
      if (wo < a.W) {
        *(uint4*)(a.y + (((long)n * a.H + ho) * a.W + wo) * a.C + cv * 8) = pack(acc[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) { ssum[i] += acc[j][i]; ssq[i] += acc[j][i] * acc[j][i]; }
      }
    }
  }
  if (a.stat) {
    // reduce over the 32 pixel-threads of each cx: reuse the tile LDS as float[32][8 cx][16]
    __syncthreads();
    float* red = reinterpret_cast<float*>(tile);
    float* mine = red + ((tid >> 3) * CVB + cx) * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) { mine[i] = ssum[i]; mine[8 + i] = ssq[i]; }
    __syncthreads();
    if (tid < CVB * 16) {
      float tot = 0.f;
      for (int r = 0; r < 32; ++r) tot += red[r * CVB * 16 + tid];
      const int lcx = tid / 16, k = tid % 16, which = k / 8, ci = k % 8;
      const int c = (cvb0 + lcx) * 8 + ci;
      if (c < a.C) a.stat[((long)blockIdx.y * 2 + which) * a.C + c] = tot;
    }
  }
}

__global__ void fill_rand(uint32_t* p, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 13; h *= 0x5bd1e995u;
    uint32_t lo = 0x3f00u | (h & 0x80ffu), hi = 0x3f00u | ((h >> 16) & 0x80ffu);
    p[i] = lo | (hi << 16); }
}

int main() {
  const int shapes[3][4] = {{2, 65, 129, 728}, {2, 257, 513, 128}, {2, 513, 1025, 128}};
  for (int si = 0; si < 3; ++si) {
    const int N = shapes[si][0], H = shapes[si][1], W = shapes[si][2], C = shapes[si][3], CV = C / 8;
    const long n = (long)N * H * W * CV;
    uint4 *x, *y, *y2; float *w, *sc, *sh, *stat, *stat2;
    CK(hipMalloc(&x, n * 16)); CK(hipMalloc(&y, n * 16)); CK(hipMalloc(&y2, n * 16));
    CK(hipMalloc(&w, 9 * C * 4)); CK(hipMalloc(&sc, C * 4)); CK(hipMalloc(&sh, C * 4));
    hipLaunchKernelGGL(fill_rand, dim3((n * 4 + 255) / 256), dim3(256), 0, 0, (uint32_t*)x, n * 4);
    std::vector<float> hw(9 * C), ones(C, 1.25f), zer(C, 0.1f);
    for (int i = 0; i < 9 * C; ++i) hw[i] = 0.1f * (float)((i * 7919) % 13 - 6);
    CK(hipMemcpy(w, hw.data(), 9 * C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sc, ones.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sh, zer.data(), C * 4, hipMemcpyHostToDevice));
    const int tiles_h = (H + TH - 1) / TH, tiles_w = (W + TW - 1) / TW;
    const int ntiles = N * tiles_h * tiles_w;
    CK(hipMalloc(&stat, (size_t)ntiles * 2 * C * 4));
    const int gy = seg_dwconv_grid_y(1, C, N, H, W, 1, 1, 0);
    CK(hipMalloc(&stat2, (size_t)gy * 2 * C * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto fn) {
      for (int i = 0; i < 3; ++i) fn();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 30; ++i) fn();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("  %-34s %7.2f us  %6.0f GB/s\n", name, ms * 1e3 / 30, 2.0 * n * 16 / (ms * 1e-3 / 30) / 1e9);
    };
    printf("shape N%d %dx%d C%d  (%.1f MB)\n", N, H, W, C, n * 16 / 1e6);
    for (int mode = 1; mode <= 3; mode += 2) {
      Args a{(const uint16_t*)x, w, (uint16_t*)y, sc, sh, stat, N, H, W, C, CV, mode, tiles_h, tiles_w};
      char nm[64];
      snprintf(nm, 64, "production mode %d + stats", mode);
      timeit(nm, [&] { if (seg_dwconv3x3(1, 0, x, C, N, H, W, C, w, 0, 1, 1, mode, sc, sh, y2, C, H, W, stat2, gy, nullptr)) printf("ERR %s\n", seg_last_error()); });
      snprintf(nm, 64, "lds-tiled  mode %d + stats", mode);
      timeit(nm, [&] { hipLaunchKernelGGL(dw_lds, dim3((CV + CVB - 1) / CVB, ntiles), dim3(256), 0, 0, a); });
      a.stat = nullptr;
      snprintf(nm, 64, "lds-tiled  mode %d no stats", mode);
      timeit(nm, [&] { hipLaunchKernelGGL(dw_lds, dim3((CV + CVB - 1) / CVB, ntiles), dim3(256), 0, 0, a); });
    }
    // correctness (mode 3): compare y (lds) with y2 (production)
    {
      Args a{(const uint16_t*)x, w, (uint16_t*)y, sc, sh, nullptr, N, H, W, C, CV, 3, tiles_h, tiles_w};
      hipLaunchKernelGGL(dw_lds, dim3((CV + CVB - 1) / CVB, ntiles), dim3(256), 0, 0, a);
      seg_dwconv3x3(1, 0, x, C, N, H, W, C, w, 0, 1, 1, 3, sc, sh, y2, C, H, W, nullptr, gy, nullptr);
      std::vector<uint16_t> ha(n * 8), hb(n * 8);
      CK(hipMemcpy(ha.data(), y, n * 16, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb.data(), y2, n * 16, hipMemcpyDeviceToHost));
      double md = 0, mx = 0;
      for (long i = 0; i < n * 8; ++i) {
        union { uint32_t u; float f; } p, q; p.u = (uint32_t)ha[i] << 16; q.u = (uint32_t)hb[i] << 16;
        md = fmax(md, fabs((double)p.f - q.f)); mx = fmax(mx, fabs((double)q.f));
      }
      printf("  max |lds - production| = %.4g (max |y| %.3g)\n", md, mx);
    }
    hipFree(x); hipFree(y); hipFree(y2); hipFree(stat); hipFree(stat2);
  }
  return 0;
}
