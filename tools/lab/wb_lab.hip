// Stand-alone lab (r06): does the store FLAVOUR of a chip-filling kernel's output change what the
// kernel boundary behind it costs?  (MI355X_MICROARCH.md "boundary" row: + B / 6 TB/s when the
// predecessor leaves B bytes dirty in the XCD L2s — a 24 MB activation fits the 32 MB of L2.)
//
//   A  streaming element-wise kernel: reads 24 MB, writes 24 MB  (bn_bwd_apply-like)
//   B  "GEMM-like": 198 blocks x 512 threads spin T, then each writes its 256 x 256 bf16 tile
//      (128 KB) in one burst at the end (every output byte is written in the last microseconds)
// each followed by a one-wave-per-channel-block reduction kernel (91 blocks) that depends on it,
// N pairs captured into one HIP graph; store flavours: plain, nt, sc1, sc0 sc1.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/wb_lab tools/lab/wb_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE> __device__ __forceinline__ void st16(void* p, u32x4 v) {
  if (MODE == 0) *reinterpret_cast<u32x4*>(p) = v;
  else if (MODE == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
  else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void k_stream(const u32x4* __restrict__ in, u32x4* __restrict__ out, long n) {
  const long i0 = (long)blockIdx.x * 1024 + threadIdx.x;
  u32x4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { const long i = i0 + u * 256; v[u] = i < n ? in[i] : u32x4{0, 0, 0, 0}; }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long i = i0 + u * 256;
    if (i < n) { v[u].x += 1; st16<MODE>(out + i, v[u]); }
  }
}

// bn_bwd_apply-like: two reads + one write (73 MB on the 24.4 MB tensors), flat mapping
__global__ __launch_bounds__(256) void k_stream2(const u32x4* __restrict__ g, const u32x4* __restrict__ x,
                                                 u32x4* __restrict__ out, long n) {
  const long i0 = (long)blockIdx.x * 1024 + threadIdx.x;
  u32x4 a[4], b[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long i = i0 + u * 256;
    a[u] = i < n ? g[i] : u32x4{0, 0, 0, 0};
    b[u] = i < n ? x[i] : u32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long i = i0 + u * 256;
    if (i < n) { a[u].x += b[u].y; a[u].z ^= b[u].w; out[i] = a[u]; }
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void k_gemmlike(const float* __restrict__ dep, u32x4* __restrict__ out,
                                                  long cycles) {
  extern __shared__ float sm[];
  const long t0 = clock64();
  float v = dep[threadIdx.x & 63];
  sm[threadIdx.x] = v;
  while (clock64() - t0 < cycles) v = fmaf(v, 1.0001f, 0.5f);
  u32x4 w = {__float_as_uint(v), 1u, 2u, (unsigned)sm[(threadIdx.x + 1) & 511]};
  u32x4* o = out + (long)blockIdx.x * 8192;  // 128 KB per block
#pragma unroll 4
  for (int k = 0; k < 16; ++k) st16<MODE>(o + k * 512 + threadIdx.x, w);
}

__global__ __launch_bounds__(256) void k_small(const float* __restrict__ part, int rows, long pitch,
                                               int C, float* __restrict__ out) {
  const int c = blockIdx.x * 8 + (threadIdx.x & 7), rl = threadIdx.x >> 3;
  __shared__ float red[32][8];
  float s = 0.f;
  if (c < C)
    for (int r = rl; r < rows; r += 32) s += part[(long)r * pitch + c];
  red[rl][threadIdx.x & 7] = s;
  __syncthreads();
  if (threadIdx.x < 8 && c < C) {
    float t = 0;
    for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
    out[c] = t * 1e-3f;
  }
}

static float replay_us(hipGraphExec_t g, hipStream_t s, int reps, int N) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(g, s));
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(g, s));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps / N;
}
template <typename F> static hipGraphExec_t capture(hipStream_t s, F body) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  body();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  return ge;
}

template <int MODE> static void run(hipStream_t s, u32x4* a, u32x4* b, float* outv, long n, long cyc, int N) {
  const char* names[] = {"plain", "nt", "sc1", "sc0 sc1"};
  const int C = 728;
  const int nb = (int)((n + 1023) / 1024);
  // A: stream chain alone, ping-pong
  hipGraphExec_t g1 = capture(s, [&] {
    for (int i = 0; i < N; ++i)
      hipLaunchKernelGGL(k_stream<MODE>, dim3(nb), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n);
  });
  const float t1 = replay_us(g1, s, 10, N);
  hipGraphExec_t g2 = capture(s, [&] {
    for (int i = 0; i < N; ++i) {
      u32x4* o = (i & 1) ? a : b;
      hipLaunchKernelGGL(k_stream<MODE>, dim3(nb), dim3(256), 0, s, (i & 1) ? b : a, o, n);
      hipLaunchKernelGGL(k_small, dim3(91), dim3(256), 0, s, (const float*)o, 64, 728L * 2, C, outv + (i & 15) * C);
    }
  });
  const float t2 = replay_us(g2, s, 10, N);
  // B: gemm-like alone / + small
  hipGraphExec_t g3 = capture(s, [&] {
    for (int i = 0; i < N; ++i)
      hipLaunchKernelGGL(k_gemmlike<MODE>, dim3(198), dim3(512), 128 * 1024, s, outv, (i & 1) ? a : b, cyc);
  });
  const float t3 = replay_us(g3, s, 10, N);
  hipGraphExec_t g4 = capture(s, [&] {
    for (int i = 0; i < N; ++i) {
      u32x4* o = (i & 1) ? a : b;
      hipLaunchKernelGGL(k_gemmlike<MODE>, dim3(198), dim3(512), 128 * 1024, s, outv + (i & 15) * C, o, cyc);
      hipLaunchKernelGGL(k_small, dim3(91), dim3(256), 0, s, (const float*)o, 64, 728L * 2, C, outv + ((i + 1) & 15) * C);
    }
  });
  const float t4 = replay_us(g4, s, 10, N);
  // C: gemm-like -> stream (a big consumer that READS what the burst wrote) pairs
  hipGraphExec_t g5 = capture(s, [&] {
    for (int i = 0; i < N; ++i) {
      hipLaunchKernelGGL(k_gemmlike<MODE>, dim3(198), dim3(512), 128 * 1024, s, outv, a, cyc);
      hipLaunchKernelGGL(k_stream<MODE>, dim3(nb), dim3(256), 0, s, a, b, n);
    }
  });
  const float t5 = replay_us(g5, s, 10, N);
  printf("%-8s | stream %.2f  stream+small %.2f (+%.2f) | gemmlike %.2f  +small %.2f (+%.2f) | gemmlike+stream %.2f\n",
         names[MODE], t1, t2, t2 - t1, t3, t4, t4 - t3, t5);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 100;
  const long cyc = argc > 2 ? atol(argv[2]) : 40000;
  const long n = 16770L * 728 * 2 / 16;  // 24.4 MB of uint4
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  u32x4 *a, *b; float* outv;
  const size_t bytes = (size_t)256 * 128 * 1024 + n * 16;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&outv, 4 * 728 * 16));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(outv, 0, 4 * 728 * 16));
  CK(hipFuncSetAttribute((const void*)k_gemmlike<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  CK(hipFuncSetAttribute((const void*)k_gemmlike<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  CK(hipFuncSetAttribute((const void*)k_gemmlike<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  CK(hipFuncSetAttribute((const void*)k_gemmlike<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  printf("us per node-group, N = %d per graph, 24.4 MB tensors\n", N);
  {
    u32x4* c; CK(hipMalloc(&c, bytes)); CK(hipMemset(c, 0, bytes));
    const int nb = (int)((n + 1023) / 1024);
    hipGraphExec_t g2 = capture(s, [&] {
      for (int i = 0; i < N; ++i)
        hipLaunchKernelGGL(k_stream2, dim3(nb), dim3(256), 0, s, (i & 1) ? b : a, c, (i & 1) ? a : b, n);
    });
    const float t = replay_us(g2, s, 10, N);
    printf("stream2 (2 reads + 1 write, 73.2 MB): %.2f us = %.2f TB/s\n", t, 3 * n * 16 / t * 1e-6);
  }
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(s, a, b, outv, n, cyc, N);
    run<1>(s, a, b, outv, n, cyc, N);
    run<2>(s, a, b, outv, n, cyc, N);
    run<3>(s, a, b, outv, n, cyc, N);
  }
  return 0;
}
