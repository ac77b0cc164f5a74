// Stand-alone lab (r06): what a dependent launch costs inside a replayed HIP graph on this stack,
// and whether a captured SIDE BRANCH (fork by event, join at the end) lets a one-block reduction
// kernel run beside the next chip-filling kernel instead of behind it.
//
//   E1  chain of N trivial kernels (1 block / 256 blocks)                  -> us per node
//   E2  chain of N "finalize-like" kernels (91 blocks x 256 threads, 64 partial rows each)
//   E3  N x [big, small]: small depends on big; the next big does NOT depend on small
//         a) one stream (serial chain)      b) small on a forked stream, joined once at the end
//         c) small on a forked stream, joined before the NEXT-BUT-ONE big (bounded run-ahead)
//   E4  big alone x N (the floor of E3b)
// big = 198 blocks x 512 threads, 128 KiB of LDS (the dominant GEMM's footprint), spins T cycles.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/launch_lab tools/lab/launch_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_trivial(float* p, int i) {
  if (threadIdx.x == 0) p[blockIdx.x] = (float)i;
}

// 91 blocks x 256 threads: thread = (8 channels of one channel vector) x 32 row lanes
__global__ __launch_bounds__(256) void k_finalize(const float* __restrict__ part, int rows, int C,
                                                  float* __restrict__ out) {
  const int c = blockIdx.x * 8 + (threadIdx.x & 7), rl = threadIdx.x >> 3;
  __shared__ float red[32][8];
  float s = 0.f;
  if (c < C)
    for (int r = rl; r < rows; r += 32) s += part[(long)r * C + c];
  red[rl][threadIdx.x & 7] = s;
  __syncthreads();
  if (threadIdx.x < 8 && c < C) {
    double t = 0;
    for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
    out[c] = (float)(t * 1e-3);
  }
}

// chip-filling stand-in: every block spins `cycles` shader clocks, then writes one partial row
__global__ __launch_bounds__(512) void k_big(float* __restrict__ part, int C, long cycles,
                                             const float* __restrict__ dep) {
  extern __shared__ float sm[];
  const long t0 = clock64();
  float v = dep ? dep[threadIdx.x & 63] : 0.f;
  sm[threadIdx.x] = v;
  while (clock64() - t0 < cycles) v = fmaf(v, 1.0001f, 0.5f);
  for (int c = threadIdx.x; c < C; c += 512) part[(long)blockIdx.x * C + c] = v + sm[c & 511];
}

static float replay_ms(hipGraphExec_t g, hipStream_t s, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(g, s));
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(g, s));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <typename F> static hipGraphExec_t capture(hipStream_t s, F body) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  body();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  return ge;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 200;
  const long big_cycles = argc > 2 ? atol(argv[2]) : 2400L * 25;  // ~25 us at 2.4 GHz... clock64 = 100 MHz? measured below
  const int C = 728, ROWS = 198;
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  float *part, *out, *triv;
  CK(hipMalloc(&part, sizeof(float) * 256 * C * 4));
  CK(hipMalloc(&out, sizeof(float) * C * 64));
  CK(hipMalloc(&triv, sizeof(float) * 4096));
  CK(hipMemset(part, 0, sizeof(float) * 256 * C * 4));
  CK(hipMemset(out, 0, sizeof(float) * C * 64));
  CK(hipFuncSetAttribute((const void*)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));

  // calibrate clock64: one big launch alone
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_big, dim3(ROWS), dim3(512), 128 * 1024, s, part, C, big_cycles, nullptr);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_big alone (eager, events): %.1f us for %ld clock64 ticks\n", ms * 1e3, big_cycles);
  }
  const int reps = 20;
  for (int blocks : {1, 256}) {
    hipGraphExec_t g = capture(s, [&] {
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_trivial, dim3(blocks), dim3(64), 0, s, triv, i);
    });
    printf("E1 trivial chain, %3d blocks: %.2f us / node (graph replay)\n", blocks,
           replay_ms(g, s, reps) * 1e3 / N);
  }
  {
    // eager back-to-back for comparison (host-bound if the host is slower than the device)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < N * 5; ++i) hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, s, triv, i);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("E1 trivial chain, eager stream: %.2f us / launch\n", ms * 1e3 / (N * 5));
  }
  {
    hipGraphExec_t g = capture(s, [&] {
      for (int i = 0; i < N; ++i)
        hipLaunchKernelGGL(k_finalize, dim3(91), dim3(256), 0, s, part + (i & 1) * 64 * C, 64, C, out + (i & 31) * C);
    });
    printf("E2 finalize-like chain (91 blocks, 64 rows x 728): %.2f us / node\n", replay_ms(g, s, reps) * 1e3 / N);
  }
  {
    hipGraphExec_t g = capture(s, [&] {
      for (int i = 0; i < N; ++i)
        hipLaunchKernelGGL(k_big, dim3(ROWS), dim3(512), 128 * 1024, s, part + (i & 1) * 256 * C, C, big_cycles, nullptr);
    });
    printf("E4 big alone: %.2f us / node\n", replay_ms(g, s, reps) * 1e3 / N);
  }
  {
    hipGraphExec_t g = capture(s, [&] {
      for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(k_big, dim3(ROWS), dim3(512), 128 * 1024, s, part + (i & 1) * 256 * C, C, big_cycles, nullptr);
        hipLaunchKernelGGL(k_finalize, dim3(91), dim3(256), 0, s, part + (i & 1) * 256 * C, ROWS, C, out + (i & 31) * C);
      }
    });
    printf("E3a [big, small] serial: %.2f us / pair\n", replay_ms(g, s, reps) * 1e3 / N);
  }
  {
    // critical variant: the next big READS the small kernel's output (true dependency, one stream)
    hipGraphExec_t g = capture(s, [&] {
      for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(k_big, dim3(ROWS), dim3(512), 128 * 1024, s, part + (i & 1) * 256 * C, C, big_cycles, out + ((i + 31) & 31) * C);
        hipLaunchKernelGGL(k_finalize, dim3(91), dim3(256), 0, s, part + (i & 1) * 256 * C, ROWS, C, out + (i & 31) * C);
      }
    });
    printf("E3a' [big(dep on small), small] serial: %.2f us / pair\n", replay_ms(g, s, reps) * 1e3 / N);
  }
  {
    std::vector<hipEvent_t> ev(N + 1);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipGraphExec_t g = capture(s, [&] {
      for (int i = 0; i < N; ++i) {
        // 4 rotating partial buffers: small_i may still read buffer i while big_{i+1} writes i+1
        hipLaunchKernelGGL(k_big, dim3(ROWS), dim3(512), 128 * 1024, s, part + (i & 3) * 256 * C, C, big_cycles, nullptr);
        CK(hipEventRecord(ev[i], s));
        CK(hipStreamWaitEvent(s2, ev[i], 0));
        hipLaunchKernelGGL(k_finalize, dim3(91), dim3(256), 0, s2, part + (i & 3) * 256 * C, ROWS, C, out + (i & 31) * C);
      }
      CK(hipEventRecord(ev[N], s2));
      CK(hipStreamWaitEvent(s, ev[N], 0));
    });
    printf("E3b [big | small on a forked branch, one join at the end]: %.2f us / pair\n", replay_ms(g, s, reps) * 1e3 / N);
  }
  {
    std::vector<hipEvent_t> ev(2 * N + 2);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipGraphExec_t g = capture(s, [&] {
      for (int i = 0; i < N; ++i) {
        if (i >= 2) CK(hipStreamWaitEvent(s, ev[N + i - 2], 0));  // join small_{i-2} before big_i
        hipLaunchKernelGGL(k_big, dim3(ROWS), dim3(512), 128 * 1024, s, part + (i & 3) * 256 * C, C, big_cycles, nullptr);
        CK(hipEventRecord(ev[i], s));
        CK(hipStreamWaitEvent(s2, ev[i], 0));
        hipLaunchKernelGGL(k_finalize, dim3(91), dim3(256), 0, s2, part + (i & 3) * 256 * C, ROWS, C, out + (i & 31) * C);
        CK(hipEventRecord(ev[N + i], s2));
      }
      CK(hipEventRecord(ev[2 * N], s2));
      CK(hipStreamWaitEvent(s, ev[2 * N], 0));
    });
    printf("E3c [big | small forked, joined two bigs later]: %.2f us / pair\n", replay_ms(g, s, reps) * 1e3 / N);
  }
  {
    // E5: two smalls per big on the branch (the finalize + the fold of a separable conv)
    std::vector<hipEvent_t> ev(N + 1);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipGraphExec_t g = capture(s, [&] {
      for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(k_big, dim3(ROWS), dim3(512), 128 * 1024, s, part + (i & 3) * 256 * C, C, big_cycles, nullptr);
        CK(hipEventRecord(ev[i], s));
        CK(hipStreamWaitEvent(s2, ev[i], 0));
        hipLaunchKernelGGL(k_finalize, dim3(91), dim3(256), 0, s2, part + (i & 3) * 256 * C, ROWS, C, out + (i & 31) * C);
        hipLaunchKernelGGL(k_finalize, dim3(91), dim3(256), 0, s2, part + (i & 3) * 256 * C, ROWS, C, out + (32 + (i & 31)) * C);
      }
      CK(hipEventRecord(ev[N], s2));
      CK(hipStreamWaitEvent(s, ev[N], 0));
    });
    printf("E5 [big | two smalls forked]: %.2f us / triple\n", replay_ms(g, s, reps) * 1e3 / N);
    hipGraphExec_t g2 = capture(s, [&] {
      for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(k_big, dim3(ROWS), dim3(512), 128 * 1024, s, part + (i & 3) * 256 * C, C, big_cycles, nullptr);
        hipLaunchKernelGGL(k_finalize, dim3(91), dim3(256), 0, s, part + (i & 3) * 256 * C, ROWS, C, out + (i & 31) * C);
        hipLaunchKernelGGL(k_finalize, dim3(91), dim3(256), 0, s, part + (i & 3) * 256 * C, ROWS, C, out + (32 + (i & 31)) * C);
      }
    });
    printf("E5' [big, small, small] serial: %.2f us / triple\n", replay_ms(g2, s, reps) * 1e3 / N);
  }
  return 0;
}
