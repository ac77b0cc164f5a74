// Stand-alone lab (r06): how fast can one CU fill LDS from L2 — by LDS-DMA (global_load_lds, 16 B per
// lane), through registers (global_load_dwordx4 + ds_write_b128), or half and half?  The direct-to-LDS
// weight-gradient GEMM spends 1170 of its ~1500 cycles per 64-pixel slot on a 32 KB fill even when every
// DMA reads ONE cached zero word (profiles/r06_wgrad_ab.md): is that the LDS-DMA path, or LDS itself?
//
//   256 blocks x 512 threads, 4 x 32 KB ring in LDS; per iteration a block moves 32 KB (64 B per thread)
//   from its own L2-resident 64 KB region; one barrier per iteration, 3 slots in flight; no consumer.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/fill_lab tools/lab/fill_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) unsigned char lds_t;
typedef __attribute__((address_space(1))) const unsigned char glb_t;

template <int MODE>  // 0: DMA x4, 1: registers x4, 2: 2 DMA + 2 registers
__global__ __launch_bounds__(512) void k_fill(const unsigned char* __restrict__ src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_t* lds = (lds_t*)smem;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const unsigned char* base = src + (size_t)blockIdx.x * (64 * 1024);  // 2 MB per XCD: L2-resident
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    const unsigned char* g = base + (size_t)(it & 1) * 32768;
    lds_t* slot = lds + (it & 3) * 32768;
    uint4 r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = wave * 4 + j;  // 32 pieces of 1 KB
      const bool dma = MODE == 0 || (MODE == 2 && j < 2);
      if (dma) __builtin_amdgcn_global_load_lds((glb_t*)(g + piece * 1024 + lane * 16), slot + piece * 1024, 16, 0, 0);
      else r[j] = *reinterpret_cast<const uint4*>(g + piece * 1024 + lane * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = wave * 4 + j;
      const bool dma = MODE == 0 || (MODE == 2 && j < 2);
      if (!dma) *reinterpret_cast<uint4*>(smem + (it & 3) * 32768 + piece * 1024 + lane * 16) = r[j];
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // two older iterations may still be in flight
    __syncthreads();
    if ((it & 7) == 7) {  // a token consumer so that nothing is optimised away
      const uint4 v = *reinterpret_cast<const uint4*>(smem + ((it + 1) & 3) * 32768 + tid * 16);
      acc.x ^= v.x; acc.y ^= v.y;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc.x == 0x12345678u) out[blockIdx.x] = (float)acc.y;
}

template <int MODE> static void run(const unsigned char* src, float* out, int iters, const char* name) {
  CK(hipFuncSetAttribute((const void*)k_fill<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_fill<MODE>, dim3(256), dim3(512), 128 * 1024, 0, src, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k_fill<MODE>, dim3(256), dim3(512), 128 * 1024, 0, src, out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / 5, per_slot_ns = us * 1e3 / iters;
  printf("%-28s %8.1f us / launch  %7.1f ns per 32 KB slot = %5.1f GB/s per CU = %5.2f TB/s chip\n", name, us,
         per_slot_ns, 32768.0 / per_slot_ns, 32768.0 / per_slot_ns * 256 * 1e-3);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  unsigned char* src; float* out;
  CK(hipMalloc(&src, (size_t)256 * 512 * 1024)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(src, 1, (size_t)256 * 512 * 1024));
  run<0>(src, out, iters, "LDS-DMA x4");
  run<1>(src, out, iters, "registers + ds_write x4");
  run<2>(src, out, iters, "2 LDS-DMA + 2 registers");
  run<0>(src, out, iters, "LDS-DMA x4 (again)");
  return 0;
}
