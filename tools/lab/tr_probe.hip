// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): which LDS element lands in which
// lane / 16-bit slot for a given per-lane address pattern.  LDS holds u16 value = its own
// element index (element = 2 bytes).
//   pattern 0: lane address = lane * 8 bytes (4 consecutive elements per lane)
//   pattern 1: a [k][128-element row] image: lane l -> row (l & 15) >> 2 ... see code
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) unsigned short lds_u16;
__global__ void probe(unsigned short* out, int pattern) {
  __shared__ unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned elem;
  if (pattern == 0) elem = l * 4;
  else if (pattern == 1) elem = (l >> 2) * 128 + (l & 3) * 4;   // 16 rows x 16 cols, row pitch 128
  else if (pattern == 2) elem = (l & 15) * 128 + (l >> 4) * 4;  // row = l & 15, col4 = l >> 4
  else elem = ((l & 15) >> 2) * 128 + (l & 3) * 4 + (l >> 4) * 16;  // 4 rows; groups step 16 cols
  const unsigned addr = (unsigned)(unsigned long)(lds_u16*)lds + elem * 2;
  unsigned long long r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[(pattern * 64 + l) * 4 + j] = (unsigned short)(r >> (16 * j));
}
int main() {
  unsigned short* d;
  hipMalloc(&d, 4 * 64 * 4 * 2);
  for (int p = 0; p < 4; ++p) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
  std::vector<unsigned short> h(4 * 64 * 4);
  hipMemcpy(h.data(), d, h.size() * 2, hipMemcpyDeviceToHost);
  for (int p = 0; p < 4; ++p) {
    printf("pattern %d: lane -> 4 element indices (row = idx / 128, col = idx %% 128 for patterns 1-3)\n", p);
    for (int l = 0; l < 64; ++l) {
      printf("  l%02d:", l);
      for (int j = 0; j < 4; ++j) {
        const int v = h[(p * 64 + l) * 4 + j];
        if (p == 0) printf(" %4d", v); else printf(" (%2d,%3d)", v / 128, v % 128);
      }
      if (l % 2 == 1) printf("\n");
    }
  }
  return 0;
}
