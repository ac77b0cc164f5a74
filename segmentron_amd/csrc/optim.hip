// Multi-tensor SGD step (momentum, weight decay; dampening 0, no Nesterov — what
// segmentron/solver/optimizer.py:45-50 builds with torch.optim.SGD):
//     d = g + wd * p ;  m = first ? d : momentum * m + d ;  p = p - lr * m
// DeepLabv3+/xception65 has 440 parameter tensors: one launch handles up to 48 of
// them — pointers, sizes and a (tensor, chunk) work list travel in the kernel argument block, as
// many launches as needed follow each other on the stream.  The learning rate is read from
// DEVICE memory (one float per parameter group), so a captured HIP graph of the step keeps
// following the schedule: the host only rewrites those floats between replays.
// Parameters, gradients and momentum buffers are fp32 (the bf16 compute path keeps fp32 master
// weights; the packed bf16 copies are re-made by the convolutions' weight cache).
#include "common.h"

namespace seg {

constexpr int SGD_TENSORS = 48;      // tensors per launch
constexpr int SGD_BLOCKS = 320;      // work-list entries (= blocks) per launch
constexpr int SGD_CHUNK = 16384;     // elements per block
constexpr int SGD_THREADS = 256;

struct SgdArgs {
  float* p[SGD_TENSORS];
  const float* g[SGD_TENSORS];
  float* m[SGD_TENSORS];
  int n[SGD_TENSORS];
  unsigned char group[SGD_TENSORS];
  unsigned char blk_tensor[SGD_BLOCKS];
  int blk_chunk[SGD_BLOCKS];
  const float* lr;       // [groups], device
  const float* wd;       // [groups], device
  float momentum;
  int first;
};

__global__ __launch_bounds__(SGD_THREADS) void sgd_multi_tensor_kernel(const SgdArgs a) {
  const int t = a.blk_tensor[blockIdx.x];
  const long base = (long)a.blk_chunk[blockIdx.x] * SGD_CHUNK;
  const int n = a.n[t];
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  const float lr = a.lr[a.group[t]], wd = a.wd[a.group[t]];
  const float mom = a.momentum;
  const bool vec = (n % 4) == 0;  // (torch allocations are >= 64-byte aligned)
  if (vec) {
    for (long i = base + threadIdx.x * 4; i < min((long)n, base + SGD_CHUNK); i += SGD_THREADS * 4) {
      float4 pv = *reinterpret_cast<float4*>(p + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      float4 mv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!a.first) mv = *reinterpret_cast<float4*>(m + i);
      float pe[4] = {pv.x, pv.y, pv.z, pv.w}, me[4] = {mv.x, mv.y, mv.z, mv.w};
      const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = fmaf(wd, pe[k], ge[k]);
        me[k] = a.first ? d : fmaf(mom, me[k], d);
        pe[k] = fmaf(-lr, me[k], pe[k]);
      }
      *reinterpret_cast<float4*>(m + i) = make_float4(me[0], me[1], me[2], me[3]);
      *reinterpret_cast<float4*>(p + i) = make_float4(pe[0], pe[1], pe[2], pe[3]);
    }
  } else {
    for (long i = base + threadIdx.x; i < min((long)n, base + SGD_CHUNK); i += SGD_THREADS) {
      const float d = fmaf(wd, p[i], g[i]);
      const float mv = a.first ? d : fmaf(mom, m[i], d);
      m[i] = mv;
      p[i] = fmaf(-lr, mv, p[i]);
    }
  }
}

}  // namespace seg

// params / grads / bufs: HOST arrays of `ntensors` device pointers (fp32, contiguous); numel and
// group: host arrays; lr_dev / wd_dev: device float arrays indexed by group.  first != 0: the
// momentum buffers are initialised with d (torch: buf = clone(d_p) on the first step).
extern "C" int seg_sgd_multi_tensor(int ntensors, const void* const* params,
                                    const void* const* grads, const void* const* bufs,
                                    const long* numel, const int* group, const float* lr_dev,
                                    const float* wd_dev, float momentum, int first, void* stream) {
  using namespace seg;
  SEG_REQUIRE(ntensors >= 0 && lr_dev && wd_dev, "sgd_multi_tensor: bad arguments");
  SgdArgs a;
  a.lr = lr_dev; a.wd = wd_dev; a.momentum = momentum; a.first = first ? 1 : 0;
  int nt = 0, nb = 0;
  auto flush = [&]() -> int {
    if (nb == 0) { nt = 0; return 0; }
    hipLaunchKernelGGL(sgd_multi_tensor_kernel, dim3(nb), dim3(SGD_THREADS), 0,
                       (hipStream_t)stream, a);
    nt = nb = 0;
    return check_launch("sgd_multi_tensor");
  };
  for (int i = 0; i < ntensors; ++i) {
    SEG_REQUIRE(params[i] && grads[i] && bufs[i], "sgd_multi_tensor: null tensor %d", i);
    SEG_REQUIRE(numel[i] >= 0 && numel[i] < (1L << 31), "sgd_multi_tensor: tensor %d too large", i);
    SEG_REQUIRE(group[i] >= 0 && group[i] < 256, "sgd_multi_tensor: bad group %d", group[i]);
    const int chunks = (int)((numel[i] + SGD_CHUNK - 1) / SGD_CHUNK);
    int c = 0;
    while (c < chunks) {
      if (nt == SGD_TENSORS || nb == SGD_BLOCKS) {
        const int rc = flush();
        if (rc) return rc;
      }
      // (re-)register the tensor in the current launch
      a.p[nt] = (float*)params[i]; a.g[nt] = (const float*)grads[i]; a.m[nt] = (float*)bufs[i];
      a.n[nt] = (int)numel[i]; a.group[nt] = (unsigned char)group[i];
      while (c < chunks && nb < SGD_BLOCKS) {
        a.blk_tensor[nb] = (unsigned char)nt;
        a.blk_chunk[nb] = c;
        ++nb; ++c;
      }
      ++nt;
    }
  }
  return flush();
}
