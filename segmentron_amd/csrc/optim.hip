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


// ---- multi-tensor weight packing: the fp32 master weights of the 1x1 convolutions whose input
// BatchNorm is NOT folded (fold.hip packs the folded ones) become compute-dtype GEMM operands
// once per optimizer step — `[O][C]` as stored (forward / weight-gradient operand) and `[C][O]`
// (data-gradient operand).  torch did this with one cast / transposing-copy launch per tensor and
// direction (~40 launches of 5-9 us per DeepLabv3+ step); here up to 32 (tensor, direction) jobs
// share one launch: a block owns four consecutive 64x64 tiles of one job, the job table travels in the kernel
// argument block (as in sgd_multi_tensor above).  Same round-to-nearest-even as torch's `.to()`.
namespace seg {

constexpr int PK_JOBS = 32, PK_BLOCKS = 480, PK_TILE = 64, PK_THREADS = 256;
constexpr int PK_SPAN = 4;  // consecutive 64x64 tiles per block (the argument block stays < 4 KB)

struct PackArgs {
  const float* src[PK_JOBS];
  void* dst[PK_JOBS];
  int O[PK_JOBS], C[PK_JOBS];         // source is [O][C] fp32, contiguous
  unsigned char transpose[PK_JOBS];   // destination [C][O] instead of [O][C]
  unsigned char blk_job[PK_BLOCKS];
  int blk_tile[PK_BLOCKS];            // first tile (row-major over the job's tile grid)
  int dtype;
};

template <typename T>
__global__ __launch_bounds__(PK_THREADS) void pack_multi_kernel(const PackArgs a) {
  __shared__ float tile[PK_TILE][PK_TILE + 1];
  const int j = a.blk_job[blockIdx.x];
  const int O = a.O[j], C = a.C[j];
  const int tc = (C + PK_TILE - 1) / PK_TILE, ntiles = ((O + PK_TILE - 1) / PK_TILE) * tc;
  const float* __restrict__ src = a.src[j];
  T* __restrict__ dst = reinterpret_cast<T*>(a.dst[j]);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  const bool tr = a.transpose[j] != 0;
  for (int t = a.blk_tile[blockIdx.x]; t < min(ntiles, a.blk_tile[blockIdx.x] + PK_SPAN); ++t) {
    const int o0 = (t / tc) * PK_TILE, c0 = (t % tc) * PK_TILE;
    if (!tr) {
      for (int r = ty; r < PK_TILE; r += 4) {
        const int o = o0 + r, c = c0 + tx;
        if (o < O && c < C) Vec<T>::store1(dst + (long)o * C + c, src[(long)o * C + c]);
      }
      continue;
    }
    __syncthreads();  // the previous tile has been read out
    for (int r = ty; r < PK_TILE; r += 4) {
      const int o = o0 + r, c = c0 + tx;
      tile[r][tx] = (o < O && c < C) ? src[(long)o * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < PK_TILE; r += 4) {
      const int c = c0 + r, o = o0 + tx;
      if (c < C && o < O) Vec<T>::store1(dst + (long)c * O + o, tile[tx][r]);
    }
  }
}

}  // namespace seg

// srcs / dsts: HOST arrays of `njobs` device pointers; O / C / transpose: host arrays.
extern "C" int seg_pack_multi(int dtype, int njobs, const void* const* srcs, const void* const* dsts,
                              const int* O, const int* C, const int* transpose, void* stream) {
  using namespace seg;
  SEG_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "pack_multi: bad dtype %d", dtype);
  PackArgs a;
  a.dtype = dtype;
  int nj = 0, nb = 0;
  auto flush = [&]() -> int {
    if (nb == 0) { nj = 0; return 0; }
    if (dtype == DT_BF16)
      hipLaunchKernelGGL((pack_multi_kernel<bf16_t>), dim3(nb), dim3(PK_THREADS), 0,
                         (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((pack_multi_kernel<float>), dim3(nb), dim3(PK_THREADS), 0,
                         (hipStream_t)stream, a);
    nj = nb = 0;
    return check_launch("pack_multi");
  };
  for (int i = 0; i < njobs; ++i) {
    SEG_REQUIRE(srcs[i] && dsts[i] && O[i] >= 1 && C[i] >= 1, "pack_multi: bad job %d", i);
    const long to = (O[i] + PK_TILE - 1) / PK_TILE, tc = (C[i] + PK_TILE - 1) / PK_TILE;
    SEG_REQUIRE(to * tc < (1L << 30), "pack_multi: job %d too large", i);
    long t = 0;
    while (t < to * tc) {
      if (nj == PK_JOBS || nb == PK_BLOCKS) {
        const int rc = flush();
        if (rc) return rc;
      }
      a.src[nj] = (const float*)srcs[i]; a.dst[nj] = const_cast<void*>(dsts[i]);
      a.O[nj] = O[i]; a.C[nj] = C[i]; a.transpose[nj] = transpose[i] ? 1 : 0;
      while (t < to * tc && nb < PK_BLOCKS) {
        a.blk_job[nb] = (unsigned char)nj;
        a.blk_tile[nb] = (int)t;
        ++nb; t += PK_SPAN;
      }
      ++nj;
    }
  }
  return flush();
}
