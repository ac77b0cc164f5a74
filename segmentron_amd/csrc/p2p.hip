// SyncBatchNorm statistics exchange as ONE-HOP peer writes over xGMI (the 2C+1
// float64 sums of tools/train.py:76's SyncBatchNorm, 584 exchanges per DeepLabv3+/xception65
// train step).  A ring all-reduce of a 12-33 KB message pays 2 (W - 1) hops of latency; MI355X's
// xGMI is a full point-to-point mesh, so every rank can instead WRITE its vector straight into
// a mailbox slot on each peer (hipIpc-mapped, uncached device memory; every access a
// cache-bypassing system-scope atomic — ordering rules in p2p.h), raise a flag there, wait
// for the W flags in its own mailbox and add the W slots in rank order — one hop, one launch,
// and every rank adds the same numbers in the same order (bit-identical replicas).
//
//   mailbox (one allocation per rank, exported with hipIpcGetMemHandle):
//     [2 parities][W sender slots][slot_bytes]   data
//     [W][16] u64                                 flags: exchange number last published by sender r
//   exchange k uses parity k & 1.  A rank can reach exchange k + 2 (same parity) only after every
//   peer has published k + 1, i.e. after that peer's kernel of exchange k — the reader of the
//   slot — has ended: two parities are enough, no acknowledgements travel back.
//
// The exchange number lives in device memory and is advanced by the kernel itself, so the launch
// can be captured into a HIP graph and replayed.  Waiting is bounded (600 s of the constant
// 100 MHz clock by default, seg_p2p_set_timeout): on a timeout the kernel sets the error word,
// returns NaN from this and every later exchange (p2p.h p2p_failed), stops waiting in every
// later exchange and the host reports it (seg_p2p_status) — never a hang.
#include "p2p.h"
#include <cstring>

namespace seg {

constexpr int P2P_THREADS = 1024;

struct P2PArgs {
  unsigned char* peer[P2P_MAX_WORLD];
  void* buf;
  unsigned long long* seq;
  int* err;
  unsigned long long timeout_ticks;
  long slot_bytes;
  int n, rank, world;
};

template <typename T> struct P2PBits;
template <> struct P2PBits<double> {
  typedef unsigned long long U;
  __device__ static __forceinline__ double val(U b) { return __longlong_as_double((long long)b); }
};
template <> struct P2PBits<float> {
  typedef unsigned int U;
  __device__ static __forceinline__ float val(U b) { return __uint_as_float(b); }
};

template <typename T>
__global__ __launch_bounds__(P2P_THREADS) void p2p_allreduce_kernel(const P2PArgs a) {
  typedef typename P2PBits<T>::U U;
  __shared__ unsigned long long s_seq;
  __shared__ int s_bad;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_seq = *a.seq + 1;
    s_bad = *a.err;
  }
  __syncthreads();
  const unsigned long long seq = s_seq;
  const int par = (int)(seq & 1);
  // 1. my vector into slot [par][rank] of every mailbox (my own included)
  for (int r = 0; r < a.world; ++r) {
    U* dst = reinterpret_cast<U*>(a.peer[r] + ((long)par * a.world + a.rank) * a.slot_bytes);
    const U* src = reinterpret_cast<const U*>(a.buf);
    for (int i = tid; i < a.n; i += P2P_THREADS)
      __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  p2p_stores_done();  // this thread's slot writes are acknowledged before any flag below (p2p.h)
  __syncthreads();
  // 2. publish the exchange number on every rank; 3. wait for every rank's number here
  if (tid < a.world) {
    __hip_atomic_store(p2p_flags(a.peer[tid], a.world, a.slot_bytes) + (long)a.rank * P2P_FLAG_STRIDE,
                       seq, P2P_FLAG_STORE_ORDER, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long* mine =
        p2p_flags(a.peer[a.rank], a.world, a.slot_bytes) + (long)tid * P2P_FLAG_STRIDE;
    if (!s_bad) p2p_wait(mine, seq, a.err, a.timeout_ticks);
  }
  __syncthreads();
  // 4. the sum over ranks, in rank order (NaN if a wait timed out, now or earlier: p2p.h)
  T* out = reinterpret_cast<T*>(a.buf);
  const bool failed = p2p_failed(a.err);
  for (int i = tid; i < a.n; i += P2P_THREADS) {
    if (failed) {
      out[i] = (T)__longlong_as_double(0x7ff8000000000000ll);
      continue;
    }
    T s = (T)0;
    for (int r = 0; r < a.world; ++r) {
      const U* slot = reinterpret_cast<const U*>(
          a.peer[a.rank] + ((long)par * a.world + r) * a.slot_bytes);
      s += P2PBits<T>::val(__hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
    out[i] = s;
  }
  if (tid == 0) *a.seq = seq;
}

}  // namespace seg

#define P2P_HIP(call, what)                                                        \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      seg::set_error("%s: %s", what, hipGetErrorString(e_));                       \
      return 2;                                                                    \
    }                                                                              \
  } while (0)

extern "C" int seg_p2p_create(int rank, int world, long slot_bytes, void** handle_out) {
  using namespace seg;
  SEG_REQUIRE(world >= 1 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world,
              "p2p_create: rank %d / world %d (at most %d ranks)", rank, world, P2P_MAX_WORLD);
  SEG_REQUIRE(slot_bytes > 0 && slot_bytes % 128 == 0, "p2p_create: slot_bytes must be a multiple of 128");
  P2PState* s = new P2PState();
  s->rank = rank; s->world = world; s->slot_bytes = slot_bytes;
  s->timeout_ticks = P2P_TIMEOUT_TICKS;
  for (int r = 0; r < P2P_MAX_WORLD; ++r) { s->peer[r] = nullptr; s->opened[r] = false; }
  const long bytes = p2p_box_bytes(world, slot_bytes);
  void* box = nullptr;
  // uncached device memory: a peer's xGMI writes land in HBM behind this GPU's L2, so the
  // owner must not read the mailbox through cached lines
  hipError_t e = hipExtMallocWithFlags(&box, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&box, bytes, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) {
    delete s;
    set_error("p2p_create: no uncached / fine-grained device memory (%s)", hipGetErrorString(e));
    return 2;
  }
  void* words = nullptr;
  if (hipMalloc(&words, 128) != hipSuccess || hipMemset(box, 0, bytes) != hipSuccess ||
      hipMemset(words, 0, 128) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(box);
    if (words) (void)hipFree(words);
    delete s;
    set_error("p2p_create: cannot initialise the mailbox");
    return 2;
  }
  s->local = static_cast<unsigned char*>(box);
  s->peer[rank] = s->local;
  s->seq = static_cast<unsigned long long*>(words);
  s->arrive = reinterpret_cast<unsigned int*>(static_cast<unsigned char*>(words) + 32);
  s->err = reinterpret_cast<int*>(static_cast<unsigned char*>(words) + 64);
  *handle_out = s;
  return 0;
}

extern "C" int seg_p2p_ipc_handle(void* handle, void* out64) {
  using namespace seg;
  P2PState* s = static_cast<P2PState*>(handle);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  hipIpcMemHandle_t h;
  P2P_HIP(hipIpcGetMemHandle(&h, s->local), "p2p_ipc_handle: hipIpcGetMemHandle");
  memcpy(out64, &h, 64);
  return 0;
}

extern "C" int seg_p2p_connect(void* handle, const void* handles) {
  using namespace seg;
  P2PState* s = static_cast<P2PState*>(handle);
  for (int r = 0; r < s->world; ++r) {
    if (r == s->rank || s->peer[r] != nullptr) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, static_cast<const unsigned char*>(handles) + 64L * r, 64);
    void* p = nullptr;
    P2P_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess),
            "p2p_connect: hipIpcOpenMemHandle");
    s->peer[r] = static_cast<unsigned char*>(p);
    s->opened[r] = true;
  }
  return 0;
}

template <typename T>
static int p2p_all_reduce(void* handle, void* buf, int n, void* stream) {
  using namespace seg;
  P2PState* s = static_cast<P2PState*>(handle);
  SEG_REQUIRE(n > 0 && (long)n * (long)sizeof(T) <= s->slot_bytes,
              "p2p_all_reduce: %d elements exceed the %ld-byte slot", n, s->slot_bytes);
  P2PArgs a;
  for (int r = 0; r < P2P_MAX_WORLD; ++r) a.peer[r] = r < s->world ? s->peer[r] : nullptr;
  for (int r = 0; r < s->world; ++r)
    SEG_REQUIRE(a.peer[r] != nullptr, "p2p_all_reduce: rank %d is not connected", r);
  a.buf = buf;
  a.seq = s->seq; a.err = s->err; a.timeout_ticks = s->timeout_ticks;
  a.slot_bytes = s->slot_bytes;
  a.n = n; a.rank = s->rank; a.world = s->world;
  hipLaunchKernelGGL(p2p_allreduce_kernel<T>, dim3(1), dim3(P2P_THREADS), 0, (hipStream_t)stream, a);
  return check_launch("p2p_all_reduce");
}

extern "C" int seg_p2p_all_reduce_f64(void* handle, void* buf, int n, void* stream) {
  return p2p_all_reduce<double>(handle, buf, n, stream);
}

extern "C" int seg_p2p_all_reduce_f32(void* handle, void* buf, int n, void* stream) {
  return p2p_all_reduce<float>(handle, buf, n, stream);
}

// Synchronises the device; 0 = every exchange so far completed, 3 = a wait timed out.
extern "C" int seg_p2p_status(void* handle) {
  using namespace seg;
  P2PState* s = static_cast<P2PState*>(handle);
  int err = 0;
  P2P_HIP(hipDeviceSynchronize(), "p2p_status: synchronize");
  P2P_HIP(hipMemcpy(&err, s->err, sizeof(int), hipMemcpyDeviceToHost), "p2p_status: read");
  if (err != 0) {
    set_error("p2p: a peer did not publish its statistics within %.0f s; every exchange since "
              "then returned NaN (the error word stays set: rebuild the mailbox to continue)",
              (double)s->timeout_ticks / (double)P2P_TICKS_PER_S);
    return 3;
  }
  return 0;
}

// Bound of one in-kernel wait (default 600 s).  Takes effect for launches issued afterwards; a
// captured graph keeps the value it was captured with.
extern "C" int seg_p2p_set_timeout(void* handle, double seconds) {
  using namespace seg;
  P2PState* s = static_cast<P2PState*>(handle);
  SEG_REQUIRE(s != nullptr && seconds > 0.0 && seconds < 1e6, "p2p_set_timeout: %g s", seconds);
  s->timeout_ticks = (unsigned long long)(seconds * (double)P2P_TICKS_PER_S);
  return 0;
}

extern "C" int seg_p2p_destroy(void* handle) {
  using namespace seg;
  P2PState* s = static_cast<P2PState*>(handle);
  if (s == nullptr) return 0;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < s->world; ++r)
    if (s->opened[r]) (void)hipIpcCloseMemHandle(s->peer[r]);
  (void)hipFree(s->local);
  (void)hipFree(s->seq);
  delete s;
  return 0;
}
