// Peer-mailbox exchange (csrc/p2p.hip) — the pieces other kernels need: the per-process state,
// the by-value device view, and the IN-KERNEL exchange used by the BatchNorm finalize kernels
// (bn.hip, fold.hip): with it a SyncBatchNorm costs the same ONE finalize launch per direction as
// a plain BatchNorm (column sums -> exchange -> finalize in one kernel) instead of three.
//
//   mailbox:  [2 parities][W senders][slot_bytes]   data
//             [W][16] u64                           flags of the single-block all-reduce kernel
//             [W][P2P_MAX_BLOCKS] u64               flags of the block-indexed exchange
// Both kinds of exchange draw their number from the same device counter (`seq`), so the parity
// argument of p2p.hip covers any interleaving of the two.
#pragma once
#include "common.h"

namespace seg {

constexpr int P2P_MAX_WORLD = 16;
constexpr int P2P_MAX_BLOCKS = 512;
constexpr int P2P_FLAG_STRIDE = 16;  // u64 per flag of the single-block kernel: one line each
// default bound of one wait: 600 s of the constant 100 MHz clock (NCCL's own watchdog default is
// minutes too: a rank that saves a checkpoint, or stalls in its data loader, must not trip it);
// seg_p2p_set_timeout changes it per mailbox
constexpr unsigned long long P2P_TICKS_PER_S = 100000000ull;
constexpr unsigned long long P2P_TIMEOUT_TICKS = 600ull * P2P_TICKS_PER_S;

struct P2PState {
  int rank, world;
  long slot_bytes;
  unsigned char* local;                 // this rank's mailbox
  unsigned char* peer[P2P_MAX_WORLD];   // every rank's mailbox as mapped here (peer[rank] = local)
  bool opened[P2P_MAX_WORLD];
  unsigned long long* seq;              // device: exchanges completed
  unsigned int* arrive;                 // device: blocks of the running exchange that are done
  int* err;                             // device: 0 ok, 1 timed out (latched: never reset)
  unsigned long long timeout_ticks;     // bound of one wait (seg_p2p_set_timeout)
};

// What a kernel gets (by value).  world == 0: no exchange (single-process BatchNorm).
struct P2PDev {
  unsigned char* peer[P2P_MAX_WORLD];
  unsigned long long* seq;
  unsigned int* arrive;
  int* err;
  unsigned long long timeout_ticks;
  long slot_bytes;
  int rank, world;
};

inline P2PDev p2p_dev_none() {
  P2PDev d;
  for (int r = 0; r < P2P_MAX_WORLD; ++r) d.peer[r] = nullptr;
  d.seq = nullptr; d.arrive = nullptr; d.err = nullptr;
  d.slot_bytes = 0; d.rank = 0; d.world = 0; d.timeout_ticks = P2P_TIMEOUT_TICKS;
  return d;
}

// false (+ error text) if a peer is not connected
inline bool p2p_dev_of(void* handle, P2PDev& d) {
  P2PState* s = static_cast<P2PState*>(handle);
  d = p2p_dev_none();
  if (s == nullptr) { set_error("p2p: null handle"); return false; }
  for (int r = 0; r < s->world; ++r) {
    if (s->peer[r] == nullptr) { set_error("p2p: rank %d is not connected", r); return false; }
    d.peer[r] = s->peer[r];
  }
  d.seq = s->seq; d.arrive = s->arrive; d.err = s->err;
  d.slot_bytes = s->slot_bytes; d.rank = s->rank; d.world = s->world;
  d.timeout_ticks = s->timeout_ticks;
  return true;
}

inline long p2p_box_bytes(int world, long slot_bytes) {
  return 2L * world * slot_bytes + (long)world * P2P_FLAG_STRIDE * 8 +
         (long)world * P2P_MAX_BLOCKS * 8;
}

__device__ __forceinline__ unsigned long long* p2p_flags(unsigned char* box, int world,
                                                         long slot_bytes) {
  return reinterpret_cast<unsigned long long*>(box + 2L * world * slot_bytes);
}
__device__ __forceinline__ unsigned long long* p2p_block_flags(unsigned char* box, int world,
                                                               long slot_bytes) {
  return p2p_flags(box, world, slot_bytes) + (long)world * P2P_FLAG_STRIDE;
}

// Ordering inside the protocol.  EVERY mailbox access (slots and flags) is a system-scope atomic
// on uncached memory: write-through / cache-bypassing (global_store / global_load sc0 sc1), so
// no cache maintenance is needed for them — only ORDER: a slot store must have been acknowledged
// before the flag store is issued (s_waitcnt vmcnt(0), then the block barrier), and the slot loads
// are issued after the flag load has returned (value dependence + barrier).  The by-the-book
// __threadfence_system() / release-acquire atomics would add an L2 write-back (buffer_wbl2) of
// whatever the train step left dirty and an invalidate per poll: measured r03, 91 blocks each
// doing that turned a 5 us finalize kernel into an 18 us one.
//
// -DSEG_P2P_STRICT (make P2P_STRICT=1) builds the by-the-book protocol instead: system-scope
// release fence before a RELEASE flag store, ACQUIRE flag loads + system-scope acquire fence
// behind the wait.  Same mailbox layout, same results on one device; kept so that the day a
// multi-GPU node exists the fast and the safe ordering can be A/B'd from one source tree
// (SEGMENTRON_HIP_LIB selects the build).
#ifdef SEG_P2P_STRICT
#define P2P_FLAG_STORE_ORDER __ATOMIC_RELEASE
#define P2P_FLAG_LOAD_ORDER __ATOMIC_ACQUIRE
__device__ __forceinline__ void p2p_stores_done() { __threadfence_system(); }
__device__ __forceinline__ void p2p_waited() { __threadfence_system(); }
#else
#define P2P_FLAG_STORE_ORDER __ATOMIC_RELAXED
#define P2P_FLAG_LOAD_ORDER __ATOMIC_RELAXED
__device__ __forceinline__ void p2p_stores_done() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void p2p_waited() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#endif

// spin (bounded) until *flag >= seq; on a timeout the error word is raised (and stays raised:
// every later exchange of this mailbox skips its waits and POISONS its result, see p2p_failed)
__device__ __forceinline__ void p2p_wait(const unsigned long long* flag, unsigned long long seq,
                                         int* err, unsigned long long timeout_ticks) {
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, P2P_FLAG_LOAD_ORDER, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > timeout_ticks) {
      atomicExch(err, 1);
      break;
    }
  }
  p2p_waited();
}

// A wait of THIS exchange (or of an earlier one) timed out: the slots may hold stale data.  The
// callers then return NaN instead of a plausible-looking wrong sum, so that the statistics, the
// loss and every gradient of the step turn NaN — a stalled peer fails the step visibly even if
// nobody calls seg_p2p_status.
__device__ __forceinline__ bool p2p_failed(const int* err) {
  return __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// The exchange of one block of a multi-block kernel: block `b` of `nblocks` (the same launch
// geometry on every rank) carries NV doubles in each of its `nlead` lead threads (lead index
// li); on return v[] holds the sums over the ranks, added in rank order.  EVERY thread of the
// block must call it (barriers inside); blockDim.x >= world.  The last block to finish advances
// the exchange counter — every block has read it by then.
template <int NV>
__device__ __forceinline__ void p2p_block_exchange(const P2PDev& p, int b, int nblocks, bool lead,
                                                   int li, int nlead, double (&v)[NV]) {
  __shared__ unsigned long long s_seq;
  __shared__ int s_bad;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_seq = *p.seq + 1;
    s_bad = *p.err;
  }
  __syncthreads();
  const unsigned long long seq = s_seq;
  const long rec = ((long)b * nlead + li) * NV * 8;  // this lead's record inside a slot
  const long mine = ((long)(seq & 1) * p.world + p.rank) * p.slot_bytes + rec;
  if (lead) {
    for (int r = 0; r < p.world; ++r) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.peer[r] + mine);
#pragma unroll
      for (int k = 0; k < NV; ++k)
        __hip_atomic_store(dst + k, (unsigned long long)__double_as_longlong(v[k]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  p2p_stores_done();
  __syncthreads();
  if (tid < p.world) {
    __hip_atomic_store(p2p_block_flags(p.peer[tid], p.world, p.slot_bytes) +
                           (long)p.rank * P2P_MAX_BLOCKS + b,
                       seq, P2P_FLAG_STORE_ORDER, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!s_bad)
      p2p_wait(p2p_block_flags(p.peer[p.rank], p.world, p.slot_bytes) +
                   (long)tid * P2P_MAX_BLOCKS + b, seq, p.err, p.timeout_ticks);
  }
  __syncthreads();
  if (lead && p2p_failed(p.err)) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = __longlong_as_double(0x7ff8000000000000ll);
  } else if (lead) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
    for (int r = 0; r < p.world; ++r) {
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(
          p.peer[p.rank] + ((long)(seq & 1) * p.world + r) * p.slot_bytes + rec);
#pragma unroll
      for (int k = 0; k < NV; ++k)
        v[k] += __longlong_as_double((long long)__hip_atomic_load(
            src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
  }
  if (tid == 0) {
    const unsigned int done = atomicAdd(p.arrive, 1u);
    if (done == (unsigned int)nblocks - 1u) {
      *p.arrive = 0u;
      *p.seq = seq;
    }
  }
}

}  // namespace seg
