// Fourth-generation GEMM core (r06): the ring of gemm_glds.h — direct-to-LDS staging, four 32-k
// slots of 32 KiB, three slots' DMA in flight across raw s_barriers — driven by FOUR waves of
// 128 x 128 (4 x 4 MFMA 32x32x16 tiles, 256 fp32 accumulators, one wave per SIMD) instead of
// eight waves of 128 x 64.
//
// Why: per 32-k slot the eight-wave kernel reads 8 waves x 2 k-steps x 6 fragments x 1 KiB =
// 96 KiB from LDS and the DMA writes 32 KiB into it — 128 KiB at 128 B/clk = 1024 clk, exactly
// the 1024 clk its 32 MFMAs per SIMD take: the LDS port and the matrix pipe are co-limiting and
// every imperfect overlap stalls.  Four waves of 128 x 128 read 4 x 2 x 8 = 64 KiB per slot
// (port at 75 % of the MFMA time).  Price: one wave per SIMD — nobody else hides this wave's
// latencies, so the k-step is software-pipelined by hand: the 8 fragment reads of k-step i+1
// (and the slot's 8 DMA issues) are interleaved one by one between pairs of the 16 MFMAs of
// k-step i (sched_barrier fences keep hipcc from regrouping them), and the only waits are one
// lgkmcnt(0) per k-step (fragments requested a full k-step = 512 clk earlier) and one counted
// vmcnt + barrier per slot, placed behind the first MFMAs of the k-step so that the pipe has
// work while the waves meet.
//
// Slot image, swizzle and zero-word tails as gemm_glds.h.  No KxK gather form: measured on the
// lab harness (profiles/r06_gemm_w4.md), the stride-1 KxK convolutions were 2 - 8 % slower on four
// waves (the per-lane gather arithmetic has nobody to hide behind) and stay on eight.
#pragma once
#include "gemm_glds.h"

namespace seg {

constexpr int GL4_THREADS = 256;
// Ring depth: four slots, three requests in flight.  Five (all 160 KiB of the CU's LDS) were
// measured and are not faster (profiles/r06_gemm_w4.md: 3.699 vs 3.633 ms on the harness) — the
// slot time is not the DMA round trip divided by the ring depth.  The loop below is written for
// either; the waits for 5 are kept beside those for 4.
constexpr int GL4_RING = 4;
// MFMA pairs at the end of a k-step with nothing issued behind them (they cover the last fragment
// reads' LDS latency): 1 / 2 / 3 measured, 3.607 / 3.653 / 3.645 ms.
constexpr int GL4_COVER = 1;
constexpr int GL4_LDS_BYTES = GL4_RING * GL_SLOT_BYTES;

template <int IM, int JN> struct Gl4Frags { bf16x8 n[JN], m[IM]; };

// "these fragments are produced here": s_waitcnt lgkmcnt(0) with every register of f as an in-out
// operand — hipcc then adds no wait of its own in front of the MFMAs that consume f (it would
// also wait for the NEXT k-step's fragments, requested in between).
template <int IM, int JN>
__device__ __forceinline__ void gl4_landed(Gl4Frags<IM, JN>& f) {
  if constexpr (IM == 4 && JN == 4)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.n[0]), "+v"(f.n[1]), "+v"(f.n[2]), "+v"(f.n[3]), "+v"(f.m[0]),
                   "+v"(f.m[1]), "+v"(f.m[2]), "+v"(f.m[3])
                 :
                 : "memory");
  else if constexpr (IM == 3 && JN == 4)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.n[0]), "+v"(f.n[1]), "+v"(f.n[2]), "+v"(f.n[3]), "+v"(f.m[0]),
                   "+v"(f.m[1]), "+v"(f.m[2])
                 :
                 : "memory");
  else {
    static_assert((IM == 7 && JN == 2) || (IM == 4 && JN == 4) || (IM == 3 && JN == 4), "wave tile");
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.n[0]), "+v"(f.n[1]), "+v"(f.m[0]), "+v"(f.m[1]), "+v"(f.m[2]),
                   "+v"(f.m[3]), "+v"(f.m[4]), "+v"(f.m[5]), "+v"(f.m[6])
                 :
                 : "memory");
  }
}

// Wave layout: WM x (4 / WM) waves, each IM x JN MFMA tiles of 32 x 32: block tile
// (WM * IM * 32) rows x 256 columns —
//   2 x 2 waves of 4 x 4: 256 x 256      2 x 2 of 3 x 4: 192 x 256      1 x 4 of 7 x 2: 224 x 256
// (224 rows: 16770 pixels x 728 channels = 75 x 3 = 225 tiles, ONE round of the 256 CUs, with
// 7/8 of the 256-row tile's MFMAs per wave; 9 fragment reads per 14 MFMAs: LDS port at 91 %.)
// Staging: every wave issues 8 DMAs of 16 B per lane and slot — A chunks wave*4 + {0..3} of 16
// rows (chunks beyond the tile's rows fetch the zero word and are never read), B chunks likewise.
template <int WM, int IM, int JN>
__device__ __forceinline__ void gl4_mainloop(const GemmOperand& A, const GemmOperand& B, int K,
                                             int m0, int n0, lds_byte_t* lds,
                                             f32x16 (&acc)[JN][IM]) {
  constexpr int WN = 4 / WM;
  static_assert(WN * JN == 8, "256 columns");
  constexpr int BM_ROWS = WM * IM * 32;
  constexpr int NP = 8;  // DMA pieces per thread and slot
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_gl_zero);
  const int kv = (lane & 3) ^ ((lane >> 4) & 3);  // logical k vector of this lane, all pieces
  const int nvalid = (K - kv * 8 + 31) >> 5;      // slots in which this lane's k range exists
  const unsigned char* src[NP];
  int inc[NP];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave * 4 + j) * 16 + (lane >> 2);
    const bool aok = r < BM_ROWS && m0 + r < A.rows, bok = n0 + r < B.rows;
    src[j] = aok ? A.base + (long)(m0 + r) * A.ld_bytes + kv * 16 : zero;
    src[4 + j] = bok ? B.base + (long)(n0 + r) * B.ld_bytes + kv * 16 : zero;
    inc[j] = aok ? 64 : 0;
    inc[4 + j] = bok ? 64 : 0;
  }
  // one DMA piece of slot `slot` (pieces 0 .. 3: A, 4 .. 7: B)
  auto issue_piece = [&](int slot, int buf, int j) {
    const bool v = slot < nvalid;
    const unsigned char* p = v ? src[j] : zero;
    src[j] += inc[j];
    lds_byte_t* dst = lds + buf * GL_SLOT_BYTES + (j >> 2) * GL_SUB_BYTES +
                      (wave * 4 + (j & 3)) * 1024;
    __builtin_amdgcn_global_load_lds((glb_byte_t*)p, dst, 16, 0, 0);
  };
  const int nslot = (K + 31) >> 5;
#pragma unroll
  for (int s = 0; s < GL4_RING; ++s) {
#pragma unroll
    for (int j = 0; j < NP; ++j) issue_piece(s, s, j);
  }
  // ---- fragment addressing
  const int r32 = lane & 31, h = lane >> 5, x2 = (lane >> 2) & 3;
  const int rowA = (wm * 32 * IM + r32) * 64, rowB = GL_SUB_BYTES + (wn * 32 * JN + r32) * 64;
  const int ko0 = ((0 + h) ^ x2) << 4, ko1 = ((2 + h) ^ x2) << 4;
  typedef Gl4Frags<IM, JN> Frags;
  constexpr int NF = IM + JN;  // fragments per k-step
  // fragment q of a k-step, in the order the MFMAs first need them: m0, n0 .. n(JN-1), m1, ...
  auto read_one = [&](Frags& f, int buf, int ko, int q) {
    const lds_byte_t* s = lds + buf * GL_SLOT_BYTES;
    if (q == 0) f.m[0] = *(gl_lds_frag_t*)(s + rowA + ko);
    else if (q <= JN) f.n[q - 1] = *(gl_lds_frag_t*)(s + rowB + (q - 1) * 32 * 64 + ko);
    else f.m[q - JN] = *(gl_lds_frag_t*)(s + rowA + (q - JN) * 32 * 64 + ko);
  };
  // MFMA t of a k-step (t = im * JN + jn: row block slowest)
  auto mma_one = [&](const Frags& f, int t, auto zero_in) {
    constexpr bool ZERO = decltype(zero_in)::value;
    const int im = t / JN, jn = t % JN;
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc[jn][im] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.n[jn], f.m[im], ZERO ? z : acc[jn][im],
                                                          0, 0, 0);
  };
  constexpr int NM = IM * JN;   // MFMAs per k-step
  constexpr int P = NM / 2;     // ... issued as pairs, something else in between
  static_assert(NM % 2 == 0, "pairs");
  // k-step 0: reads spread over the gaps behind pairs 0 .. R0-1 — the last GL4_COVER pairs have
  // nothing behind them: their 64 clk each cover the last reads' LDS latency
  constexpr int R0 = P - GL4_COVER;
  // k-step 1: pair 0 goes ahead of the barrier; DMA pieces and reads spread over the gaps IN FRONT
  // of pairs 1 .. R1
  constexpr int R1 = P - 1 - GL4_COVER;
  Frags f0, f1;
  if constexpr (GL4_RING == 5) GL_WAIT_VM(32); else GL_WAIT_VM(24);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int q = 0; q < NF; ++q) read_one(f0, 0, ko0, q);

  int buf = 0;  // j % GL4_RING
  auto slot_body = [&](int j, auto first) {
    const int nbuf = buf + 1 == GL4_RING ? 0 : buf + 1;
    // ---- k-step 0: MFMAs on f0; f1 <- slot j, second k-step
#pragma unroll
    for (int p = 0; p < P; ++p) {
      mma_one(f0, 2 * p, first);
      mma_one(f0, 2 * p + 1, first);
      __builtin_amdgcn_sched_barrier(0);
      if (p < R0) {
#pragma unroll
        for (int q = p * NF / R0; q < (p + 1) * NF / R0; ++q) read_one(f1, buf, ko1, q);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- k-step 1: f1 has landed (requested >= GL4_COVER MFMA pairs ago); one pair goes ahead
    // of the barrier so that the pipe has work while the waves meet
    gl4_landed(f1);
    mma_one(f1, 0, std::false_type{});
    mma_one(f1, 1, std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
    // every read of slot j by this wave has returned; my DMA of slot j+1 has landed
    if constexpr (GL4_RING == 5) GL_WAIT_VM(24); else GL_WAIT_VM(16);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // slot j+RING -> the buffer slot j occupied; f0 <- slot j+1, first k-step
#pragma unroll
    for (int p = 1; p < P; ++p) {
      const int g = p - 1;
      if (g < R1) {
#pragma unroll
        for (int i = g * NP / R1; i < (g + 1) * NP / R1; ++i) issue_piece(j + GL4_RING, buf, i);
#pragma unroll
        for (int q = g * NF / R1; q < (g + 1) * NF / R1; ++q) read_one(f0, nbuf, ko0, q);
        __builtin_amdgcn_sched_barrier(0);
      }
      mma_one(f1, 2 * p, std::false_type{});
      mma_one(f1, 2 * p + 1, std::false_type{});
      __builtin_amdgcn_sched_barrier(0);
    }
    buf = nbuf;
  };
  slot_body(0, std::true_type{});
  for (int j = 1; j < nslot; ++j) slot_body(j, std::false_type{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // trailing zero DMAs + last read
  __builtin_amdgcn_s_barrier();
}

}  // namespace seg
