// Third-generation MFMA GEMM core for gfx950: direct-to-LDS staging (global_load_lds, 16 B per
// lane), XOR-swizzled lane-linear LDS image, a RING of four 32-k slots with three slots' DMA kept
// in flight across the barriers (counted s_waitcnt vmcnt, raw s_barrier), 256 x 256 block tile,
// eight waves as 2 (row halves) x 4 (column quarters), 128 x 64 per wave = 4 x 2 MFMA 32x32x16
// tiles (128 fp32 accumulators).
//
//   D[m][n] = sum_k A[m][k] * B[n][k]        A: [M][lda] bf16 (k contiguous)
//                                            B: [N][ldb] bf16 (k contiguous)
//
// Used by the 1x1 / stride-1 convolution forward + data gradient (conv_gemm_glds.hip: A = pixels,
// B = packed weights) — the operands that need no element-wise prologue, which is what the BN
// fold (fold.hip) leaves for 63 of the 79 GEMM-shaped convolutions of DeepLabv3+/xception65 —
// and, with the per-lane DMA source gathered per tap, by the stride-1 KxK convolutions.
// (The two-stage 64-k predecessors of the ring and their ablation hooks — profiles/r02_gemm_lab.md
// — were removed in r03.)
//
// Tails: rows >= M / N and k >= K are fetched from a 16-byte zero word (per-lane source
// address), so no operand padding is required and nothing is read out of bounds.
#pragma once
#include <type_traits>
#include "common.h"

namespace seg {

constexpr int GL_BM = 256, GL_BN = 256;
constexpr int GL_THREADS = 512;
constexpr int GL_LDS_BYTES = 128 * 1024;                 // four ring slots of 32 KiB

typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
typedef __attribute__((address_space(1))) const unsigned char glb_byte_t;

struct GemmOperand {
  const unsigned char* base;  // element (0,0)
  long ld_bytes;              // row pitch in bytes
  int rows;                   // valid rows
};

// 16 bytes of zeros in global memory: the DMA source of every out-of-range vector
__device__ __attribute__((aligned(16))) const unsigned int g_gl_zero[4] = {0u, 0u, 0u, 0u};

#define GL_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// Fragment registers of one k-step (16 k): 2 column-block fragments, IMS row-block fragments.
template <int IMS> struct GlFragsT { bf16x8 n[2], m[IMS]; };
typedef GlFragsT<4> GlFrags;
typedef const __attribute__((address_space(3))) bf16x8 gl_lds_frag_t;

// MFMAs of row blocks [IM0, IM1) of one k-step.  ZERO: the accumulator input is the constant 0
// (first k-step of a tile: saves zero-filling 128 registers before the loop).
template <int IM0, int IM1, bool ZERO = false, int IMS = 4>
__device__ __forceinline__ void gl_mma_part(const GlFragsT<IMS>& f, f32x16 (&acc)[2][IMS]) {
#pragma unroll
  for (int im = IM0; im < IM1; ++im) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc[0][im] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.n[0], f.m[im], ZERO ? z : acc[0][im],
                                                         0, 0, 0);
    acc[1][im] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.n[1], f.m[im], ZERO ? z : acc[1][im],
                                                         0, 0, 0);
  }
}
template <int IMS>
__device__ __forceinline__ void gl_mma_step(const GlFragsT<IMS>& f, f32x16 (&acc)[2][IMS]) {
  gl_mma_part<0, IMS, false, IMS>(f, acc);
}

// ---------------------------------------------------------------------------------------------
// The ring: the DMA round trip under full load (~2 us: 198 CUs pulling 64 KB per
// K tile from L2) is longer than one K tile of MFMAs, so two 64-k stages leave the waves parked
// in s_waitcnt vmcnt (tools/lab ablation: 1.58 us per K tile with the in-loop DMA, 1.16 without).
// Here the 128 KiB hold FOUR slots of 32 k (A [256][64 B] + B [256][64 B] = 32 KiB each), slot
// j+4 is issued as soon as slot j has been read, and up to three slots (96 KiB per CU) are in
// flight while one is multiplied.  Slot image: row pitch 64 B, vector kv (0..3) of row r at
// r*64 + ((kv ^ ((r >> 2) & 3)) << 4) — conflict-free for the ds_read_b128 fragment pattern.
// One barrier per slot, placed after the slot's last fragment read has returned.
// Slots past the end of K are issued as all-zero DMAs (every lane fetches the zero word), which
// keeps the vmcnt arithmetic uniform; the caller's epilogue starts after vmcnt(0) + barrier.
constexpr int GL_SLOT_BYTES = 32 * 1024, GL_SUB_BYTES = 16 * 1024;

// Implicit-GEMM form of the A operand (KXK): row m = output pixel (n, ho, wo) of a stride-1 KxK
// convolution over an NHWC tensor with C % 32 == 0, so that every 32-k slot lies inside ONE tap
// (kh, kw): the per-lane DMA source of slot s is the input pixel (ho - pad + kh*dil,
// wo - pad + kw*dil), channels cb*32 .. +31 — or the zero word outside the image.
struct GlConvA {
  int M, Hi, Wi, Ho, Wo, KW, pad, dil, cpt;  // cpt = C / 32: slots per tap
};

// IMS: 32-row blocks per wave — 4 = the 256-row tile, 3 = a 192-row tile (r05: the launcher picks
// the row count whose tile count fills the 256 CUs in fewer / shorter rounds; the A half of a
// slot then holds 192 rows, its last four DMA pieces fetch the zero word)
template <bool KXK = false, int IMS = 4>
__device__ __forceinline__ void gl_mainloop_ring(const GemmOperand& A, const GemmOperand& B, int K,
                                                 int m0, int n0, lds_byte_t* lds,
                                                 f32x16 (&acc)[2][IMS], const GlConvA* cg = nullptr) {
  constexpr int BM_ROWS = 64 * IMS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_gl_zero);
  // ---- DMA pieces of this thread: chunks wave*2 + {0,1} (16 rows each) of A and of B
  const int kv = (lane & 3) ^ ((lane >> 4) & 3);  // logical k vector of this lane, all pieces
  const int nvalid = (K - kv * 8 + 31) >> 5;      // slots in which this lane's k range exists
  const unsigned char* src[4];
  long inc[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 16 + (lane >> 2);
    const bool aok = r < BM_ROWS && m0 + r < A.rows, bok = n0 + r < B.rows;
    src[j] = aok ? A.base + (long)(m0 + r) * A.ld_bytes + kv * 16 : zero;
    src[2 + j] = bok ? B.base + (long)(n0 + r) * B.ld_bytes + kv * 16 : zero;
    inc[j] = aok ? 64 : 0;
    inc[2 + j] = bok ? 64 : 0;
  }
  // KXK: this thread's two output pixels and the running tap of the slot being issued (slots are
  // issued strictly in order)
  int cpix[2] = {0, 0}, ch0[2] = {0, 0}, cw0[2] = {0, 0};
  bool cval[2] = {false, false};
  int tkh = 0, tkw = 0, tcb = 0;
  if (KXK) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rt = (wave * 2 + j) * 16 + (lane >> 2);
      const int m = m0 + rt;
      cval[j] = rt < BM_ROWS && m < cg->M;
      const int mm = cval[j] ? m : 0;
      const int wo = mm % cg->Wo, t = mm / cg->Wo;
      const int ho = t % cg->Ho, n = t / cg->Ho;
      cpix[j] = n * cg->Hi * cg->Wi;
      ch0[j] = ho - cg->pad;
      cw0[j] = wo - cg->pad;
    }
  }
  auto issue = [&](int slot) {
    lds_byte_t* base = lds + (slot & 3) * GL_SLOT_BYTES + (wave * 2) * 1024;
    const bool v = slot < nvalid;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned char* p = v ? src[j] : zero;
      if (KXK && j < 2) {
        const int hi = ch0[j] + tkh * cg->dil, wi = cw0[j] + tkw * cg->dil;
        const bool ok = v && cval[j] && hi >= 0 && hi < cg->Hi && wi >= 0 && wi < cg->Wi;
        p = ok ? A.base + (long)(cpix[j] + hi * cg->Wi + wi) * A.ld_bytes + tcb * 64 + kv * 16
               : zero;
      }
      src[j] += inc[j];
      lds_byte_t* dst = base + (j & 1) * 1024 + (j >> 1) * GL_SUB_BYTES;
      __builtin_amdgcn_global_load_lds((glb_byte_t*)p, dst, 16, 0, 0);
    }
    if (KXK) {  // (uniform) advance the tap: channel block fastest, then kw, then kh
      if (++tcb == cg->cpt) {
        tcb = 0;
        if (++tkw == cg->KW) { tkw = 0; ++tkh; }
      }
    }
  };
  const int nslot = (K + 31) >> 5;
  issue(0); issue(1); issue(2); issue(3);
  // ---- fragment addressing
  const int r32 = lane & 31, h = lane >> 5, x2 = (lane >> 2) & 3;
  const int rowA = (wm * 32 * IMS + r32) * 64, rowB = GL_SUB_BYTES + (wn * 64 + r32) * 64;
  const int ko0 = ((0 + h) ^ x2) << 4, ko1 = ((2 + h) ^ x2) << 4;
  typedef GlFragsT<IMS> Frags;
  auto read = [&](Frags& f, int slot, int ko) {
    const lds_byte_t* s = lds + (slot & 3) * GL_SLOT_BYTES;
    f.n[0] = *(gl_lds_frag_t*)(s + rowB + ko);
    f.n[1] = *(gl_lds_frag_t*)(s + rowB + 32 * 64 + ko);
#pragma unroll
    for (int im = 0; im < IMS; ++im) f.m[im] = *(gl_lds_frag_t*)(s + rowA + im * 32 * 64 + ko);
  };
  Frags f0, f1;
  GL_WAIT_VM(12);
  __builtin_amdgcn_s_barrier();
  read(f0, 0, ko0);
  // One slot.  The second k-step's fragment reads are issued BEHIND the first two MFMAs of the
  // first k-step: hipcc waits lgkmcnt(0) — every outstanding LDS read — in front of the first
  // MFMA that consumes f0, so nothing newer may be in flight at that point.
  auto slot_body = [&](int j, auto first) {
    constexpr bool FIRST = decltype(first)::value;
    gl_mma_part<0, 1, FIRST, IMS>(f0, acc);
    __builtin_amdgcn_sched_barrier(0);
    read(f1, j, ko1);
    __builtin_amdgcn_sched_barrier(0);
    gl_mma_part<1, IMS, FIRST, IMS>(f0, acc);
    __builtin_amdgcn_sched_barrier(0);
    // my reads of slot j have returned; my DMA of slot j+1 has landed (j+2, j+3 in flight).
    // f1 is passed THROUGH the asm: hipcc then sees it as produced here and does not put its
    // own lgkmcnt(0) — which would also wait for the NEXT slot's fragments requested right
    // behind the barrier — in front of the MFMAs that consume it.
    if constexpr (IMS == 4)
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)"
                   : "+v"(f1.n[0]), "+v"(f1.n[1]), "+v"(f1.m[0]), "+v"(f1.m[1]), "+v"(f1.m[2]),
                     "+v"(f1.m[3])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)"
                   : "+v"(f1.n[0]), "+v"(f1.n[1]), "+v"(f1.m[0]), "+v"(f1.m[1]), "+v"(f1.m[2])
                   :
                   : "memory");
    __builtin_amdgcn_s_barrier();
    issue(j + 4);
    read(f0, j + 1, ko0);
    __builtin_amdgcn_sched_barrier(0);
    gl_mma_step<IMS>(f1, acc);
    __builtin_amdgcn_sched_barrier(0);
  };
  slot_body(0, std::true_type{});
  for (int j = 1; j < nslot; ++j) slot_body(j, std::false_type{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // trailing zero DMAs + last read
  __builtin_amdgcn_s_barrier();
}

}  // namespace seg
