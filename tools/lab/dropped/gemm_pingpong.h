// DROPPED r05 experiment (kept for the record, not compiled into the library): the ping-pong
// form of gemm_glds.h's ring.  Paste into gemm_glds.h (namespace seg) and call instead of
// gl_mainloop_ring to reproduce: 74.2 vs 68.4 us on 1024 -> 1536 @16770, 103.8 vs 94.1 us on
// 1536 -> 1536, bit-identical results (tools/lab/gemm_ab, profiles/r05_gemm_ab.md).
// ---------------------------------------------------------------------------------------------
// Ping-pong form of the same ring (r05).  In gl_mainloop_ring all eight waves run the same
// program in lockstep: after the slot's barrier BOTH waves of a SIMD issue DMA pieces and wait
// for fragment reads, then BOTH queue 16 MFMAs on the one matrix pipe they share (PMC r04: the
// pipe is busy 24 % of the launch, ~50 % of the loop).  Here a slot is two barrier intervals —
//   X: issue the DMA of slot j+3, read ALL fragments of slot j (12 x ds_read_b128), wait
//   Y: the slot's 16 MFMAs, nothing else
// — and waves 4-7 pass one extra barrier before their first interval (waves 0-3 one after their
// last), so that in every interval one wave of each SIMD is in Y while its partner is in X: the
// loads of one hide under the matrix work of the other.
// Ring hazards (4 slots, s = barrier interval; group 0 runs X_j in s = 2j, Y_j in 2j+1, group 1
// one interval later): a wave waits for ITS pieces of slot j+1 at the end of X_j (vmcnt(8):
// slots j+2, j+3 stay in flight) — that is interval 2j / 2j+1, and the first read of slot j+1 is
// in interval 2j+2, behind a barrier both groups passed after their waits.  The DMA of slot j+3
// overwrites the image of slot j-1, last read in interval 2j-1 (group 1's X_{j-1}), i.e. before
// the barrier that opens interval 2j.  Same accumulation order per output as gl_mainloop_ring
// (k-step 0 of a slot, then k-step 1): bit-identical results.
template <bool KXK = false, int IMS = 4, bool PRIO = false>
__device__ __forceinline__ void gl_mainloop_pingpong(const GemmOperand& A, const GemmOperand& B,
                                                     int K, int m0, int n0, lds_byte_t* lds,
                                                     f32x16 (&acc)[2][IMS],
                                                     const GlConvA* cg = nullptr) {
  constexpr int BM_ROWS = 64 * IMS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_gl_zero);
  const int kv = (lane & 3) ^ ((lane >> 4) & 3);
  const int nvalid = (K - kv * 8 + 31) >> 5;
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
    if (KXK) {
      if (++tcb == cg->cpt) {
        tcb = 0;
        if (++tkw == cg->KW) { tkw = 0; ++tkh; }
      }
    }
  };
  const int nslot = (K + 31) >> 5;
  issue(0); issue(1); issue(2);
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
  GL_WAIT_VM(8);  // my pieces of slot 0 (slots 1, 2 in flight)
  __builtin_amdgcn_s_barrier();
  if (wave >= 4) __builtin_amdgcn_s_barrier();  // group 1 runs one interval behind
  auto slot_body = [&](int j, auto first) {
    constexpr bool FIRST = decltype(first)::value;
    // ---- X
    issue(j + 3);
    read(f0, j, ko0);
    read(f1, j, ko1);
    // my fragment reads have returned, my pieces of slot j+1 have landed (j+2, j+3 in flight);
    // the fragments are passed THROUGH the wait so that no MFMA can be scheduled above it
    if constexpr (IMS == 4)
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)"
                   : "+v"(f0.n[0]), "+v"(f0.n[1]), "+v"(f0.m[0]), "+v"(f0.m[1]), "+v"(f0.m[2]),
                     "+v"(f0.m[3]), "+v"(f1.n[0]), "+v"(f1.n[1]), "+v"(f1.m[0]), "+v"(f1.m[1]),
                     "+v"(f1.m[2]), "+v"(f1.m[3])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)"
                   : "+v"(f0.n[0]), "+v"(f0.n[1]), "+v"(f0.m[0]), "+v"(f0.m[1]), "+v"(f0.m[2]),
                     "+v"(f1.n[0]), "+v"(f1.n[1]), "+v"(f1.m[0]), "+v"(f1.m[1]), "+v"(f1.m[2])
                   :
                   : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- Y
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    gl_mma_part<0, IMS, FIRST, IMS>(f0, acc);
    gl_mma_part<0, IMS, false, IMS>(f1, acc);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  slot_body(0, std::true_type{});
  for (int j = 1; j < nslot; ++j) slot_body(j, std::false_type{});
  if (wave < 4) __builtin_amdgcn_s_barrier();  // pairs with group 1's last barrier
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // trailing zero DMAs
  __builtin_amdgcn_s_barrier();
}

