// 256x256 implicit-GEMM conv with a DEEPER ACTIVATION PREFETCH, for the deep-K layers (3x3 256->256 of
// FPN/RPN/mask head, fc1/fc2, res4/res5 conv2).
//
// Same math, operand roles (weights = MFMA A operand, activations = B operand, v_mfma_f32_16x16x32_f16),
// LDS-DMA staging, 128-byte XOR-swizzled LDS rows, lockstep schedule and epilogue as the 256x256 variant of
// conv_igemm.hip.  What changes is the LDS budget: that kernel double-buffers both operands (2 x 64 KB), so
// every operand byte has ONE K step (~1 us of MFMA work) to arrive, and with one resident workgroup per CU
// nothing else runs while it waits (rocprofv3: MFMA busy 45-49 %, L2 hit ~70 % -> a third of the activation
// rows come from HBM/MALL with more latency than that).  Here the 160 KB are split by operand:
//
//      activations   3 stages x 32 KB   -> issued TWO K steps ahead
//      weights       2 stages x 32 KB   -> issued one step ahead (the 1.2 MB filter stays L2-resident)
//
// Issue order is always weights then activations, so the wait that publishes a step is `s_waitcnt vmcnt(4)`:
// everything but the 4 youngest LDS-DMA pieces (the activations two steps ahead) has landed; raw s_barrier.
// The fragment reads are software-pipelined over half K steps against the MFMAs (see the main loop).
#include "common.h"

namespace {

constexpr int BN = 256, NT = 512, MI = 4, WCH = 4;   // NJ (16-pixel blocks per wave) = 8: 256-pixel tile; 4: the 128-pixel tiles of a split last round
constexpr int ASTAGE = 256 * 128, WSTAGE = BN * 128;      // 32 KB each
constexpr int W_BASE = 3 * ASTAGE;
constexpr int LDS_BYTES = 3 * ASTAGE + 2 * WSTAGE;        // 160 KB

__device__ __forceinline__ void glds16(const half_t* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// One tile: NJ * 32 pixels x 256 channels.  TRAIN: instantiation with the backward-epilogue options down / res32 / mask (conv_igemm.hip),
// compiled out of the inference kernel
// SPLIT: the split-operand precision mode (ConvParams::split, common.h).  A K step covers 32 channels and its 128-byte LDS rows are [32 hi halfs | 32 lo
// halfs] (data chunks 0-3 from the hi plane, 4-7 from the lo plane: a per-lane constant in the LDS-DMA source pointers), so the stages, the LDS-DMA pieces
// and the fragment reads of a step are those of the fp16 kernel -- the "first half" fragments are the hi ones, the "second half" the lo ones -- while the
// step runs THREE MFMA blocks, W_hi.X_lo, W_hi.X_hi, W_lo.X_hi (the order conv_igemm.hip uses), with the reads of the other fragment sets in flight under
// them (own main loop below): 96 MFMAs per 24 fragment reads and 8 LDS-DMA pieces instead of 64, so the LDS port that co-bounds the fp16 kernel has slack.
// The epilogue descales by the row's power of two and writes hi / lo planes.
template <int DBG, bool TRAIN, int NJ, bool SPLIT>
__device__ __forceinline__ void conv_deep_tile(const ConvParams& p, char* smem, const half_t* g_in, const half_t* g_w, const float* g_bias, void* g_out,
                                               float* g_head_out, const int Ho, const int Wo, const int in_Hp, const int in_Wp, const int out_Hp, const int out_Wp,
                                               const int M, const int m0, const int n0, const int q, const long long in_lo, const long long out_lo,
                                               const float* g_wscale) {
  constexpr int APS = (NJ + 1) / 2;    // activation staging passes of 64 rows = LDS-DMA pieces per wave and stage (odd NJ: the upper half of the
                                       // last pass lands in LDS rows nothing reads -- every wave issues the same number of pieces, so one counted wait serves all)
  constexpr int WPX = NJ * 16;         // pixels per wave
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpx = wave / WCH, wch = wave % WCH;
  (void)q;

  // ---- staging pointers (identical to conv_igemm<2,4,4,8>): 4 passes of 64 rows for each operand
  const int lrow = lane >> 3, lchk = lane & 7;
  const half_t* aptr[APS];
  const half_t* wptr[4];
  // SPLIT: the same per-lane source positions as 32-bit BYTE offsets from the (wave-uniform) tensor bases, so that an LDS-DMA piece is addressed as
  // SGPR base + VGPR offset: half the address registers and one 32-bit add per piece instead of a 64-bit one -- the registers the split loop's
  // placement of the pieces between its MFMA blocks needs (launch_conv_deep checks that both planes of a tensor lie within 4 GB of its base)
  unsigned aoffb[APS], woffb[4];
#pragma unroll
  for (int ps = 0; ps < APS; ++ps) {
    int m = m0 + ps * 64 + wave * 8 + lrow;
    if (m >= M) m = M - 1;
    const int x = m % Wo;
    const int t = m / Wo;
    const int y = t % Ho;
    const int n = t / Ho;
    const long long base =
        ((long long)(n * in_Hp + y * p.stride + p.in_off) * in_Wp + x * p.stride + p.in_off) * p.in_Cs;
    if constexpr (SPLIT) aoffb[ps] = (unsigned)((base + ((lchk ^ lrow) & 3) * 8 + ((lchk ^ lrow) >> 2) * in_lo) * 2);
    else aptr[ps] = g_in + base + (lchk ^ lrow) * 8;
  }
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = ps * 64 + wave * 8 + lrow;
    const int key = (row & 3) | (((row >> 4) & 1) << 2);
    if constexpr (SPLIT) woffb[ps] = (unsigned)(((long long)(n0 + row) * p.Kpad + ((lchk ^ key) & 3) * 8 + ((lchk ^ key) >> 2) * p.w_lo) * 2);
    else wptr[ps] = g_w + (long long)(n0 + row) * p.Kpad + (lchk ^ key) * 8;
  }
  constexpr int KST = SPLIT ? 32 : 64;      // channels a K step covers
  const int nk = p.KH * p.KW * (p.Cin / KST);

  // two independent walkers over the K steps (64-channel slice outer, taps inner -- conv_igemm.hip): the
  // activation walker runs one step ahead of the weight walker
  int akh = 0, akw = 0, ac0 = 0, an = 0;      // an = index of the next activation step to issue
  auto stage_a = [&]() {
    const int off = (akh * in_Wp + akw) * p.in_Cs + ac0;
    if (++akw == p.KW) { akw = 0; if (++akh == p.KH) { akh = 0; ac0 += KST; } }
    char* abase = smem + (an % 3) * ASTAGE;
    ++an;
    if (DBG & 1) return;   // ceiling experiment: no global traffic
    if (DBG & 8) return;   // ... no activation traffic only
    if constexpr (SPLIT) {
#pragma unroll
      for (int ps = 0; ps < APS; ++ps) glds16((const half_t*)((const char*)g_in + (aoffb[ps] + (unsigned)(off * 2))), abase + (ps * 64 + wave * 8) * 128);
    } else {
#pragma unroll
      for (int ps = 0; ps < APS; ++ps) glds16(aptr[ps] + off, abase + (ps * 64 + wave * 8) * 128);
    }
  };
  int wkh = 0, wkw = 0, wc0 = 0, wn = 0;
  auto stage_w = [&]() {
    const int koff = (wkh * p.KW + wkw) * p.Cin + wc0;
    if (++wkw == p.KW) { wkw = 0; if (++wkh == p.KH) { wkh = 0; wc0 += KST; } }
    char* wbase = smem + W_BASE + (wn & 1) * WSTAGE;
    ++wn;
    if (DBG & 1) return;
    if (DBG & 4) return;   // ... no weight traffic only
    if constexpr (SPLIT) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) glds16((const half_t*)((const char*)g_w + (woffb[ps] + (unsigned)(koff * 2))), wbase + (ps * 64 + wave * 8) * 128);
    } else {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) glds16(wptr[ps] + koff, wbase + (ps * 64 + wave * 8) * 128);
    }
  };

  const int fi = lane & 15, fq = lane >> 4, fkey = lane & 7;
  int w_off[MI], x_off[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) w_off[i] = W_BASE + (wch * 64 + (fi >> 2) * 16 + i * 4 + (fi & 3)) * 128;
#pragma unroll
  for (int j = 0; j < NJ; ++j) x_off[j] = (wpx * WPX + j * 16 + fi) * 128;
  const int c0_off = (fq ^ fkey) * 16, c1_off = ((4 + fq) ^ fkey) * 16;

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline over half K steps.  F0 / F1 hold the fragments of the first / second 32-deep half of a
  // step.  While the MFMAs of one half run, the reads of the next half are in flight; the workgroup barrier that
  // publishes step t+1 sits in the MIDDLE of step t (after the first half's MFMAs), so the reads of
  // step t+1's first half overlap the MFMAs of step t's second half:
  //
  //   reads F1(t) | MFMA F0(t) | wait own pieces of t+1, barrier | issue w(t+2), acts(t+3) | reads F0(t+1) | MFMA F1(t)
  //
  // At the barrier every wave has finished reading step t's buffers (lgkmcnt(0)), so they are refilled there.
  // The fragment reads are inline asm: hipcc's own waitcnt insertion drains lgkmcnt(0) before the first MFMA of a
  // half step even though only the OLDER twelve reads feed it, which serialises the LDS phase against the matrix
  // phase again.  With asm reads the compiler sees no pending LDS operation; the counted waits below are the only
  // ones, fenced with sched_barrier on both sides (MFMAs are register-only and would otherwise move across them).
  half8 wf0[MI], xf0[NJ], wf1[MI], xf1[NJ];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned wrow = (unsigned)(W_BASE + (wch * 64 + (fi >> 2) * 16 + (fi & 3)) * 128);
  const unsigned xrow = (unsigned)((wpx * WPX + fi) * 128);
  const unsigned wa0 = lds0 + wrow + c0_off, wa1 = lds0 + wrow + c1_off;
  const unsigned xa0 = lds0 + xrow + c0_off, xa1 = lds0 + xrow + c1_off;
#define RS_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RS_READS(wf, xf, wa, xa)                                                        \
  RS_DSR(wf[0], wa, 0); RS_DSR(wf[1], wa, 512); RS_DSR(wf[2], wa, 1024); RS_DSR(wf[3], wa, 1536);   \
  RS_DSR(xf[0], xa, 0); RS_DSR(xf[1], xa, 2048);                                                    \
  if constexpr (NJ > 2) { RS_DSR(xf[2], xa, 4096); }                                                \
  if constexpr (NJ > 3) { RS_DSR(xf[3], xa, 6144); }                                                \
  if constexpr (NJ > 4) { RS_DSR(xf[4], xa, 8192); }                                                \
  if constexpr (NJ > 5) { RS_DSR(xf[5], xa, 10240); }                                               \
  if constexpr (NJ > 6) { RS_DSR(xf[6], xa, 12288); }                                               \
  if constexpr (NJ > 7) { RS_DSR(xf[7], xa, 14336); }
  auto reads0 = [&](int ab, int wb) {
    if (DBG & 16) return;  // ceiling experiment: no fragment reads (MFMAs on whatever the registers hold)
    const unsigned wa = wa0 + ((DBG & 2) ? 0 : wb) * WSTAGE, xa = xa0 + ((DBG & 2) ? 0 : ab) * ASTAGE;
    RS_READS(wf0, xf0, wa, xa)
  };
  auto reads1 = [&](int ab, int wb) {
    if (DBG & 16) return;
    const unsigned wa = wa1 + ((DBG & 2) ? 0 : wb) * WSTAGE, xa = xa1 + ((DBG & 2) ? 0 : ab) * ASTAGE;
    RS_READS(wf1, xf1, wa, xa)
  };
#ifdef RS_CLOCK_PROBE
  // in-kernel clock = d(s_memtime) / d(s_memrealtime) * 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6); stamps go to a
  // buffer nothing else reads
  const unsigned long long pr_t0 = __builtin_amdgcn_s_memtime(), pr_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef RS_SPLIT_PHASES
  const unsigned long long ph_entry = __builtin_amdgcn_s_memtime();
  unsigned long long ph_loop0 = 0, ph_loop1 = 0;
#endif
  // prologue: w(0), acts(0), acts(1); publish step 0; then w(1), acts(2) and the first fragments
  stage_w();
  stage_a();
  if (nk > 1) stage_a();
  if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APS) : "memory");       // all but the youngest activation stage (APS pieces)
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if (nk > 1) stage_w();
#ifndef RS_SPLIT_SCHED_V1
  if constexpr (!SPLIT) { if (nk > 2) stage_a(); }      // the split loop issues acts(t+2) inside step t, acts(2) included
#else
  if (nk > 2) stage_a();
#endif
  if constexpr (SPLIT) {
    // Split-operand main loop.  Register sets: wf0 = W_hi, wf1 = W_lo, xf0 = X_hi, xf1 = X_lo of the current step.  Per step, three MFMA blocks with
    // the reads of the sets they do not touch in flight, one barrier:
    //
    //   reads X_hi(t) | MFMA W_hi.X_lo | reads W_lo(t) | MFMA W_hi.X_hi | own pieces of t+1 landed, barrier | issue w(t+2), acts(t+3),
    //   reads W_hi(t+1), X_lo(t+1) | MFMA W_lo.X_hi
    //
    // At the barrier every wave has read all four sets of step t (lgkmcnt(0)), so step t's buffers are refilled there.  The two waves of a SIMD take
    // the last block and the issue work in opposite order, as in the fp16 loop below.
#define RS_RD_W(wf, wa) RS_DSR(wf[0], wa, 0); RS_DSR(wf[1], wa, 512); RS_DSR(wf[2], wa, 1024); RS_DSR(wf[3], wa, 1536);
#define RS_RD_X(xf, xa)                                                                             \
  RS_DSR(xf[0], xa, 0); RS_DSR(xf[1], xa, 2048);                                                    \
  if constexpr (NJ > 2) { RS_DSR(xf[2], xa, 4096); }                                                \
  if constexpr (NJ > 3) { RS_DSR(xf[3], xa, 6144); }                                                \
  if constexpr (NJ > 4) { RS_DSR(xf[4], xa, 8192); }                                                \
  if constexpr (NJ > 5) { RS_DSR(xf[5], xa, 10240); }                                               \
  if constexpr (NJ > 6) { RS_DSR(xf[6], xa, 12288); }                                               \
  if constexpr (NJ > 7) { RS_DSR(xf[7], xa, 14336); }
    {
      const unsigned wa = wa0, xa = xa1;         // W_hi(0), X_lo(0)
      RS_RD_W(wf0, wa)
      RS_RD_X(xf1, xa)
    }
    int sab = 0;                                 // t % 3
#ifdef RS_SPLIT_PHASES
    // diagnostic build (tools/ubench/split_phases.py): shader-clock stamps at the three points of a step where no LDS read is in flight (s_memtime is
    // counted in lgkmcnt, so it cannot sit between the counted waits): before the vmcnt wait, before the barrier, after the barrier
    unsigned long long ph_work = 0, ph_vm = 0, ph_bar = 0, ph_last = __builtin_amdgcn_s_memtime();
    ph_loop0 = ph_last;
#endif
    for (int t = 0; t < nk; ++t) {
      const unsigned wst = (t & 1) * WSTAGE, ast = sab * ASTAGE;
      __builtin_amdgcn_sched_barrier(0);
      { const unsigned xa = xa0 + ast; RS_RD_X(xf0, xa) }                 // X_hi(t)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NJ) : "memory");        // W_hi, X_lo (older) have landed; X_hi stays in flight
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[i], xf1[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#ifndef RS_SPLIT_SCHED_V1
      // The activation pieces acts(t+2) are issued HERE, between the first two MFMA blocks, not together with the weight pieces behind the barrier: that
      // chunk (8 LDS-DMA pieces with their address arithmetic + 12 fragment reads, ~900 cycles) was longer than the 512-cycle MFMA block the partner wave
      // of the SIMD covers it with -- in-kernel stamps (tools/ubench/split_phases.py) showed waves 0-3 waiting 1 085 cycles per step at the barrier for
      // waves 4-7.  (acts(t+2) goes to the buffer of step t-1, free since the last barrier -- for t = 0 a buffer nothing has used yet.)
      if (t + 2 < nk) stage_a();
#endif
      { const unsigned wa = wa1 + wst; RS_RD_W(wf1, wa) }                 // W_lo(t)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                  // X_hi has landed
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[i], xf0[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const int snext = sab == 2 ? 0 : sab + 1;
      if (t + 1 < nk) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // W_lo landed = my reads of step t's buffers are done
#ifdef RS_SPLIT_PHASES
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long ph1 = __builtin_amdgcn_s_memtime();
        ph_work += ph1 - ph_last;
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APS) : "memory");   // my pieces of step t+1 (all but acts(t+2))
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RS_SPLIT_PHASES
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long ph2 = __builtin_amdgcn_s_memtime();
        ph_vm += ph2 - ph1;
        __builtin_amdgcn_sched_barrier(0);
#endif
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#ifdef RS_SPLIT_PHASES
        ph_last = __builtin_amdgcn_s_memtime();
        ph_bar += ph_last - ph2;
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (wpx) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[i], xf0[j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 2 < nk) stage_w();                 // w(t+2)    -> weight buffer of step t
#ifdef RS_SPLIT_SCHED_V1
        if (t + 3 < nk) stage_a();                 // acts(t+3) -> activation buffer of step t
#endif
        {
          const unsigned wa = wa0 + ((t + 1) & 1) * WSTAGE, xa = xa1 + snext * ASTAGE;      // W_hi(t+1), X_lo(t+1)
          RS_RD_W(wf0, wa)
          RS_RD_X(xf1, xa)
        }
        if (wpx) { sab = snext; continue; }
      } else {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // last step: W_lo has landed
      }
      sab = snext;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[i], xf0[j], acc[i][j], 0, 0, 0);
    }
#undef RS_RD_W
#undef RS_RD_X
#ifdef RS_SPLIT_PHASES
    ph_loop1 = __builtin_amdgcn_s_memtime();
    if (p.probe && lane == 0) {
      long long* o = p.probe + ((long long)q * 8 + wave) * 8;
      o[0] = (long long)ph_work; o[1] = (long long)ph_vm; o[2] = (long long)ph_bar; o[3] = nk;
      o[4] = (long long)(ph_loop0 - ph_entry); o[5] = (long long)(ph_loop1 - ph_loop0);
    }
#endif
  } else {
  reads0(0, 0);
  int abuf = 0;                                  // t % 3
  for (int t = 0; t < nk; ++t) {
    __builtin_amdgcn_sched_barrier(0);
    reads1(abuf, t & 1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4 + NJ) : "memory");   // F0 (the 4 + NJ older reads) has landed; F1 stays in flight
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[i], xf0[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const int anext = abuf == 2 ? 0 : abuf + 1;
    if (t + 1 < nk) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // F1 landed = my reads of step t's buffers are done
      if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APS) : "memory");   // my pieces of step t+1 (all but acts(t+2))
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#ifndef RS_DEEP_LOCKSTEP
      // The two waves of a SIMD (w and w+4, i.e. wpx 0 / 1) take the second half's MFMAs and the issue work (next loads, next
      // fragment reads) in OPPOSITE order, so that one wave's non-matrix instructions run while the other keeps the MFMA pipe busy.
      if (wpx) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[i], xf1[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      if (t + 2 < nk) stage_w();                 // w(t+2)    -> weight buffer of step t
      if (t + 3 < nk) stage_a();                 // acts(t+3) -> activation buffer of step t
      reads0(anext, (t + 1) & 1);
#ifndef RS_DEEP_LOCKSTEP
      if (wpx) { abuf = anext; continue; }
#endif
    }
    else {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // last step: F1 has landed
    }
    abuf = anext;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[i], xf1[j], acc[i][j], 0, 0, 0);
  }
  }

#ifdef RS_CLOCK_PROBE
  if (p.probe && tid == 0) {
    p.probe[2 * q] = (long long)(__builtin_amdgcn_s_memtime() - pr_t0);
    p.probe[2 * q + 1] = (long long)(__builtin_amdgcn_s_memrealtime() - pr_r0);
  }
#endif
  // ---- epilogue (as conv_igemm.hip, mode 0): lane holds channels crow .. crow+15 of pixel (j, fi)
  const int crow = n0 + wch * 64 + fq * 16;
  float bias[16];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const f32x4 b4 = *(const f32x4*)(g_bias + crow + i * 4);
    bias[i * 4 + 0] = b4[0]; bias[i * 4 + 1] = b4[1]; bias[i * 4 + 2] = b4[2]; bias[i * 4 + 3] = b4[3];
  }
  float wsc[SPLIT ? 16 : 1];
  if constexpr (SPLIT) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const f32x4 s4 = *(const f32x4*)(g_wscale + crow + i * 4);
      wsc[i * 4 + 0] = s4[0]; wsc[i * 4 + 1] = s4[1]; wsc[i * 4 + 2] = s4[2]; wsc[i * 4 + 3] = s4[3];
    }
  }
  if (!TRAIN && SPLIT && p.head_w) {
    // The fused 16-row head in the split-operand mode: relu(acc * scale + bias) is split into hi + lo exactly as the store would split it, and
    // the head is three MFMA products per 32-deep step (H_hi.B_hi, H_hi.B_lo, H_lo.B_hi) on the head's own hi / lo planes (row-scaled; the
    // inverse scale is applied when the four channel waves' partial sums are added).
    const half_t* hw = p.head_w + (long long)fi * 256 + wch * 64 + fq * 8;
    const half8 ha0 = *(const half8*)hw, ha1 = *(const half8*)(hw + 32);
    const half8 hl0 = *(const half8*)(hw + p.head_w_lo), hl1 = *(const half8*)(hw + p.head_w_lo + 32);
    f32x4 hacc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      half8 b0, b1, l0, l1;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float f = acc[i][j][r] * wsc[i * 4 + r] + bias[i * 4 + r];
          if (p.relu) f = f > 0.f ? f : 0.f;
          f = f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f);
          const half_t h = (half_t)f;
          const half_t l = (half_t)(f - (float)h);
          if (i < 2) { b0[i * 4 + r] = h; l0[i * 4 + r] = l; } else { b1[(i - 2) * 4 + r] = h; l1[(i - 2) * 4 + r] = l; }
        }
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha0, b0, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha1, b1, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha0, l0, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha1, l1, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(hl0, b0, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(hl1, b1, a, 0, 0, 0);
      hacc[j] = a;
    }
    asm volatile("s_barrier" ::: "memory");          // every wave is past its last fragment reads: the stage buffers are free
    float* part = (float*)smem;                      // [4 channel waves][2 WPX pixels][16] partial sums
#pragma unroll
    for (int j = 0; j < NJ; ++j) *(f32x4*)(part + ((wch * 2 * WPX + wpx * WPX + j * 16 + fi) * 16 + fq * 4)) = hacc[j];
    __syncthreads();
    const int px = tid >> 1, oh = (tid & 1) * 8;
    const int m = m0 + px;
    if (px < 2 * WPX && m < M) {
      f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s0 += *(const f32x4*)(part + ((w * 2 * WPX + px) * 16 + oh));
        s1 += *(const f32x4*)(part + ((w * 2 * WPX + px) * 16 + oh + 4));
      }
      s0 = s0 * *(const f32x4*)(p.head_scale + oh) + *(const f32x4*)(p.head_b + oh);
      s1 = s1 * *(const f32x4*)(p.head_scale + oh + 4) + *(const f32x4*)(p.head_b + oh + 4);
      const int x = m % Wo, t = m / Wo, y = t % Ho, n = t / Ho;
      float* op = g_head_out + ((long long)(n * Ho + y) * Wo + x) * 16 + oh;
      *(f32x4*)op = s0;
      *(f32x4*)(op + 4) = s1;
    }
    return;
  }
  if (!TRAIN && !SPLIT && p.head_w) {
    // Fused 16-row 1x1 head on top of this convolution (RPN: objectness + anchor deltas): the 256-channel output tile never leaves
    // the CU.  relu(acc + bias), rounded to fp16 exactly as the store would round it, is per lane 16 consecutive channels of a
    // pixel = the B operand of two 32-deep MFMA steps when the head's K columns are stored in the chaining order (weights.py
    // _perm_k64, see bneck_fused.hip).  Each wave reduces its 64 channels; the four channel waves are summed through LDS in a
    // fixed order, so the result is bitwise reproducible.
    const half_t* hw = p.head_w + (long long)fi * 256 + wch * 64 + fq * 8;
    const half8 ha0 = *(const half8*)hw, ha1 = *(const half8*)(hw + 32);
    f32x4 hacc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      half8 b0, b1;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float f = acc[i][j][r] + bias[i * 4 + r];
          if (p.relu) f = f > 0.f ? f : 0.f;
          f = f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f);
          if (i < 2) b0[i * 4 + r] = (half_t)f; else b1[(i - 2) * 4 + r] = (half_t)f;
        }
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha0, b0, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha1, b1, a, 0, 0, 0);
      hacc[j] = a;
    }
    asm volatile("s_barrier" ::: "memory");          // every wave is past its last fragment reads (lgkmcnt(0) in the last K step): the stage buffers are free
    float* part = (float*)smem;                      // [4 channel waves][2 WPX pixels][16] partial sums
#pragma unroll
    for (int j = 0; j < NJ; ++j) *(f32x4*)(part + ((wch * 2 * WPX + wpx * WPX + j * 16 + fi) * 16 + fq * 4)) = hacc[j];
    __syncthreads();
    const int px = tid >> 1, oh = (tid & 1) * 8;
    const int m = m0 + px;
    if (px < 2 * WPX && m < M) {
      f32x4 s0 = *(const f32x4*)(p.head_b + oh), s1 = *(const f32x4*)(p.head_b + oh + 4);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s0 += *(const f32x4*)(part + ((w * 2 * WPX + px) * 16 + oh));
        s1 += *(const f32x4*)(part + ((w * 2 * WPX + px) * 16 + oh + 4));
      }
      const int x = m % Wo, t = m / Wo, y = t % Ho, n = t / Ho;
      float* op = g_head_out + ((long long)(n * Ho + y) * Wo + x) * 16 + oh;
      *(f32x4*)op = s0;
      *(f32x4*)(op + 4) = s1;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int m = m0 + wpx * WPX + j * 16 + fi;
    if (m >= M) continue;
    const int x = m % Wo;
    const int t = m / Wo;
    const int y = t % Ho;
    const int n = t / Ho;
    const long long opix = (long long)(n * out_Hp + y + p.out_pad) * out_Wp + x + p.out_pad;
    float v[16];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (SPLIT) v[i * 4 + r] = acc[i][j][r] * wsc[i * 4 + r] + bias[i * 4 + r];
        else v[i * 4 + r] = acc[i][j][r] + bias[i * 4 + r];
      }
    if (p.res) {
      const half_t* rp = p.res + opix * p.out_Cs + crow;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const half4 h = *(const half4*)(rp + i * 4);
        if constexpr (SPLIT) {
          const half4 l = *(const half4*)(rp + p.res_lo + i * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i * 4 + r] += (float)h[r] + (float)l[r];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i * 4 + r] += (float)h[r];
        }
      }
    }
    if (p.up) {
      const long long upix = (long long)(n * p.up_Hp + (y >> 1) + p.up_pad) * p.up_Wp + (x >> 1) + p.up_pad;
      const half_t* up = p.up + upix * p.up_Cs + crow;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const half4 h = *(const half4*)(up + i * 4);
        if constexpr (SPLIT) {
          const half4 l = *(const half4*)(up + p.up_lo + i * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i * 4 + r] += (float)h[r] + (float)l[r];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i * 4 + r] += (float)h[r];
        }
      }
    }
    if (TRAIN && p.down) {        // backward of the nearest 2x upsample: add the 2x2 block of the finer gradient map
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {
        const long long dpix = (long long)(n * p.down_Hp + 2 * y + (dd >> 1) + p.down_pad) * p.down_Wp + 2 * x + (dd & 1) + p.down_pad;
        const half_t* dp = p.down + dpix * p.down_Cs + crow;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const half4 h = *(const half4*)(dp + i * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i * 4 + r] += (float)h[r];
        }
      }
    }
    if (TRAIN && p.res32) {
      const float* rp = p.res32 + opix * p.out_Cs + crow;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const f32x4 h = *(const f32x4*)(rp + i * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[i * 4 + r] += h[r];
      }
    }
    if (TRAIN && p.mask) {        // ReLU backward
      const half_t* mp = p.mask + opix * p.out_Cs + crow;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const half4 h = *(const half4*)(mp + i * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[i * 4 + r] = (float)h[r] > 0.f ? v[i * 4 + r] : 0.f;
      }
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    if (p.out_f32) {
      float* op = (float*)g_out + opix * p.out_Cs + crow;
#pragma unroll
      for (int i = 0; i < MI; ++i) *(f32x4*)(op + i * 4) = f32x4{v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]};
    } else {
      half_t* op = (half_t*)g_out + opix * p.out_Cs + crow;
#pragma unroll
      for (int i = 0; i < MI; i += 2) {
        half8 h, l;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float f = v[i * 4 + r];
          f = f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f);
          h[r] = (half_t)f;
          if constexpr (SPLIT) l[r] = (half_t)(f - (float)h[r]);
        }
        *(half8*)(op + i * 4) = h;
        if constexpr (SPLIT) *(half8*)(op + out_lo + i * 4) = l;
      }
    }
  }
#ifdef RS_SPLIT_PHASES
  if constexpr (SPLIT) {
    const unsigned long long ph_issued = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ph_drained = __builtin_amdgcn_s_memtime();
    if (p.probe && lane == 0) {
      long long* o = p.probe + ((long long)q * 8 + wave) * 8;
      o[6] = (long long)(ph_issued - ph_loop1); o[7] = (long long)(ph_drained - ph_issued);
    }
  }
#endif
}


// Grid: [full 256-pixel tiles in XCD-aware order][the p.tail_tiles last logical tiles again as two 128-pixel tiles each].  One
// workgroup per CU is resident (160 KB of LDS), so a launch runs in rounds of 256 tiles; when the last round would fill at most half the
// chip, its tiles are split so that it takes about half a round (launch_conv_deep: tail rule).
// NJF: 16-pixel blocks per wave of a whole tile = tile height / 32 pixels.  8 (256 pixels) is the production tile; 5 / 6 / 7 (160 / 192 /
// 224 pixels) exist for maps whose pixel count fills the 256 CUs badly in 256-pixel tiles (50 x 50 x 16 images = 40 000 pixels: 157 tiles
// = 0.61 of a round; 250 tiles of 160 pixels = 0.98 of one) -- launch_conv variants 15 / 16 / 17.
template <int DBG, bool TRAIN = false, int NJF = 8, bool SPLIT = false>
__global__ __launch_bounds__(NT) void conv_deep_kernel(const ConvParams p) {
  constexpr int BM = 32 * NJF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int M = p.M;
  if (p.m_count) {
    long long mc = (long long)(*p.m_count) * p.m_mul;
    if (mc < M) M = (int)mc;
  }
  const int tiles_n = p.Cout / BN;
  const int ntiles = p.nseg ? p.seg_tiles : tiles_n * ((M + BM - 1) / BM);
  const int nfull = ntiles - p.tail_tiles;
  const int q = blockIdx.x;
  if (q >= nfull + 2 * p.tail_tiles) return;
  // tensors and geometry of the map this tile belongs to (multi-map launch: p.seg[], selected with static indices -- all scalar)
  const half_t* g_in = p.in;
  const half_t* g_w = p.w;
  const float* g_bias = p.bias;
  void* g_out = p.out;
  float* g_head_out = p.head_out;
  long long in_lo = p.in_lo, out_lo = p.out_lo;
  const float* g_wscale = p.wscale;
  int Ho = p.Ho, Wo = p.Wo, in_Hp = p.in_Hp, in_Wp = p.in_Wp, out_Hp = p.out_Hp, out_Wp = p.out_Wp;
  int L, half = -1;
  if (q < nfull) {
    const int qn = nfull >> 3, r = nfull & 7, x = q & 7;      // XCD-aware order, see conv_igemm.hip
    L = (x < r ? x * (qn + 1) : r * (qn + 1) + (x - r) * qn) + (q >> 3);
  } else {
    L = nfull + ((q - nfull) >> 1);
    half = (q - nfull) & 1;
  }
  if (p.nseg) {
    int t0 = 0;
#pragma unroll
    for (int i = 0; i < RS_MAX_SEGS; ++i)
      if (i < p.nseg && L >= p.seg[i].tile0) {
        g_in = p.seg[i].in; g_w = p.seg[i].w; g_bias = p.seg[i].bias; g_out = p.seg[i].out; g_head_out = p.seg[i].head_out;
        Ho = p.seg[i].Ho; Wo = p.seg[i].Wo; in_Hp = p.seg[i].in_Hp; in_Wp = p.seg[i].in_Wp;
        out_Hp = p.seg[i].out_Hp; out_Wp = p.seg[i].out_Wp; M = p.seg[i].M; t0 = p.seg[i].tile0;
        if constexpr (SPLIT) { in_lo = p.seg[i].in_lo; out_lo = p.seg[i].out_lo; g_wscale = p.seg[i].wscale; }
      }
    L -= t0;
  }
  const int n0 = (L % tiles_n) * BN;
  const int m0 = (L / tiles_n) * BM;
  if (half < 0 || NJF != 8) {
    conv_deep_tile<DBG, TRAIN, NJF, SPLIT>(p, smem, g_in, g_w, g_bias, g_out, g_head_out, Ho, Wo, in_Hp, in_Wp, out_Hp, out_Wp, M, m0, n0, q, in_lo, out_lo, g_wscale);
  } else {
    if (m0 + half * (BM / 2) >= M) return;                     // second half of a partial last tile: nothing to do (workgroup-uniform)
    conv_deep_tile<DBG, TRAIN, NJF == 8 ? 4 : NJF, SPLIT>(p, smem, g_in, g_w, g_bias, g_out, g_head_out, Ho, Wo, in_Hp, in_Wp, out_Hp, out_Wp, M, m0 + half * (BM / 2), n0, q, in_lo, out_lo, g_wscale);
  }
}

}  // namespace

// Tail rule.  One workgroup per CU is resident, all tiles of a launch take the same time, so T tiles run in ceil(T / CUs) rounds.
// If the last round holds r <= CUs / 2 tiles, those r tiles are split into 2 r tiles of 128 pixels, which fit one round of about
// half the length (625 tiles on 256 CUs: 3 rounds -> ~2.55).  Returns r (0: no split).
int rs_device_cu_count() {
  // per device ordinal: a process may drive several (different) devices; without a device (the dispatch rule queried on a CPU-only
  // box: rs_op_conv_variant, tests/golden/conv_variants.json) the MI355X's 256
  static int ncu[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!ncu[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    ncu[dev] = n;
  }
  return ncu[dev];
}

static int deep_tail_tiles(long long tiles) {
  const int ncu = rs_device_cu_count();
  if (!rs_debug().deep_tail) return 0;
  const int r = (int)(tiles % ncu);
  return (r > 0 && 2 * r <= ncu) ? r : 0;
}

// The fragment reads of these kernels are inline asm whose results the compiler believes ready at once; the hand-placed s_waitcnt cover the
// REGISTERS.  A register the allocator spills right after such a read is stored before the data has landed (seen in the ISA of a variant of the split
// loop that the compiler peeled: scratch_store of a fragment one instruction after its ds_read_b128), so a build of these kernels that uses scratch is
// refused at first launch instead of computing garbage.
static int no_scratch(const void* kernel, const char* what) {
  hipFuncAttributes a;
  RS_HIP(hipFuncGetAttributes(&a, kernel));
  RS_CHECK(a.localSizeBytes == 0, RS_ERR_UNSUPPORTED, "%s was built with %zu bytes of scratch per thread: its asm fragment reads must not be spilled (rebuild with the documented toolchain)",
           what, (size_t)a.localSizeBytes);
  return RS_OK;
}

// Inference kernel with tiles of 32 * NJF pixels (variants 15 / 16 / 17; no training epilogue, no split last round)
template <int NJF>
static int launch_deep_nj(ConvParams& p, hipStream_t stream) {
  static bool done = false;
  if (!done) {
    RS_HIP(hipFuncSetAttribute((const void*)conv_deep_kernel<0, false, NJF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    if constexpr (NJF >= 5) RS_HIP(hipFuncSetAttribute((const void*)conv_deep_kernel<0, false, NJF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    { int rc = no_scratch((const void*)conv_deep_kernel<0, false, NJF>, "conv_deep_kernel"); if (rc) return rc; }
    if constexpr (NJF >= 5) { int rc = no_scratch((const void*)conv_deep_kernel<0, false, NJF, true>, "conv_deep_kernel (split operands)"); if (rc) return rc; }
    done = true;
  }
  const long long nblk = (long long)(p.Cout / BN) * cdiv(p.M, 32 * NJF);
  RS_CHECK(nblk < (1ll << 30), RS_ERR_ARG, "conv_deep: grid too large");
  p.tail_tiles = 0;
  if (p.split) {
    if constexpr (NJF >= 5) hipLaunchKernelGGL((conv_deep_kernel<0, false, NJF, true>), dim3((unsigned)nblk), dim3(NT), LDS_BYTES, stream, p);
    else RS_CHECK(false, RS_ERR_UNSUPPORTED, "conv_deep: the split-operand mode has the 160 .. 256-pixel tiles only");
    RS_HIP(hipGetLastError());
    return RS_OK;
  }
  hipLaunchKernelGGL((conv_deep_kernel<0, false, NJF>), dim3((unsigned)nblk), dim3(NT), LDS_BYTES, stream, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

// Requirements: mode 0, single K source, Cin % 64 == 0, Cout % 256 == 0.  tile_px: 256 (variant 12), 160 / 192 / 224 (15 / 16 / 17) or
// 64 / 96 / 128 (18 / 19 / 20).
int launch_conv_deep(const ConvParams& p0, hipStream_t stream, int tile_px) {
  ConvParams p = p0;
  p.nseg = 0;
  RS_CHECK(p.mode == 0 && !p.in2 && p.Cin % 64 == 0 && p.Cout % BN == 0 && p.M > 0, RS_ERR_ARG,
           "conv_deep: unsupported shape (mode %d, Cin %d, Cout %d)", p.mode, p.Cin, p.Cout);
  RS_CHECK(!p.head_w || (p.Cout == BN && p.head_b && p.head_out && !p.res && !p.up && !p.down && !p.res32 && !p.mask), RS_ERR_ARG,
           "conv_deep: the fused head needs Cout == 256, its bias and output, and no other epilogue option");
  RS_CHECK(p.out_stride <= 1, RS_ERR_UNSUPPORTED, "conv_deep: no strided scatter");
  RS_CHECK(!p.split || (p.wscale && (!p.head_w || p.head_scale) && !p.down && !p.res32 && !p.mask), RS_ERR_UNSUPPORTED, "conv_deep: the split-operand mode needs the row scales (the fused head's too) and has no training epilogue");
  if (p.split) {       // the split loop addresses its LDS-DMA pieces as SGPR base + 32-bit byte offset: both planes within 4 GB of the tensor's base
    const long long in_px = (long long)cdiv(p.M, p.Ho * p.Wo) * p.in_Hp * p.in_Wp;
    RS_CHECK((p.in_lo + in_px * p.in_Cs) * 2 < (1ll << 32) && (p.w_lo + (long long)p.Cout * p.Kpad) * 2 < (1ll << 32), RS_ERR_UNSUPPORTED,
             "conv_deep: split-operand tensor of more than 4 GB (input %lld elements per plane)", in_px * p.in_Cs);
  }
  if (tile_px != 256) {
    RS_CHECK(tile_px >= 64 && tile_px <= 224 && tile_px % 32 == 0 && !p.down && !p.res32 && !p.mask, RS_ERR_ARG,
             "conv_deep: tile height %d (256, or 64 .. 224 in steps of 32 without training epilogue options)", tile_px);
    switch (tile_px / 32) {
      case 2: return launch_deep_nj<2>(p, stream);
      case 3: return launch_deep_nj<3>(p, stream);
      case 4: return launch_deep_nj<4>(p, stream);
      case 5: return launch_deep_nj<5>(p, stream);
      case 6: return launch_deep_nj<6>(p, stream);
      default: return launch_deep_nj<7>(p, stream);
    }
  }
  constexpr int BM = 256;
  static bool done = false;
  if (!done) {
    RS_HIP(hipFuncSetAttribute((const void*)conv_deep_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RS_HIP(hipFuncSetAttribute((const void*)conv_deep_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RS_HIP(hipFuncSetAttribute((const void*)conv_deep_kernel<0, false, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    { int rc = no_scratch((const void*)conv_deep_kernel<0>, "conv_deep_kernel"); if (rc) return rc; }
    { int rc = no_scratch((const void*)conv_deep_kernel<0, true>, "conv_deep_kernel (training epilogue)"); if (rc) return rc; }
    { int rc = no_scratch((const void*)conv_deep_kernel<0, false, 8, true>, "conv_deep_kernel (split operands)"); if (rc) return rc; }
#ifdef RS_DEEP_CEILING
    RS_HIP(hipFuncSetAttribute((const void*)conv_deep_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RS_HIP(hipFuncSetAttribute((const void*)conv_deep_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RS_HIP(hipFuncSetAttribute((const void*)conv_deep_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RS_HIP(hipFuncSetAttribute((const void*)conv_deep_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RS_HIP(hipFuncSetAttribute((const void*)conv_deep_kernel<17>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
#endif
    done = true;
  }
  // Ceiling experiments (built only with -DRS_DEEP_CEILING; results are WRONG by construction): DBG bit 0 = no global
  // traffic, bit 1 = every step reads LDS stage 0, 4 = no weight traffic only, 8 = no activation traffic only, 16 = no
  // fragment reads.  Measured on fpn_output2 (755 GFLOP, tools/ubench/conv_time.py, one box): full kernel 0.635 ms
  // (1.19 PFLOP/s); without weight loads 0.578, without activation loads 0.562, without any global traffic 0.532
  // (1.42 PFLOP/s); without global traffic AND without the ds_read_b128 fragment reads 0.422 (1.79 PFLOP/s).  So of the
  // kernel's time 66 % is the MFMA loop with its barriers and epilogue, 17 % the LDS fragment reads (192 KB per K step
  // and workgroup: their issue and VGPR write-back are not free next to the MFMAs even though the LDS array is only
  // ~40 % busy), 16 % global traffic, split about evenly between the weight pieces (issued ONE step ahead, from L2) and
  // the activation pieces (two steps ahead, from HBM / MALL): 160 KB of LDS leave no room for a deeper prefetch at
  // this tile size.
#ifdef RS_DEEP_CEILING
  const int dbg = rs_debug().deep_dbg;
#else
  constexpr int dbg = 0;
#endif
  long long nblk = (long long)(p.Cout / BN) * cdiv(p.M, BM);
  RS_CHECK(nblk < (1ll << 30), RS_ERR_ARG, "conv_deep: grid too large");
  p.tail_tiles = (p.m_count || dbg) ? 0 : deep_tail_tiles(nblk);     // a device-side row count changes the tile count behind the host's back
  nblk += p.tail_tiles;
#ifdef RS_DEEP_CEILING
  if (dbg == 1) { hipLaunchKernelGGL(conv_deep_kernel<1>, dim3((unsigned)nblk), dim3(NT), LDS_BYTES, stream, p); return RS_OK; }
  if (dbg == 3) { hipLaunchKernelGGL(conv_deep_kernel<3>, dim3((unsigned)nblk), dim3(NT), LDS_BYTES, stream, p); return RS_OK; }
  if (dbg == 4) { hipLaunchKernelGGL(conv_deep_kernel<4>, dim3((unsigned)nblk), dim3(NT), LDS_BYTES, stream, p); return RS_OK; }
  if (dbg == 8) { hipLaunchKernelGGL(conv_deep_kernel<8>, dim3((unsigned)nblk), dim3(NT), LDS_BYTES, stream, p); return RS_OK; }
  if (dbg == 17) { hipLaunchKernelGGL(conv_deep_kernel<17>, dim3((unsigned)nblk), dim3(NT), LDS_BYTES, stream, p); return RS_OK; }
#endif
  (void)dbg;
  if (p.split) hipLaunchKernelGGL((conv_deep_kernel<0, false, 8, true>), dim3((unsigned)nblk), dim3(NT), LDS_BYTES, stream, p);
  else if (p.down || p.res32 || p.mask) hipLaunchKernelGGL((conv_deep_kernel<0, true>), dim3((unsigned)nblk), dim3(NT), LDS_BYTES, stream, p);
  else hipLaunchKernelGGL(conv_deep_kernel<0>, dim3((unsigned)nblk), dim3(NT), LDS_BYTES, stream, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

// The same convolution (kernel size, stride 1, Cin, Cout, channel pitches, halos, ReLU) over up to RS_MAX_SEGS maps with their own
// geometry, tensors and weights, as one grid: tiles of all maps share the 256 CUs, so the small maps of a feature pyramid fill the
// partial last round of the large ones instead of each paying a round (and a launch) of their own.
int launch_conv_deep_multi(const ConvParams& common, const ConvSeg* segs, const int* m_per_image, int nseg, int images, hipStream_t stream) {
  RS_CHECK(nseg >= 1 && nseg <= RS_MAX_SEGS && images >= 1, RS_ERR_ARG, "conv_deep_multi: %d maps", nseg);
  RS_CHECK(common.mode == 0 && !common.in2 && common.Cin % 64 == 0 && common.Cout % BN == 0 && !common.res && !common.up && !common.m_count &&
               !common.down && !common.res32 && !common.mask && common.out_stride <= 1,
           RS_ERR_ARG, "conv_deep_multi: unsupported shape (mode %d, Cin %d, Cout %d) or epilogue option", common.mode, common.Cin, common.Cout);
  ConvParams p = common;
  p.nseg = nseg;
  long long tiles = 0;
  for (int i = 0; i < nseg; ++i) {
    RS_CHECK(segs[i].in && segs[i].w && segs[i].bias && segs[i].out && m_per_image[i] > 0, RS_ERR_ARG, "conv_deep_multi: map %d incomplete", i);
    RS_CHECK(!common.split || segs[i].wscale, RS_ERR_ARG, "conv_deep_multi: map %d has no row scales (split-operand mode)", i);
    RS_CHECK(!common.split || ((segs[i].in_lo + (long long)images * segs[i].in_Hp * segs[i].in_Wp * common.in_Cs) * 2 < (1ll << 32) &&
                               (common.w_lo + (long long)common.Cout * common.Kpad) * 2 < (1ll << 32)),
             RS_ERR_UNSUPPORTED, "conv_deep_multi: map %d: split-operand tensor of more than 4 GB", i);
    p.seg[i] = segs[i];
    p.seg[i].M = images * m_per_image[i];
    p.seg[i].tile0 = (int)tiles;
    tiles += (long long)(p.Cout / BN) * cdiv(p.seg[i].M, 256);
  }
  RS_CHECK(tiles < (1ll << 30), RS_ERR_ARG, "conv_deep_multi: grid too large");
  if (p.head_w) {
    RS_CHECK(p.Cout == BN && p.head_b && !p.out_f32, RS_ERR_ARG, "conv_deep_multi: the fused head needs Cout == 256 and its bias");
    for (int i = 0; i < nseg; ++i) RS_CHECK(segs[i].head_out, RS_ERR_ARG, "conv_deep_multi: map %d has no head output", i);
  }
  p.seg_tiles = (int)tiles;
  p.tail_tiles = deep_tail_tiles(tiles);
  tiles += p.tail_tiles;
  p.in = segs[0].in; p.w = segs[0].w; p.bias = segs[0].bias; p.out = segs[0].out; p.M = p.seg[0].M;
  static bool done = false;
  if (!done) {
    RS_HIP(hipFuncSetAttribute((const void*)conv_deep_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RS_HIP(hipFuncSetAttribute((const void*)conv_deep_kernel<0, false, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    { int rc = no_scratch((const void*)conv_deep_kernel<0>, "conv_deep_kernel"); if (rc) return rc; }
    { int rc = no_scratch((const void*)conv_deep_kernel<0, false, 8, true>, "conv_deep_kernel (split operands)"); if (rc) return rc; }
    done = true;
  }
  if (p.split) {
    RS_CHECK(!p.head_w || p.head_scale, RS_ERR_ARG, "conv_deep_multi: the fused head of the split-operand mode needs its row scales");
    hipLaunchKernelGGL((conv_deep_kernel<0, false, 8, true>), dim3((unsigned)tiles), dim3(NT), LDS_BYTES, stream, p);
    RS_HIP(hipGetLastError());
    return RS_OK;
  }
  hipLaunchKernelGGL(conv_deep_kernel<0>, dim3((unsigned)tiles), dim3(NT), LDS_BYTES, stream, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
