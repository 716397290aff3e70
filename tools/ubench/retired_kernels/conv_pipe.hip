// EXPERIMENTAL (not on the default path; RS_CONV_BIG_TILE=5 or rs_op_conv2d variant 6): software-pipelined
// 256x256 implicit-GEMM conv for the deep-K layers (3x3 256->256 of FPN/RPN/mask head, res4 conv2, fc1/fc2).
// MEASURED on MI355X: correct (parity tests pass) but 5-25 % SLOWER than the two-buffer conv_igemm<2,4,4,8>
// (fpn_output2 0.765 vs 0.702 ms, fc1 0.446 vs 0.355 ms): with all 8 waves behind one barrier per 32-deep step,
// the LDS-DMA/ds_read issue time of every step is exposed twice as often.  Kept as the starting point for a
// staggered two-wave-group schedule (cdna guide, 8-phase template), which is what this structure needs.
//
// Same math, operand roles, LDS-DMA staging, epilogue and bank-conflict-free XOR layout idea as
// conv_igemm.hip (weights = MFMA A operand, activations = B operand, v_mfma_f32_16x16x32_f16).  What changes
// is the schedule.  In the two-buffer kernel all 8 waves of the single resident workgroup hit the barrier,
// then all read their fragments from LDS at once (96 KB per half step against 256 B/clk), then all issue
// MFMAs: rocprofv3 shows the waves parked 37 % of the time (SQ_WAIT_ANY).  Here:
//   * K step = 32 (64-byte LDS rows), FOUR 32 KB LDS stages, three of them in flight: LDS-DMA of step t+3 is
//     issued while step t computes, and its arrival is awaited with a COUNTED s_waitcnt vmcnt(4) (never 0 in
//     the loop) followed by a raw s_barrier (a __syncthreads() fence would drain the queue);
//   * fragments are double buffered in registers: the ds_reads of step t+1 are issued (after the barrier that
//     publishes stage t+1) BEFORE the 32 MFMAs of step t, so LDS latency and bandwidth hide under the MFMAs.
// Ordering argument (cdna guide, "read a staged buffer one phase after the wait that retires it"): a wave
// passes barrier t only after its own pieces of stage t+1 landed (vmcnt) and after it issued MFMAs(t-1), i.e.
// after the fragments of step t-1 reached its registers; so after barrier t every piece of stage t+1 is
// visible to every wave, and nobody still reads stage (t-1)&3 = (t+3)&3, the one refilled next.
#include "common.h"

namespace {

constexpr int BM = 256, BN = 256, NT = 512, MI = 4, NJ = 8, WCH = 4;
constexpr int ROWB = 64;                       // bytes per LDS row (32 fp16 of K)
constexpr int STAGE = (BM + BN) * ROWB;        // 32 KB
constexpr int NSTAGE = 4;
constexpr int LDS_BYTES = NSTAGE * STAGE;      // 128 KB

__device__ __forceinline__ void glds16(const half_t* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct Frags {
  half8 w[MI];
  half8 x[NJ];
};

__global__ __launch_bounds__(NT) void conv_pipe_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpx = wave / WCH, wch = wave % WCH;

  int M = p.M;
  if (p.m_count) {
    long long mc = (long long)(*p.m_count) * p.m_mul;
    if (mc < M) M = (int)mc;
  }
  const int tiles_n = p.Cout / BN;
  const int ntiles = tiles_n * ((M + BM - 1) / BM);
  const int q = blockIdx.x;
  if (q >= ntiles) return;
  int m0, n0;
  {
    const int qn = ntiles >> 3, r = ntiles & 7, x = q & 7;      // XCD-aware order, see conv_igemm.hip
    const int L = (x < r ? x * (qn + 1) : r * (qn + 1) + (x - r) * qn) + (q >> 3);
    n0 = (L % tiles_n) * BN;
    m0 = (L / tiles_n) * BM;
  }

  // ---- staging: a wave-instruction writes 16 rows x 64 B; lane -> (row = lane>>2, 16-byte slot = lane&3).
  // LDS slot s of row r holds data chunk s ^ key(r); key = 3 * bit3(r) for activations, 3 * bit5(r) for
  // weights (their fragment rows are permuted, see w_row below): both make every ds_read_b128 group hit 16
  // distinct 16-byte slots of the 256-byte bank row.
  const int lrow = lane >> 2, lslot = lane & 3;
  const half_t* aptr[2];
  const half_t* wptr[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int r = ps * 128 + wave * 16 + lrow;
    int m = m0 + r;
    if (m >= M) m = M - 1;
    const int x = m % p.Wo;
    const int t = m / p.Wo;
    const int y = t % p.Ho;
    const int n = t / p.Ho;
    const long long base =
        ((long long)(n * p.in_Hp + y * p.stride + p.in_off) * p.in_Wp + x * p.stride + p.in_off) * p.in_Cs;
    aptr[ps] = p.in + base + (lslot ^ (((r >> 3) & 1) * 3)) * 8;
    wptr[ps] = p.w + (long long)(n0 + r) * p.Kpad + (lslot ^ (((r >> 5) & 1) * 3)) * 8;
  }
  const int nk = p.KH * p.KW * (p.Cin >> 5);     // K steps of 32; even and >= 4 (checked by the launcher)

  int kh = 0, kw = 0, c0 = 0;
  auto next_off = [&]() {
    const int off = (kh * p.in_Wp + kw) * p.in_Cs + c0;
    c0 += 32;
    if (c0 >= p.Cin) {
      c0 = 0;
      if (++kw == p.KW) { kw = 0; ++kh; }
    }
    return off;
  };
  auto stage = [&](int t) {                       // 4 LDS-DMA pieces per lane per stage (the vmcnt unit)
    char* abase = smem + (t & 3) * STAGE;
    char* wbase = abase + BM * ROWB;
    const int a_off = next_off();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      glds16(aptr[ps] + a_off, abase + (ps * 128 + wave * 16) * ROWB);
      glds16(wptr[ps] + t * 32, wbase + (ps * 128 + wave * 16) * ROWB);
    }
  };

  // ---- fragment addresses
  const int fi = lane & 15, fq = lane >> 4;
  const int fkey = ((fi >> 3) & 1) * 3;
  const int foff = (fq ^ fkey) * 16;
  const int w_base = BM * ROWB + (wch * 64 + (fi >> 2) * 16 + (fi & 3)) * ROWB + foff;   // + i*4 rows per tile i
  const int x_base = (wpx * 128 + fi) * ROWB + foff;                                     // + j*16 rows per tile j
  // Fragment reads are inline-asm ds_read_b128: hipcc's waitcnt pass cannot see them, so it does not put its
  // (conservative) s_waitcnt lgkmcnt(0) between these reads and the MFMAs that follow -- those MFMAs use the
  // OTHER register set, fetched one step earlier and retired by the explicit wait at the end of that step.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned w_addr = lds0 + (unsigned)w_base, x_addr = lds0 + (unsigned)x_base;
#define RS_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  auto load_frags = [&](Frags& f, int t) {
    const unsigned so = (unsigned)((t & 3) * STAGE);
    const unsigned wa = w_addr + so, xa = x_addr + so;
    RS_DSR(f.w[0], wa, 0 * 4 * ROWB);
    RS_DSR(f.w[1], wa, 1 * 4 * ROWB);
    RS_DSR(f.w[2], wa, 2 * 4 * ROWB);
    RS_DSR(f.w[3], wa, 3 * 4 * ROWB);
    RS_DSR(f.x[0], xa, 0 * 16 * ROWB);
    RS_DSR(f.x[1], xa, 1 * 16 * ROWB);
    RS_DSR(f.x[2], xa, 2 * 16 * ROWB);
    RS_DSR(f.x[3], xa, 3 * 16 * ROWB);
    RS_DSR(f.x[4], xa, 4 * 16 * ROWB);
    RS_DSR(f.x[5], xa, 5 * 16 * ROWB);
    RS_DSR(f.x[6], xa, 6 * 16 * ROWB);
    RS_DSR(f.x[7], xa, 7 * 16 * ROWB);
  };
  auto retire_frags = [&]() {
    // MFMAs are register-only, so a "memory" clobber does not order them against the asm wait: fence the
    // scheduler on both sides (this step's MFMAs stay above the wait, the next step's below it).
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mfmas = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[i], f.x[j], acc[i][j], 0, 0, 0);
  };

  // One pipeline step: publish stage t+1, refill stage t+3, fetch the fragments of step t+1, then compute
  // step t from the fragments fetched one step earlier.
  auto step = [&](int t, const Frags& cur, Frags& nxt) {
    if (t + 1 < nk) {
      if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      if (t + 3 < nk) stage(t + 3);
      load_frags(nxt, t + 1);
    }
    mfmas(cur);
    retire_frags();     // `nxt` was requested ~32 MFMAs ago: this wait is free, and it is the only one that guards `nxt`
  };

  stage(0);
  stage(1);
  stage(2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  Frags f0, f1;
  load_frags(f0, 0);
  retire_frags();
  for (int t = 0; t < nk; t += 2) {
    step(t, f0, f1);
    step(t + 1, f1, f0);
  }

  // ---- epilogue (as conv_igemm.hip, mode 0): lane holds channels cb .. cb+15 of pixel (j, fi)
  const int crow = n0 + wch * 64 + fq * 16;
  float bias[16];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const f32x4 b4 = *(const f32x4*)(p.bias + crow + i * 4);
    bias[i * 4 + 0] = b4[0]; bias[i * 4 + 1] = b4[1]; bias[i * 4 + 2] = b4[2]; bias[i * 4 + 3] = b4[3];
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int m = m0 + wpx * 128 + j * 16 + fi;
    if (m >= M) continue;
    const int x = m % p.Wo;
    const int t = m / p.Wo;
    const int y = t % p.Ho;
    const int n = t / p.Ho;
    const long long opix = (long long)(n * p.out_Hp + y + p.out_pad) * p.out_Wp + x + p.out_pad;
    float v[16];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r] + bias[i * 4 + r];
    if (p.res) {
      const half_t* rp = p.res + opix * p.out_Cs + crow;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const half4 h = *(const half4*)(rp + i * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[i * 4 + r] += (float)h[r];
      }
    }
    if (p.up) {
      const long long upix = (long long)(n * p.up_Hp + (y >> 1) + p.up_pad) * p.up_Wp + (x >> 1) + p.up_pad;
      const half_t* up = p.up + upix * p.up_Cs + crow;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const half4 h = *(const half4*)(up + i * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[i * 4 + r] += (float)h[r];
      }
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    if (p.out_f32) {
      float* op = (float*)p.out + opix * p.out_Cs + crow;
#pragma unroll
      for (int i = 0; i < MI; ++i) *(f32x4*)(op + i * 4) = f32x4{v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]};
    } else {
      half_t* op = (half_t*)p.out + opix * p.out_Cs + crow;
#pragma unroll
      for (int i = 0; i < MI; i += 2) {
        half8 h;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float f = v[i * 4 + r];
          f = f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f);
          h[r] = (half_t)f;
        }
        *(half8*)(op + i * 4) = h;
      }
    }
  }
}

}  // namespace

// Requirements: mode 0, Cin % 64 == 0 (=> an even number of 32-deep K steps), Cout % 256 == 0, >= 4 K steps.
int launch_conv_pipe(const ConvParams& p, hipStream_t stream) {
  RS_CHECK(p.mode == 0 && p.Cin % 64 == 0 && p.Cout % BN == 0 && p.KH * p.KW * (p.Cin >> 5) >= 4 && p.M > 0, RS_ERR_ARG,
           "conv_pipe: unsupported shape (mode %d, Cin %d, Cout %d)", p.mode, p.Cin, p.Cout);
  static bool done = false;
  if (!done) {
    RS_HIP(hipFuncSetAttribute((const void*)conv_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    done = true;
  }
  const long long nblk = (long long)(p.Cout / BN) * cdiv(p.M, BM);
  RS_CHECK(nblk < (1ll << 31), RS_ERR_ARG, "conv_pipe: grid too large");
  hipLaunchKernelGGL(conv_pipe_kernel, dim3((unsigned)nblk), dim3(NT), LDS_BYTES, stream, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
