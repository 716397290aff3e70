// Staggered 256x256 implicit-GEMM conv for the deep-K layers (3x3 256->256 of FPN/RPN/mask head, fc1/fc2).
//
// Same math, operand roles (weights = MFMA A operand, activations = B operand, v_mfma_f32_16x16x32_f16),
// LDS-DMA staging, 128-byte XOR-swizzled LDS rows, two 64 KB K-step buffers and epilogue as the 256x256
// variant of conv_igemm.hip.  What changes is the schedule.  There, the 8 waves of the (single resident)
// workgroup run in lockstep behind one barrier per K step: they all read fragments from LDS at the same time
// and then all issue MFMAs at the same time, so each SIMD's matrix pipe idles while its two waves read
// (rocprofv3: waves parked 37 % of their lifetime, ~1.05 PFLOP/s).  Here the two waves of every SIMD belong
// to different GROUPS (A = waves 0-3, B = waves 4-7) that run the same four-phase program one phase apart:
//
//      interval      4t        4t+1       4t+2       4t+3      4t+4 ...
//      group A    reads(t,0)  MFMA(t,0)  reads(t,1)  MFMA(t,1)  reads(t+1,0)
//      group B    MFMA(t-1,1) reads(t,0) MFMA(t,0)   reads(t,1) MFMA(t,1)
//
// (one workgroup barrier per interval; B takes one extra barrier up front, A one at the end).  In every
// interval exactly one wave per SIMD issues 32 MFMAs while its partner fetches fragments, so the LDS phase
// hides under the partner's matrix phase.  Hazards: buffer (t+1)&1 is refilled (LDS-DMA, all waves, during
// interval 4t) only after its last reader (B, reads(t-1,1), interval 4t-1) passed the barrier; every wave
// waits for its own pieces of step t+1 (vmcnt(0)) before the barrier that ends interval 4t+3, and the
// first reader of step t+1 (A) starts in interval 4t+4.
#include "common.h"

namespace {

constexpr int BM = 256, BN = 256, NT = 512, MI = 4, NJ = 8, WCH = 4;
constexpr int STAGE = (BM + BN) * 128;         // 64 KB
constexpr int LDS_BYTES = 2 * STAGE;           // 128 KB

__device__ __forceinline__ void glds16(const half_t* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void wg_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");      // raw barrier: no vmcnt(0) drain (LDS-DMA stays in flight across it)
  __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(NT) void conv_stag_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpx = wave / WCH, wch = wave % WCH;
  const bool grpB = wave >= 4;                  // waves w and w+4 share a SIMD (dispatch order 0..3, 4..7)

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

  // ---- staging pointers (identical to conv_igemm<2,4,4,8>): 4 passes of 64 rows for each operand
  const int lrow = lane >> 3, lchk = lane & 7;
  const half_t* aptr[4];
  const half_t* wptr[4];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    int m = m0 + ps * 64 + wave * 8 + lrow;
    if (m >= M) m = M - 1;
    const int x = m % p.Wo;
    const int t = m / p.Wo;
    const int y = t % p.Ho;
    const int n = t / p.Ho;
    const long long base =
        ((long long)(n * p.in_Hp + y * p.stride + p.in_off) * p.in_Wp + x * p.stride + p.in_off) * p.in_Cs;
    aptr[ps] = p.in + base + (lchk ^ lrow) * 8;
    const int row = ps * 64 + wave * 8 + lrow;
    const int key = (row & 3) | (((row >> 4) & 1) << 2);
    wptr[ps] = p.w + (long long)(n0 + row) * p.Kpad + (lchk ^ key) * 8;
  }
  const int nk = p.KH * p.KW * (p.Cin >> 6);

  int kh = 0, kw = 0, c0 = 0;
  auto next_off = [&]() {
    const int off = (kh * p.in_Wp + kw) * p.in_Cs + c0;
    c0 += 64;
    if (c0 >= p.Cin) {
      c0 = 0;
      if (++kw == p.KW) { kw = 0; ++kh; }
    }
    return off;
  };
  auto stage = [&](int t) {
    char* abase = smem + (t & 1) * STAGE;
    char* wbase = abase + BM * 128;
    const int a_off = next_off();
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      glds16(aptr[ps] + a_off, abase + (ps * 64 + wave * 8) * 128);
      glds16(wptr[ps] + t * 64, wbase + (ps * 64 + wave * 8) * 128);
    }
  };

  const int fi = lane & 15, fq = lane >> 4, fkey = lane & 7;
  int w_off[MI], x_off[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) w_off[i] = BM * 128 + (wch * 64 + (fi >> 2) * 16 + i * 4 + (fi & 3)) * 128;
#pragma unroll
  for (int j = 0; j < NJ; ++j) x_off[j] = (wpx * 128 + j * 16 + fi) * 128;
  const int c_off[2] = {(fq ^ fkey) * 16, ((4 + fq) ^ fkey) * 16};

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  half8 wf[MI], xf[NJ];
  auto reads = [&](int t, int kk) {
    const char* sb = smem + (t & 1) * STAGE + c_off[kk];
#pragma unroll
    for (int i = 0; i < MI; ++i) wf[i] = *(const half8*)(sb + w_off[i]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) xf[j] = *(const half8*)(sb + x_off[j]);
  };
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // prologue: steps 0 and 1 in flight, step 0 published
  stage(0);
  if (nk > 1) stage(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg_barrier();
  if (grpB) wg_barrier();                        // B runs one interval behind A
  for (int t = 0; t < nk; ++t) {
    // phase 0: (A) refill the buffer step t-1 used; fetch the first half-step's fragments
    if (!grpB && t >= 1 && t + 1 < nk) stage(t + 1);
    reads(t, 0);
    wg_barrier();
    // phase 1
    mfmas();
    wg_barrier();
    // phase 2: second half-step's fragments; (B) its pieces of step t+1 must have landed before the barrier
    reads(t, 1);
    if (grpB && t + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    // phase 3: (B) refill the buffer of step t (both groups finished reading it); (A) wait for step t+1
    if (grpB && t + 2 < nk) stage(t + 2);
    mfmas();
    if (!grpB && t + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
  }
  if (!grpB) wg_barrier();

  // ---- epilogue (as conv_igemm.hip, mode 0): lane holds channels crow .. crow+15 of pixel (j, fi)
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

// Requirements: mode 0, Cin % 64 == 0, Cout % 256 == 0.
int launch_conv_stag(const ConvParams& p, hipStream_t stream) {
  RS_CHECK(p.mode == 0 && p.Cin % 64 == 0 && p.Cout % BN == 0 && p.M > 0, RS_ERR_ARG,
           "conv_stag: unsupported shape (mode %d, Cin %d, Cout %d)", p.mode, p.Cin, p.Cout);
  static bool done = false;
  if (!done) {
    RS_HIP(hipFuncSetAttribute((const void*)conv_stag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    done = true;
  }
  const long long nblk = (long long)(p.Cout / BN) * cdiv(p.M, BM);
  RS_CHECK(nblk < (1ll << 31), RS_ERR_ARG, "conv_stag: grid too large");
  hipLaunchKernelGGL(conv_stag_kernel, dim3((unsigned)nblk), dim3(NT), LDS_BYTES, stream, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
