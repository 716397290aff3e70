// Fused tail of an identity-shortcut bottleneck block, for the HBM-bound res2 stage (bottleneck width 64):
//
//     t2  = relu(conv3x3(t1, W2) + b2)             64 -> 64      [EXT d2: modeling/backbone/resnet.py BottleneckBlock.conv2]
//     out = relu(W3 . t2 + b3 + x)                 64 -> 256     conv3 + shortcut + ReLU
//     t1n = relu(W1n . out + b1n)                  256 -> 64     the NEXT block's conv1 (optional)
//
// in ONE launch.  Layer by layer this chain moves, per block of the batch-16 forward (640 000 pixels), t1 82 MB in, t2 82 MB
// out + 82 MB in, x 328 MB in, out 328 MB out, and for the next conv1 out 328 MB in + t1n 82 MB out = 1 312 MB at the
// 4.3-4.6 TB/s these layers stream at (profiles/r01/bench_b16_stage_table.txt).  Here t2 never leaves the registers and `out`
// is consumed by the next conv1 before it is stored: t1 + x in, out + t1n out = 820 MB.  No halo recomputation is needed -- the
// only 3x3 in the chain reads t1 from memory as before; the two 1x1 GEMMs that follow are per pixel.
//
// How three GEMMs chain through registers.  D[channel][pixel] = sum_k W[channel][k] . X[pixel][k] with the weights as the
// MFMA A operand (conv_igemm.hip): the accumulator of v_mfma_f32_16x16x32_f16 leaves, in lane (fi = lane & 15, fq = lane >> 4),
// rows 4 fq + r (r = 0..3) of column fi.  With the weight rows of a 64-channel block read from LDS in the order
// row(i, fi) = (fi >> 2) * 16 + i * 4 + (fi & 3), accumulator i of a lane holds channels 16 fq + 4 i + r: 16 CONSECUTIVE
// channels of one pixel per lane.  The B operand of the next GEMM wants, per lane, 8 consecutive k of its pixel:
//     k-step s (32 k), lane (fi, fq), element j  <->  logical k = 32 s + 8 fq + j.
// Converting accumulators 2s and 2s+1 of the lane to fp16 gives channels 16 fq + 8 s + j, j = 0..7 -- so if the NEXT layer's
// weight matrix has its K columns stored in the order  kappa = 32 s + 8 fq + j  ->  channel 16 fq + 8 s + j  (per group of 64
// channels; weights.py `_perm_k64`), the accumulators ARE its B operand: no LDS round trip, no shuffles.  The K order inside an
// MFMA changes the fp32 summation order only.
//
// Work decomposition: one workgroup = BM consecutive pixels (n, y, x order), 4 waves, each wave owns BM/4 pixels and ALL
// channels of every stage.  conv2 is the usual implicit GEMM (9 taps = 9 K steps of 64, activations and the tap's 64x64 weight
// block staged by LDS-DMA into 128-byte XOR-swizzled rows, double buffered).  conv3 + next conv1 run as 4 passes over output
// channel groups of 64: the pass's W3 rows (8 KB) and W1n columns (8 KB) are staged into the same two LDS buffers one pass
// ahead; per pixel block the wave computes out (bias, residual, ReLU, 32-byte stores), rounds it to fp16 exactly as the store
// does and feeds it to the t1n accumulators.  The block input (residual) of a pass is loaded one pass ahead into registers.
#include <type_traits>

#include "common.h"

namespace {

// 4 waves x NJ*16 pixels.  NJ = 2 (128-pixel tiles): THREE (by registers: two) workgroups per CU -- measured against NJ = 4
// (256-pixel tiles, 80 KB: only ONE workgroup fits a CU, two would need exactly the 160 KB the CU has): with a single resident
// workgroup nothing overlaps its waits and the kernel's costs simply add up (0.26 ms = 0.10 compute skeleton + 0.10 residual
// stream + 0.04 conv2 staging + 0.02 stores, ablation builds -DRS_BNECK_DIAG).
constexpr int NJ = 2, NT = 256, BM = 4 * NJ * 16, PXW = NJ * 16;

// CB = bottleneck width / 64: 1 = res2 (64 -> 256), 2 = res3 (128 -> 512).
template <int CB>
struct Cfg {
  static constexpr int CBW = 64 * CB;              // bottleneck width (t1, t2, t1n channels)
  static constexpr int C4 = 256 * CB;              // block input / output channels
  static constexpr int MIB = 4 * CB;               // 16-row blocks of a CBW-row weight matrix (conv2, next conv1)
  static constexpr int NPASS = 4 * CB;             // conv3 output channel groups of 64
  static constexpr int KS2 = 9 * CB;               // conv2 K steps of 64: 64-channel slice outer, taps inner
  static constexpr int RING = CB == 1 ? 3 : 2;     // LDS stage buffers of the conv2 loop
  static constexpr int STAGE = (BM + CBW) * 128;   // activations [BM][128 B] + weights [CBW][128 B]; a pass's slices fit the same space
  static constexpr int W1_OFF = CB * 8192;         // pass slice layout: W3 (CB sub-tiles of [64][128 B]), W1n [CBW][128 B], Wsc [64][128 B]
  static constexpr int SC_OFF = 2 * CB * 8192;
  static constexpr int NBIAS = 2 * CBW + C4;       // b2 [CBW], b3 [C4], b1n [CBW] as fp32
  static constexpr int BIAS_OFF = RING * STAGE;
  static constexpr int LDS_BYTES = RING * STAGE + NBIAS * 4;   // CB 1: 73.5 KB, CB 2: 67 KB -- two workgroups per CU
};

__device__ __forceinline__ void glds16(const half_t* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// A 16-byte global load the COMPILER does not track (hipcc waits vmcnt(0) at the first use of any ordinary load while LDS-DMA
// pieces are in flight, cdna_hip_programming.md "Projection GEMM" item 4b): the caller owns the wait -- a counted s_waitcnt that
// names the destination registers as operands, so no use can move above it.
__device__ __forceinline__ void gload16_untracked(half8& dst, const half_t* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}

__device__ __forceinline__ half8 pack8(const f32x4& a, const f32x4& b) {
  half8 h;
  h[0] = (half_t)a[0]; h[1] = (half_t)a[1]; h[2] = (half_t)a[2]; h[3] = (half_t)a[3];
  h[4] = (half_t)b[0]; h[5] = (half_t)b[1]; h[6] = (half_t)b[2]; h[7] = (half_t)b[3];
  return h;
}
__device__ __forceinline__ float clamp_h(float f) { return f > 65504.f ? 65504.f : f; }     // after ReLU: only the upper bound matters

// DBG (diagnostic builds through RS_BNECK_DBG, results wrong by construction): 1 = no residual loads, 2 = no `out` stores,
// 4 = no conv2 activation staging
// SC: the block's shortcut is a 1x1 projection of a 64-channel input x0 at the same resolution (res2.0: the stem output) instead
// of the identity: out = relu(W3 . t2 + Wsc . x0 + b), i.e. two more K steps per pass whose B operand (x0 of the lane's pixels)
// is loaded once per tile straight into registers in the MFMA's own k order (Wsc keeps its natural column order).
template <int CB, bool NEXT, int DBG = 0, bool SC = false>
__global__ __launch_bounds__(NT, 2) void bneck_tail_kernel(const BneckParams p) {
  using G = Cfg<CB>;
  constexpr int CBW = G::CBW, C4 = G::C4, MIB = G::MIB, NPASS = G::NPASS, KS2 = G::KS2, RING = G::RING, STAGE = G::STAGE;
  static_assert(!SC || CB == 1, "the projection-shortcut form exists for the 64-wide stage only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = p.M;
  const int ntiles = (M + BM - 1) / BM;
  const int q = blockIdx.x;
  if (q >= ntiles) return;
  int m0;
  {
    const int qn = ntiles >> 3, r = ntiles & 7, x = q & 7;      // XCD-aware order (conv_igemm.hip): neighbouring tiles share an L2
    m0 = ((x < r ? x * (qn + 1) : r * (qn + 1) + (x - r) * qn) + (q >> 3)) * BM;
  }
  const int lrow = lane >> 3, lchk = lane & 7;
  // ---- conv2 staging pointers: BM/32 passes of 32 activation rows, CBW/32 passes of weight rows (this wave: rows ps*32 + wave*8 + lrow)
  const half_t* aptr[BM / 32];
#pragma unroll
  for (int ps = 0; ps < BM / 32; ++ps) {
    int m = m0 + ps * 32 + wave * 8 + lrow;
    if (m >= M) m = M - 1;
    const int x = m % p.W, t = m / p.W, y = t % p.H, n = t / p.H;
    aptr[ps] = p.t1 + ((long long)(n * p.Hp + y) * p.Wp + x) * CBW + (lchk ^ lrow) * 8;     // tap (0,0) of the zero-haloed map
  }
  const half_t* wptr[CBW / 32];
#pragma unroll
  for (int ps = 0; ps < CBW / 32; ++ps) {
    const int row = ps * 32 + wave * 8 + lrow;
    const int key = (row & 3) | (((row / (4 * MIB)) & 1) << 2);
    wptr[ps] = p.w2 + (long long)row * (9 * CBW) + (lchk ^ key) * 8;
  }
  auto stage_tap = [&](int buf, int step) {                     // K step = (64-channel slice, tap)
    char* abase = smem + buf * STAGE;
    const int slice = step / 9, tap = step - slice * 9;
    const int a_off = ((tap / 3) * p.Wp + (tap % 3)) * CBW + slice * 64;
    if (!(DBG & 4)) {
#pragma unroll
      for (int ps = 0; ps < BM / 32; ++ps) glds16(aptr[ps] + a_off, abase + (ps * 32 + wave * 8) * 128);
    }
#pragma unroll
    for (int ps = 0; ps < CBW / 32; ++ps) glds16(wptr[ps] + tap * CBW + slice * 64, abase + BM * 128 + (ps * 32 + wave * 8) * 128);
  };
  // pass slices of conv3 / next conv1: W3p rows 64 pass .. +63 as CB sub-tiles of 64 K columns, W1p columns 64 pass .. +63 of its
  // CBW rows, (SC) Wsc rows 64 pass .. +63
  auto stage_pass = [&](int buf, int pass) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = ps * 32 + wave * 8 + lrow;
      const int key = (row & 3) | (((row >> 4) & 1) << 2);
#pragma unroll
      for (int c = 0; c < CB; ++c)
        glds16(p.w3p + (long long)(pass * 64 + row) * CBW + c * 64 + (lchk ^ key) * 8, base + c * 8192 + (ps * 32 + wave * 8) * 128);
      if (SC) glds16(p.wsc + (long long)(pass * 64 + row) * 64 + (lchk ^ key) * 8, base + G::SC_OFF + (ps * 32 + wave * 8) * 128);
    }
    if (NEXT) {
#pragma unroll
      for (int ps = 0; ps < CBW / 32; ++ps) {
        const int row = ps * 32 + wave * 8 + lrow;
        const int key = (row & 3) | (((row / (4 * MIB)) & 1) << 2);
        glds16(p.w1p + (long long)row * C4 + pass * 64 + (lchk ^ key) * 8, base + G::W1_OFF + (ps * 32 + wave * 8) * 128);
      }
    }
  };

  const int fi = lane & 15, fq = lane >> 4, fkey = lane & 7;
  int w_off[4], wB_off[MIB], x_off[NJ];          // rows of a 64-row slice / of a CBW-row matrix this lane reads as MFMA A rows
#pragma unroll
  for (int i = 0; i < 4; ++i) w_off[i] = ((fi >> 2) * 16 + i * 4 + (fi & 3)) * 128;
#pragma unroll
  for (int i = 0; i < MIB; ++i) wB_off[i] = ((fi >> 2) * 4 * MIB + i * 4 + (fi & 3)) * 128;
#pragma unroll
  for (int j = 0; j < NJ; ++j) x_off[j] = (wave * PXW + j * 16 + fi) * 128;
  const int c_off[2] = {(fq ^ fkey) * 16, ((4 + fq) ^ fkey) * 16};

  // ---- pixel geometry of this lane's 4 pixels (pixel block j, column fi)
  long long opix[NJ];
  bool valid[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int m = m0 + wave * PXW + j * 16 + fi;
    valid[j] = m < M;
    const int mm = valid[j] ? m : M - 1;
    const int x = mm % p.W, t = mm / p.W, y = t % p.H, n = t / p.H;
    opix[j] = (long long)(n * p.Hp + y + 1) * p.Wp + x + 1;
  }
  // Residual (block input) of one pass: 32 bytes per pixel and lane, loaded ONE PASS AHEAD into registers -- issued right after
  // the wait that opens pass q, consumed in pass q+1 behind that pass's wait, so the HBM latency of the 328 MB residual stream
  // hides behind a pass of MFMAs instead of being paid 16 times per tile.
  half8 rb[2][NJ][2];
  half8 xq[2][NJ];              // SC: x0 of this lane's pixels as the B operand of two K steps (channels 32 s + 8 fq .. +7)
  auto load_res = [&](half8 (&dst)[NJ][2], int pass) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const half_t* xp = p.x + opix[j] * C4 + pass * 64 + fq * 16;
      if ((DBG & 1) || SC) { dst[j][0] = half8{}; dst[j][1] = half8{}; continue; }
      gload16_untracked(dst[j][0], xp);
      gload16_untracked(dst[j][1], xp + 8);
    }
  };

  // Biases go through LDS: an ordinary global load anywhere between two prefetch points would make the compiler wait vmcnt(0) at
  // its first use and so drain the LDS-DMA pieces and residual loads that are supposed to stay in flight across a pass.
  float* bias_s = (float*)(smem + G::BIAS_OFF);
  for (int i = tid; i < G::NBIAS; i += NT) bias_s[i] = i < CBW ? p.b2[i] : (i < CBW + C4 ? p.b3[i - CBW] : (NEXT ? p.b1[i - CBW - C4] : 0.f));

  // ================================================================ conv2: KS2 K steps
  f32x4 acc[MIB][NJ];
#pragma unroll
  for (int i = 0; i < MIB; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // CB 1, three-buffer ring: step t+2 is issued while step t is computed, so a piece has TWO K steps plus the other workgroup's
  // share of the CU to arrive; the wait that opens step t leaves exactly the pieces of step t+1 (BM/32 activation + CBW/32 weight
  // pieces per wave) in flight.  CB 2 (32 KB stages): two buffers, one step ahead.  The pass slices reuse buffers 0 and 1.
  constexpr int PIECES = BM / 32 + CBW / 32;
  __syncthreads();                                  // biases are in LDS; nothing else is in flight
  stage_tap(0, 0);
  if (RING == 3) stage_tap(1, 1);
#pragma unroll
  for (int t = 0; t < KS2; ++t) {
    if (RING == 3 && t + 1 < KS2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // raw barrier: __syncthreads() would drain the piece in flight (its fence waits vmcnt(0))
    if (t + RING - 1 < KS2) stage_tap((t + RING - 1) % RING, t + RING - 1);
    else if (t == KS2 - 1) {                        // the buffer(s) of the earlier steps are free: first conv3 / conv1n slices + residuals
      stage_pass(0, 0);
      load_res(rb[0], 0);
      if (SC) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const half_t* xp = p.x0 + opix[j] * 64 + fq * 8;
          xq[0][j] = *(const half8*)xp;
          xq[1][j] = *(const half8*)(xp + 32);
        }
      }
    }
    const char* sb = smem + (t % RING) * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8 wf[MIB], xf[NJ];
#pragma unroll
      for (int i = 0; i < MIB; ++i) wf[i] = *(const half8*)(sb + BM * 128 + wB_off[i] + c_off[kk]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) xf[j] = *(const half8*)(sb + x_off[j] + c_off[kk]);
#pragma unroll
      for (int i = 0; i < MIB; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }
  // ---- t2 = relu(acc + b2), rounded to fp16 exactly as a store would: the B operand of conv3 (2 CB K steps of 32); the lane holds
  // channels 16 CB fq + 4 i + r, K step s takes accumulators 2s and 2s+1 = channels 16 CB fq + 8 s + j
  half8 tf[2 * CB][NJ];
  {
    f32x4 b2v[MIB];
#pragma unroll
    for (int i = 0; i < MIB; ++i) b2v[i] = *(const f32x4*)(bias_s + fq * 4 * MIB + i * 4);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int i = 0; i < MIB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r] + b2v[i][r]; acc[i][j][r] = clamp_h(v > 0.f ? v : 0.f); }
#pragma unroll
      for (int s2 = 0; s2 < 2 * CB; ++s2) tf[s2][j] = pack8(acc[2 * s2][j], acc[2 * s2 + 1][j]);
    }
  }
  const bool full_tile = m0 + BM <= M;
  f32x4 acc3[MIB][NJ];         // t1n accumulators: channels 16 CB fq + 4 i + r of pixel (j, fi)
#pragma unroll
  for (int i = 0; i < MIB; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ================================================================ conv3 (+ residual + ReLU) and the next conv1, 4 passes of 64 output channels
  // the pass loop exists twice: full tiles leave the previous pass's stores in flight, the partial last tile (some waves skip
  // stores, so their count is unknown) drains everything
  auto passes = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    // this pass's slices and residuals were issued BEFORE the previous pass's 2*NJ `out` stores: leave exactly those in flight
    // (a partial last tile skips stores, so it drains everything)
    static_assert(NJ == 2, "the wait below names the 2*NJ residual registers of the pass");
    // ONE wait statement per pass and code path: with two alternatives behind a run-time branch the compiler merges the "+v"
    // operands through copies placed BEFORE the wait -- copies of registers whose loads are still in flight (seen in the ISA).
    // (pass 0: its slices and residuals were waited for before the two code paths split, below)
    if constexpr (FULL) {
      if (pass > 0) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(rb[pass & 1][0][0]), "+v"(rb[pass & 1][0][1]), "+v"(rb[pass & 1][1][0]), "+v"(rb[pass & 1][1][1]) : "n"(2 * NJ) : "memory");
    } else {
      if (pass > 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb[pass & 1][0][0]), "+v"(rb[pass & 1][0][1]), "+v"(rb[pass & 1][1][0]), "+v"(rb[pass & 1][1][1]) :: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // pass slices + this pass's residuals landed; everyone is done with the other buffer
    if (pass + 1 < NPASS) { stage_pass((pass + 1) & 1, pass + 1); load_res(rb[(pass + 1) & 1], pass + 1); }
    const char* sb = smem + (pass & 1) * STAGE;
    const int cb = pass * 64 + fq * 16;              // this lane's 16 output channels of the pass
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f32x4 a2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CB; ++c) {               // K sub-tile c of the W3 slice = K steps 2c, 2c+1
          const half8 wa = *(const half8*)(sb + c * 8192 + w_off[i] + c_off[0]), wb = *(const half8*)(sb + c * 8192 + w_off[i] + c_off[1]);
          a2[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, tf[2 * c][j], a2[i], 0, 0, 0);
          a2[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, tf[2 * c + 1][j], a2[i], 0, 0, 0);
        }
        if (SC) {
          const half8 sa = *(const half8*)(sb + G::SC_OFF + w_off[i] + c_off[0]), sc = *(const half8*)(sb + G::SC_OFF + w_off[i] + c_off[1]);
          a2[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sa, xq[0][j], a2[i], 0, 0, 0);
          a2[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sc, xq[1][j], a2[i], 0, 0, 0);
        }
      }
      f32x4 b3q[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) b3q[i] = *(const f32x4*)(bias_s + CBW + cb + i * 4);      // from LDS, per pixel block: no registers held across the pass
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float res = SC ? 0.f : (float)rb[pass & 1][j][i >> 1][(i & 1) * 4 + r];     // loaded one pass ago (gload16_untracked), covered by the wait above
          const float v = a2[i][r] + b3q[i][r] + res;
          a2[i][r] = clamp_h(v > 0.f ? v : 0.f);
        }
      const half8 o0 = pack8(a2[0], a2[1]), o1 = pack8(a2[2], a2[3]);
      if (valid[j] && !(DBG & 2)) {
        half_t* op = p.out + opix[j] * C4 + cb;
        *(half8*)op = o0;
        *(half8*)(op + 8) = o1;
      }
      if (NEXT) {
#pragma unroll
        for (int i = 0; i < MIB; ++i) {
          const half8 wa = *(const half8*)(sb + G::W1_OFF + wB_off[i] + c_off[0]), wb = *(const half8*)(sb + G::W1_OFF + wB_off[i] + c_off[1]);
          acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, o0, acc3[i][j], 0, 0, 0);
          acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, o1, acc3[i][j], 0, 0, 0);
        }
      }
    }
  }
  };
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb[0][0][0]), "+v"(rb[0][0][1]), "+v"(rb[0][1][0]), "+v"(rb[0][1][1]) :: "memory");   // pass 0's pieces
  if (full_tile) passes(std::true_type{});
  else passes(std::false_type{});
  if (NEXT) {
    f32x4 b1v[MIB];
#pragma unroll
    for (int i = 0; i < MIB; ++i) b1v[i] = *(const f32x4*)(bias_s + CBW + C4 + fq * 4 * MIB + i * 4);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (!valid[j]) continue;
#pragma unroll
      for (int i = 0; i < MIB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float v = acc3[i][j][r] + b1v[i][r]; acc3[i][j][r] = clamp_h(v > 0.f ? v : 0.f); }
      half_t* op = p.t1n + opix[j] * CBW + fq * 4 * MIB;
#pragma unroll
      for (int k = 0; k < MIB / 2; ++k) *(half8*)(op + 8 * k) = pack8(acc3[2 * k][j], acc3[2 * k + 1][j]);
    }
  }
}

template <int CB>
int launch_cb(const BneckParams& p, hipStream_t stream) {
  using G = Cfg<CB>;
  const long long nblk = cdiv(p.M, BM);
  RS_CHECK(nblk < (1ll << 31), RS_ERR_ARG, "bneck_tail: grid too large");
  const void* k;
  if constexpr (CB == 1)
    k = p.x0 ? (p.w1p ? (const void*)bneck_tail_kernel<1, true, 0, true> : (const void*)bneck_tail_kernel<1, false, 0, true>)
             : (p.w1p ? (const void*)bneck_tail_kernel<1, true> : (const void*)bneck_tail_kernel<1, false>);
  else
    k = p.w1p ? (const void*)bneck_tail_kernel<CB, true> : (const void*)bneck_tail_kernel<CB, false>;
#ifdef RS_BNECK_DIAG
  {
    static bool once = false;
    if (!once) {
      once = true;
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, NT, G::LDS_BYTES);
      fprintf(stderr, "[bneck diag] CB %d occupancy API: %d workgroups of %d threads / %d B LDS per CU\n", CB, nb, NT, G::LDS_BYTES);
    }
  }
  if (p.w1p && !p.x0) {                // diagnostic build only (make EXTRA=-DRS_BNECK_DIAG): RS_DEEP_DBG selects the ablation of the NEXT form
    const void* kd = nullptr;
    switch (rs_debug().deep_dbg) {
      case 1: kd = (const void*)bneck_tail_kernel<CB, true, 1>; break;
      case 2: kd = (const void*)bneck_tail_kernel<CB, true, 2>; break;
      case 4: kd = (const void*)bneck_tail_kernel<CB, true, 4>; break;
      case 7: kd = (const void*)bneck_tail_kernel<CB, true, 7>; break;
      default: break;
    }
    if (kd) {
      RS_HIP(hipFuncSetAttribute(kd, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
      BneckParams pd = p;
      void* argsd[] = {&pd};
      RS_HIP(hipLaunchKernel(kd, dim3((unsigned)nblk), dim3(NT), argsd, G::LDS_BYTES, stream));
      return RS_OK;
    }
  }
#endif
  static bool attr[4] = {false, false, false, false};
  const int ai = (p.w1p ? 1 : 0) | (p.x0 ? 2 : 0);
  if (!attr[ai]) {
    RS_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr[ai] = true;
  }
  BneckParams pc = p;
  void* args[] = {&pc};
  RS_HIP(hipLaunchKernel(k, dim3((unsigned)nblk), dim3(NT), args, G::LDS_BYTES, stream));
  return RS_OK;
}

}  // namespace

int launch_bneck_tail(const BneckParams& p, hipStream_t stream) {
  RS_CHECK(p.M > 0 && p.t1 && p.w2 && p.b2 && p.w3p && p.b3 && p.out, RS_ERR_ARG, "bneck_tail: null argument");
  RS_CHECK((p.x != nullptr) != (p.x0 != nullptr && p.wsc != nullptr), RS_ERR_ARG, "bneck_tail: give either the identity residual x or the projection shortcut x0 + wsc");
  RS_CHECK(p.Hp == p.H + 2 && p.Wp == p.W + 2, RS_ERR_ARG, "bneck_tail: maps must carry a halo of 1");
  RS_CHECK(!p.w1p || (p.b1 && p.t1n), RS_ERR_ARG, "bneck_tail: next conv1 needs weights, bias and output");
  RS_CHECK(p.CB == 1 || (p.CB == 2 && !p.x0), RS_ERR_UNSUPPORTED, "bneck_tail: bottleneck width %d (64 or 128; projection shortcut: 64 only)", 64 * p.CB);
  return p.CB == 1 ? launch_cb<1>(p, stream) : launch_cb<2>(p, stream);
}
