// Fused tail of an identity-shortcut bottleneck block in the SPLIT-OPERAND precision mode (rs_spec.precision == 2; common.h ConvParams::split):
//
//     t2  = relu(conv3x3(t1, W2) + b2)             CBW -> CBW     [EXT d2: modeling/backbone/resnet.py BottleneckBlock.conv2]
//     out = relu(W3 . t2 + b3 + x)                 CBW -> 4 CBW   conv3 + identity shortcut + ReLU
//     t1n = relu(W1n . out + b1n)                  4 CBW -> CBW   the NEXT block's conv1 (optional)
//
// in ONE launch, every tensor as hi + lo fp16 planes (value = hi + lo, 22 significand bits) and every product as W_hi.X_lo + W_hi.X_hi + W_lo.X_hi on
// v_mfma_f32_16x16x32_f16.  Layer by layer the split engine moves, per block of res2 at batch 16 (640 000 pixels, 4 bytes per element): t1 164 MB in, t2 164
// out + 164 in, x 655 in, out 655 out, and for the next conv1 out 655 in + t1n 164 out = 2.6 GB at 3.4-4.4 TB/s (profiles/r04/split_stage_table.txt: 0.70 ms
// for the three launches).  Here t2 never leaves the registers and `out` is consumed by the next conv1 before it is stored: 1.64 GB.
//
// The chain through registers is bneck_fused.hip's (the fp16 form of this kernel), with two fragments per operand instead of one: with the weight rows of
// a GEMM read from LDS in the order row(i, fi) = (fi >> 2) * 4 MIB + i * 4 + (fi & 3), accumulator i of lane (fi, fq) holds channels 16 CB fq + 4 i + r of
// pixel fi; relu(acc * scale + bias), split into hi = fp16(v), lo = fp16(v - hi) exactly as a store would split it, gives per lane 8 consecutive channels per
// pair of accumulators -- the B operand (hi fragment, lo fragment) of one 32-deep K step of the next GEMM when that layer's weight has its K columns in the
// order of weights.py `_perm_k64` (group = CBW for conv3, 64 for the next conv1).  The K order inside an MFMA changes the fp32 summation order only.
//
// Unlike bneck_fused.hip this file takes no liberties with the compiler: operands are staged by LDS-DMA behind plain waits and barriers, the residual is a
// plain global load.  Two workgroups per CU (64 KB of LDS each) overlap one's waits with the other's matrix work.
//
// Projection form (PROJ; the first block of res2, whose 1x1 shortcut reads the 64-channel stem output at the block's own resolution): no residual; the
// shortcut's GEMM is two more K steps of conv3 -- w3p = [conv3 in the chained order | shortcut in natural order] under one scale per row -- whose B
// fragments are read once per tile from a copy of the x0 tile staged over the retired conv2 stages.  At batch 16: conv2 0.200 + conv3 (dual source)
// 0.268 + the next conv1 0.195 ms -> 0.455 ms.
//
// What bounds it (profiles/r04/tail_split_ablation.txt): every workgroup alternates a matrix-bound conv2 phase with an HBM-bound conv3 phase and the two
// overlap badly across workgroups; a deeper-pipelined variant (three-stage ring, double-buffered pass slices, residual one pass ahead) measured the same.
#include "common.h"

namespace {

constexpr int NJ = 2, NT = 256, BM = 4 * NJ * 16, PXW = NJ * 16;     // 4 waves x 32 pixels

template <int CB, bool PROJ = false>
struct SCfg {
  static constexpr int CBW = 64 * CB;              // bottleneck width
  static constexpr int C4 = 256 * CB;              // block input / output channels
  static constexpr int MIB = 4 * CB;               // 16-row blocks of a CBW-row weight matrix (conv2, next conv1)
  static constexpr int NPASS = 4 * CB;             // conv3 output channel groups of 64
  static constexpr int KS2 = 9 * 2 * CB;           // conv2 K steps of 32 channels x (hi, lo): 32-channel slice outer, taps inner
  static constexpr int STAGE = (BM + CBW) * 128;   // conv2 stage: activation rows + weight rows, 128 B = [32 hi | 32 lo] each
  static constexpr int S3 = 2 * CB + (PROJ ? 2 : 0);   // K steps of conv3 (+ the 64-channel projection shortcut)
  static constexpr int W3K = CBW + (PROJ ? 64 : 0);    // K columns of a row of w3p
  static constexpr int W3_BYTES = S3 * 64 * 128;       // pass slice of W3p: S3 K steps x 64 rows x 128 B
  static constexpr int W1_BYTES = 2 * CBW * 128;       // pass slice of W1p: 2 K steps x CBW rows x 128 B
  static constexpr int PASS_BYTES = W3_BYTES + W1_BYTES;
  static constexpr int LDS_BYTES = 2 * STAGE > PASS_BYTES ? 2 * STAGE : PASS_BYTES;     // CB 1: 48 KB, CB 2: 64 KB
};

__device__ __forceinline__ void glds16s(const half_t* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// fp32 values of two accumulators (8 consecutive channels) -> hi / lo fragments
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, half8& h, half8& l) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    h[r] = (half_t)a[r]; l[r] = (half_t)(a[r] - (float)h[r]);
    h[4 + r] = (half_t)b[r]; l[4 + r] = (half_t)(b[r] - (float)h[4 + r]);
  }
}
__device__ __forceinline__ float relu_clamp(float v) { v = v > 0.f ? v : 0.f; return v > 65504.f ? 65504.f : v; }

#define RS_MFMA3(acc, wh, wl, xh, xl)                                      \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc, 0, 0, 0);      \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc, 0, 0, 0);      \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc, 0, 0, 0);

template <int CB, bool NEXT, bool PROJ>
__global__ __launch_bounds__(NT, 2) void bneck_tail_split_kernel(const BneckSplitParams p) {
  using G = SCfg<CB, PROJ>;
  static_assert(!PROJ || CB == 1, "projection form: 64-wide stage only");
  constexpr int CBW = G::CBW, C4 = G::C4, MIB = G::MIB, NPASS = G::NPASS, KS2 = G::KS2, STAGE = G::STAGE;
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
  // ---- conv2 staging: data chunk d of a 128-byte row = hi halfs 8 d .. 8 d + 7 of the step's 32 channels (d < 4) or their lo halfs (d >= 4)
  const half_t* aptr[BM / 32];
#pragma unroll
  for (int ps = 0; ps < BM / 32; ++ps) {
    int m = m0 + ps * 32 + wave * 8 + lrow;
    if (m >= M) m = M - 1;
    const int x = m % p.W, t = m / p.W, y = t % p.H, n = t / p.H;
    const int d = lchk ^ lrow;
    aptr[ps] = p.t1 + ((long long)(n * p.Hp + y) * p.Wp + x) * CBW + (d & 3) * 8 + (d >> 2) * p.t1_lo;     // tap (0,0) of the zero-haloed map
  }
  const half_t* wptr[CBW / 32];
#pragma unroll
  for (int ps = 0; ps < CBW / 32; ++ps) {
    const int row = ps * 32 + wave * 8 + lrow;
    const int key = (row & 3) | (((row / (4 * MIB)) & 1) << 2);
    const int d = lchk ^ key;
    wptr[ps] = p.w2 + (long long)row * (9 * CBW) + (d & 3) * 8 + (d >> 2) * p.w2_lo;
  }
  auto stage_tap = [&](int buf, int step) {                     // K step = (32-channel slice, tap)
    char* abase = smem + buf * STAGE;
    const int slice = step / 9, tap = step - slice * 9;
    const int a_off = ((tap / 3) * p.Wp + (tap % 3)) * CBW + slice * 32;
#pragma unroll
    for (int ps = 0; ps < BM / 32; ++ps) glds16s(aptr[ps] + a_off, abase + (ps * 32 + wave * 8) * 128);
#pragma unroll
    for (int ps = 0; ps < CBW / 32; ++ps) glds16s(wptr[ps] + tap * CBW + slice * 32, abase + BM * 128 + (ps * 32 + wave * 8) * 128);
  };
  // pass slices: W3p rows 64 pass .. +63 as 2 CB K steps of 32 (its K = CBW columns), W1p columns 64 pass .. +63 (2 K steps) of its CBW rows
  auto stage_pass = [&](int pass) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = ps * 32 + wave * 8 + lrow;
      const int key = (row & 3) | (((row >> 4) & 1) << 2);
      const int d = lchk ^ key;
#pragma unroll
      for (int s = 0; s < G::S3; ++s)
        glds16s(p.w3p + (long long)(pass * 64 + row) * G::W3K + s * 32 + (d & 3) * 8 + (d >> 2) * p.w3_lo, smem + s * 8192 + (ps * 32 + wave * 8) * 128);
    }
    if (NEXT) {
#pragma unroll
      for (int ps = 0; ps < CBW / 32; ++ps) {
        const int row = ps * 32 + wave * 8 + lrow;
        const int key = (row & 3) | (((row / (4 * MIB)) & 1) << 2);
        const int d = lchk ^ key;
#pragma unroll
        for (int s = 0; s < 2; ++s)
          glds16s(p.w1p + (long long)row * C4 + pass * 64 + s * 32 + (d & 3) * 8 + (d >> 2) * p.w1_lo, smem + G::W3_BYTES + s * CBW * 128 + (ps * 32 + wave * 8) * 128);
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
  const int ch_off = (fq ^ fkey) * 16, cl_off = ((4 + fq) ^ fkey) * 16;        // this lane's hi / lo fragment inside a row

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

  // ================================================================ conv2
  f32x4 acc[MIB][NJ];
#pragma unroll
  for (int i = 0; i < MIB; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  stage_tap(0, 0);
  auto stage_x0 = [&](char* base) {      // the shortcut's input at the tile's own pixels (tap (1,1) of its haloed map), both 32-channel steps, rows swizzled like conv2's
#pragma unroll
    for (int ps = 0; ps < BM / 32; ++ps) {
      int m = m0 + ps * 32 + wave * 8 + lrow;
      if (m >= M) m = M - 1;
      const int x = m % p.W, t = m / p.W, y = t % p.H, n = t / p.H;
      const int d = lchk ^ lrow;
      const half_t* xp = p.x0 + ((long long)(n * p.Hp + y + 1) * p.Wp + x + 1) * 64 + (d & 3) * 8 + (d >> 2) * p.x0_lo;
#pragma unroll
      for (int s = 0; s < 2; ++s) glds16s(xp + s * 32, base + s * BM * 128 + (ps * 32 + wave * 8) * 128);
    }
  };
  for (int t = 0; t < KS2; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                // step t has landed; everyone is done reading the other buffer
    if (t + 1 < KS2) stage_tap((t + 1) & 1, t + 1);
    const char* sb = smem + (t & 1) * STAGE;
    half8 xh[NJ], xl[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { xh[j] = *(const half8*)(sb + x_off[j] + ch_off); xl[j] = *(const half8*)(sb + x_off[j] + cl_off); }
#pragma unroll
    for (int i = 0; i < MIB; ++i) {       // weight fragments of one 16-row block at a time: 2 CB x 16 registers less than all of them up front
      const half8 wh = *(const half8*)(sb + BM * 128 + wB_off[i] + ch_off), wl = *(const half8*)(sb + BM * 128 + wB_off[i] + cl_off);
#pragma unroll
      for (int j = 0; j < NJ; ++j) { RS_MFMA3(acc[i][j], wh, wl, xh[j], xl[j]) }
    }
  }
  // ---- t2 = relu(acc * s2 + b2) as hi / lo fragments: K step s of conv3 takes accumulators 2s, 2s+1 = channels 16 CB fq + 8 s + j
  half8 th[2 * CB][NJ], tl[2 * CB][NJ];
  {
#pragma unroll
    for (int i = 0; i < MIB; ++i) {
      const f32x4 sc = *(const f32x4*)(p.s2 + fq * 4 * MIB + i * 4), bv = *(const f32x4*)(p.b2 + fq * 4 * MIB + i * 4);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = relu_clamp(acc[i][j][r] * sc[r] + bv[r]);
    }
#pragma unroll
    for (int s = 0; s < 2 * CB; ++s)
#pragma unroll
      for (int j = 0; j < NJ; ++j) split8(acc[2 * s][j], acc[2 * s + 1][j], th[s][j], tl[s][j]);
  }
  half8 x0h[PROJ ? 2 : 1][NJ], x0l[PROJ ? 2 : 1][NJ];      // projection form: B fragments of the shortcut's two K steps
  if (PROJ) {
    // staged over the conv2 stages once they are done with (a resident copy from the prologue on -- 80 KB of LDS -- measured the same: 0.455 / 0.453 ms)
    const char* xb = smem;
    __syncthreads();
    stage_x0(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        x0h[s][j] = *(const half8*)(xb + s * BM * 128 + x_off[j] + ch_off);
        x0l[s][j] = *(const half8*)(xb + s * BM * 128 + x_off[j] + cl_off);
      }
  }
  f32x4 acc3[MIB][NJ];         // t1n accumulators: channels 16 CB fq + 4 i + r of pixel (j, fi)
#pragma unroll
  for (int i = 0; i < MIB; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ================================================================ conv3 (+ residual + ReLU) and the next conv1: NPASS passes of 64 output channels
  for (int pass = 0; pass < NPASS; ++pass) {
    __syncthreads();                                // everyone is done with the stage buffers / the previous pass's slices
    stage_pass(pass);
    const int cb = pass * 64 + fq * 16;             // this lane's 16 output channels of the pass
    // residual of the pass (block input, both planes): plain loads, in flight next to the slices
    half8 rh[NJ][2], rl[NJ][2];
    if (!PROJ) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const half_t* xp = p.x + opix[j] * C4 + cb;
        rh[j][0] = *(const half8*)xp; rh[j][1] = *(const half8*)(xp + 8);
        rl[j][0] = *(const half8*)(xp + p.x_lo); rl[j][1] = *(const half8*)(xp + p.x_lo + 8);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                // the pass slices have landed
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f32x4 a2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2 * CB; ++s) {
          const half8 wh = *(const half8*)(smem + s * 8192 + w_off[i] + ch_off), wl = *(const half8*)(smem + s * 8192 + w_off[i] + cl_off);
          RS_MFMA3(a2[i], wh, wl, th[s][j], tl[s][j])
        }
        if (PROJ) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const half8 wh = *(const half8*)(smem + (2 * CB + s) * 8192 + w_off[i] + ch_off), wl = *(const half8*)(smem + (2 * CB + s) * 8192 + w_off[i] + cl_off);
            RS_MFMA3(a2[i], wh, wl, x0h[s][j], x0l[s][j])
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 s3v = *(const f32x4*)(p.s3 + cb + i * 4), b3v = *(const f32x4*)(p.b3 + cb + i * 4);      // L1-resident: 2 x 256 CB floats per block
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float res = PROJ ? 0.f : (float)rh[j][i >> 1][(i & 1) * 4 + r] + (float)rl[j][i >> 1][(i & 1) * 4 + r];
          a2[i][r] = relu_clamp(a2[i][r] * s3v[r] + b3v[r] + res);
        }
      }
      half8 o0h, o0l, o1h, o1l;
      split8(a2[0], a2[1], o0h, o0l);
      split8(a2[2], a2[3], o1h, o1l);
      if (valid[j]) {
        half_t* op = p.out + opix[j] * C4 + cb;
        *(half8*)op = o0h; *(half8*)(op + 8) = o1h;
        *(half8*)(op + p.out_lo) = o0l; *(half8*)(op + p.out_lo + 8) = o1l;
      }
      if (NEXT) {
#pragma unroll
        for (int i = 0; i < MIB; ++i) {
          const char* wb = smem + G::W3_BYTES + wB_off[i];
          const half8 w0h = *(const half8*)(wb + ch_off), w0l = *(const half8*)(wb + cl_off);
          const half8 w1h = *(const half8*)(wb + CBW * 128 + ch_off), w1l = *(const half8*)(wb + CBW * 128 + cl_off);
          RS_MFMA3(acc3[i][j], w0h, w0l, o0h, o0l)
          RS_MFMA3(acc3[i][j], w1h, w1l, o1h, o1l)
        }
      }
    }
  }
  if (NEXT) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (!valid[j]) continue;
      half_t* op = p.t1n + opix[j] * CBW + fq * 4 * MIB;
#pragma unroll
      for (int k = 0; k < MIB / 2; ++k) {
        const f32x4 sa = *(const f32x4*)(p.s1 + fq * 4 * MIB + 8 * k), sb2 = *(const f32x4*)(p.s1 + fq * 4 * MIB + 8 * k + 4);
        const f32x4 ba = *(const f32x4*)(p.b1 + fq * 4 * MIB + 8 * k), bb = *(const f32x4*)(p.b1 + fq * 4 * MIB + 8 * k + 4);
        f32x4 va, vb;
#pragma unroll
        for (int r = 0; r < 4; ++r) { va[r] = relu_clamp(acc3[2 * k][j][r] * sa[r] + ba[r]); vb[r] = relu_clamp(acc3[2 * k + 1][j][r] * sb2[r] + bb[r]); }
        half8 h, l;
        split8(va, vb, h, l);
        *(half8*)(op + 8 * k) = h;
        *(half8*)(op + p.t1n_lo + 8 * k) = l;
      }
    }
  }
}

template <int CB, bool PROJ>
int launch_split_cb(const BneckSplitParams& p, hipStream_t stream) {
  using G = SCfg<CB, PROJ>;
  const long long nblk = cdiv(p.M, BM);
  RS_CHECK(nblk < (1ll << 31), RS_ERR_ARG, "bneck_tail_split: grid too large");
  const void* k = p.w1p ? (const void*)bneck_tail_split_kernel<CB, true, PROJ> : (const void*)bneck_tail_split_kernel<CB, false, PROJ>;
  static bool attr[2] = {false, false};
  const int ai = p.w1p ? 1 : 0;
  if (!attr[ai]) {
    RS_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr[ai] = true;
  }
  BneckSplitParams pc = p;
  void* args[] = {&pc};
  RS_HIP(hipLaunchKernel(k, dim3((unsigned)nblk), dim3(NT), args, G::LDS_BYTES, stream));
  return RS_OK;
}

}  // namespace

int launch_bneck_tail_split(const BneckSplitParams& p, hipStream_t stream) {
  RS_CHECK(p.M > 0 && p.t1 && p.w2 && p.b2 && p.s2 && p.w3p && p.b3 && p.s3 && (p.x || p.x0) && !(p.x && p.x0) && p.out, RS_ERR_ARG, "bneck_tail_split: null argument");
  RS_CHECK(p.Hp == p.H + 2 && p.Wp == p.W + 2, RS_ERR_ARG, "bneck_tail_split: maps must carry a halo of 1");
  RS_CHECK(!p.w1p || (p.b1 && p.s1 && p.t1n), RS_ERR_ARG, "bneck_tail_split: next conv1 needs weights, scales, bias and output");
  RS_CHECK(p.CB == 1 || p.CB == 2, RS_ERR_UNSUPPORTED, "bneck_tail_split: bottleneck width %d (64 or 128)", 64 * p.CB);
  if (p.x0) {
    RS_CHECK(p.CB == 1, RS_ERR_UNSUPPORTED, "bneck_tail_split: the projection-shortcut form exists for the 64-wide stage only");
    return launch_split_cb<1, true>(p, stream);
  }
  return p.CB == 1 ? launch_split_cb<1, false>(p, stream) : launch_split_cb<2, false>(p, stream);
}
