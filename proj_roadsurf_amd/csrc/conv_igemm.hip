// Implicit-GEMM convolution / GEMM for gfx950 (MI355X), fp16 in, fp32 accumulate on MFMA.
//
// Replaces, on the hot path: cuDNN conv fwd, cuBLAS sgemm and the
// cuDNN 2x2 transposed conv that detectron2 0.6 reaches through ATen
// ([EXT d2: layers/wrappers.py Conv2d, modeling/backbone/{resnet,fpn}.py,
//   modeling/roi_heads/{box_head,mask_head}.py]; layer list fixed by
//   R:config/detectron2_config_3bands.yaml:57-112,159-221).
//
// Formulation: D[channel][pixel] = sum_k W[channel][k] * X[pixel][k], k = (kh, kw, cin).
//   * weights are the MFMA "A" operand, activations the "B" operand, so each lane ends up with
//     4*MI CONSECUTIVE output channels of one pixel -> the epilogue stores 16-byte NHWC pieces
//     straight from registers (bias + residual + FPN top-down add + ReLU fused, fp32 math).
//   * both operands are staged global -> LDS with global_load_lds_dwordx4 (no VGPR round trip),
//     128-byte LDS rows, XOR swizzle applied on the per-lane SOURCE address and on the ds_read
//     (LDS destination stays lane-linear), double buffered, one barrier per 64-deep K step.
//   * activations carry a zero halo (common.h), so the gather has no bounds checks.
// (A channel permutation that makes every 16-byte epilogue store part of a 64-byte run was tried and
//  measured 5-13 % SLOWER on the HBM-bound 1x1 layers than the present 32-bytes-per-lane layout.)
// Tile variants (pixels x channels per workgroup): 128x128 (4 waves 2x2), 256x64 (4 waves 4x1),
// 256x16 fp32-out (small heads).  Each wave owns NJ*16 pixels x MI*16 channels.
#include "common.h"

namespace {

template <int WPX, int WCH, int MI, int NJ>
struct Tile {
  static constexpr int NW = WPX * WCH;
  static constexpr int NT = NW * 64;
  static constexpr int BM = WPX * NJ * 16;   // pixels
  static constexpr int BN = WCH * MI * 16;   // channels
  static constexpr int STAGE = (BM + BN) * 128;
  static constexpr int LDS = 2 * STAGE + 1024;   // + K-chunk offset table (small-Cin path)
};

__device__ __forceinline__ void glds16(const half_t* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// TRAIN: instantiation with the backward-epilogue options (down / res32 / mask / out_stride).  They are compiled out of the
// inference instantiations: carrying them as run-time branches cost the small-tile, occupancy-sensitive layers 20-70 %
// (measured: fused deconv 0.33 -> 0.58 ms, fpn_lateral2 0.24 -> 0.35 ms, 1760 -> 1530 tiles/s end to end).
// SPLIT: the split-operand precision mode (ConvParams::split, common.h).  A K step covers 32 channels: the 128-byte LDS row of an operand is
// [32 hi halfs | 32 lo halfs] of the same channels (data chunks 0-3 from the hi plane, 4-7 from the lo plane -- a per-lane constant added to the
// LDS-DMA source address), so that both planes of both operands are staged ONCE and the three products W_hi.X_lo, W_hi.X_hi, W_lo.X_hi (in this
// order, the same in conv_deep.hip: every tile accumulates an output in the same order) run on fragments of one stage.  K order: 32-channel slice
// outer, taps inner.  The stem (SMALLC: per-chunk tap table) keeps three passes over its padded K instead (hi.hi, hi.lo, lo.hi).  Per-row weight
// descale in the epilogue, outputs written as hi / lo planes.  Compiled out of the fp16 instantiations.
template <int WPX, int WCH, int MI, int NJ, bool SMALLC, bool GLDS, bool PERSIST, bool TRAIN = false, bool SPLIT = false>
__global__ __launch_bounds__(WPX* WCH * 64) void conv_igemm_kernel(const ConvParams p) {
  using T = Tile<WPX, WCH, MI, NJ>;
  constexpr int NW = T::NW, BM = T::BM, BN = T::BN;
  constexpr int PA = BM / (NW * 8);               // activation staging passes per K step
  constexpr int PW = (BN + NW * 8 - 1) / (NW * 8);  // weight staging passes
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
  // Tiles are walked in a grid-stride loop: launched with one workgroup per tile it runs once; launched
  // "persistent" (fewer workgroups than tiles) a workgroup streams through tiles q, q+G, q+2G ... and the
  // operands of the NEXT tile's first K step are already in flight while the current tile's epilogue
  // (residual reads, stores) runs -- that overlap is what the HBM-bound shallow-K layers lack otherwise.
  const int tiles_n = (p.Cout * (p.mode != 0 ? 4 : 1) + BN - 1) / BN;
  const int ntiles = tiles_n * ((M + BM - 1) / BM);
  const int G = gridDim.x;
  int q = blockIdx.x;
  if (q >= ntiles) return;

  const int lrow = lane >> 3;      // row inside an 8-row glds piece
  const int lchk = lane & 7;       // 16-byte slot inside the 128-byte LDS row
  const half_t* aptr[PA];
  const half_t* wptr[PW];
  int m0 = 0, n0 = 0;
  // XCD-aware tile order: positions q and q+8 share an XCD (L2); each XCD gets a contiguous run of logical
  // tiles, channel tiles fastest, so tiles that re-read one activation panel hit the same L2.
  // per-lane source pointers of the activation rows this lane stages; `second` selects ConvParams::in2
  auto setup_acts = [&](bool second) {
#pragma unroll
    for (int ps = 0; ps < PA; ++ps) {
      int m = m0 + ps * NW * 8 + wave * 8 + lrow;
      if (m >= M) m = M - 1;
      const int x2 = m % p.Wo;
      const int t = m / p.Wo;
      const int y = t % p.Ho;
      const int n = t / p.Ho;
      // LDS slot lchk of row r holds data chunk (lchk ^ (r & 7)); r & 7 == lrow here.
      const int dch = lchk ^ lrow;      // data chunk this lane stages
      if (second) {
        const long long base =
            ((long long)(n * p.in2_Hp + y * p.stride2 + p.in2_off) * p.in2_Wp + x2 * p.stride2 + p.in2_off) * p.in2_Cs;
        if constexpr (SPLIT) aptr[ps] = p.in2 + base + (dch & 3) * 8 + (dch >> 2) * p.in2_lo;
        else aptr[ps] = p.in2 + base + dch * 8;
      } else {
        const long long base =
            ((long long)(n * p.in_Hp + y * p.stride + p.in_off) * p.in_Wp + x2 * p.stride + p.in_off) * p.in_Cs;
        if constexpr (SPLIT && !SMALLC) aptr[ps] = p.in + base + (dch & 3) * 8 + (dch >> 2) * p.in_lo;
        else aptr[ps] = p.in + base + (SMALLC ? 0 : (dch * 8));
      }
    }
  };
  auto setup_tile = [&](int qq) {
    const int qn = ntiles >> 3, r = ntiles & 7, x = qq & 7;
    const int L = (x < r ? x * (qn + 1) : r * (qn + 1) + (x - r) * qn) + (qq >> 3);
    const int tile_n = L % tiles_n;
    const int tile_m = L / tiles_n;
    m0 = tile_m * BM;
    n0 = tile_n * BN;
    setup_acts(false);
#pragma unroll
    for (int ps = 0; ps < PW; ++ps) {
      const int row = ps * NW * 8 + wave * 8 + lrow;
      const int key = (row & 3) | (((row / (4 * MI)) & 1) << 2);
      const int rr = row < BN ? row : BN - 1;
      if constexpr (SPLIT && !SMALLC) wptr[ps] = p.w + (long long)(n0 + rr) * p.Kpad + ((lchk ^ key) & 3) * 8 + ((lchk ^ key) >> 2) * p.w_lo;
      else wptr[ps] = p.w + (long long)(n0 + rr) * p.Kpad + (lchk ^ key) * 8;
    }
  };

  constexpr int KSH = (SPLIT && !SMALLC) ? 5 : 6;      // log2 of the channels a K step covers
  constexpr int KST = 1 << KSH;
  const int nkb = SMALLC ? (p.Kpad >> 6) : (p.KH * p.KW * (p.Cin >> KSH) + (p.in2 ? (p.Cin2 >> KSH) : 0));
  const int nk = (SPLIT && SMALLC) ? 3 * nkb : nkb;
  const int nst = p.stages == 1 ? 1 : 2;   // LDS K-step buffers: 1 = shallow-K layers (more workgroups per CU)
  int* koff_s = (int*)(smem + nst * T::STAGE);
  if constexpr (SMALLC) {
    for (int i = tid; i < (p.Kpad >> 3); i += T::NT) koff_s[i] = p.koff[i];
    __syncthreads();
  }

  int w_koff = 0;               // element offset of the current K step inside a weight row (set by next_off())
  long long a_pl = 0, w_pl = 0; // SPLIT && SMALLC: plane offsets (elements) of the current pass's operands (0 = hi plane)
  auto stage = [&](int buf, int t, int a_off) {
    char* abase = smem + buf * T::STAGE;
    char* wbase = abase + BM * 128;
    if constexpr (SPLIT && SMALLC) {      // the stem: pass outermost (hi.hi, hi.lo, lo.hi over the whole padded K)
      const int pass = t / nkb;
      t -= pass * nkb;
      a_pl = pass == 1 ? p.in_lo : 0;
      w_pl = pass == 2 ? p.w_lo : 0;
    }
#pragma unroll
    for (int ps = 0; ps < PA; ++ps) {
      const half_t* g;
      if constexpr (SMALLC) {
        g = aptr[ps] + koff_s[t * 8 + (lchk ^ lrow)];
      } else {
        g = aptr[ps] + a_off;
      }
      if constexpr (SPLIT && SMALLC) g += a_pl;
      char* dst = abase + (ps * NW * 8 + wave * 8) * 128;
      if constexpr (GLDS) {
        glds16(g, dst);
      } else {
        *(half8*)(dst + lane * 16) = *(const half8*)g;
      }
    }
#pragma unroll
    for (int ps = 0; ps < PW; ++ps) {
      if (ps * NW * 8 + wave * 8 < BN) {   // wave-uniform
        const half_t* g = wptr[ps] + (SMALLC ? t * 64 : w_koff);
        if constexpr (SPLIT && SMALLC) g += w_pl;
        char* dst = wbase + (ps * NW * 8 + wave * 8) * 128;
        if constexpr (GLDS) {
          glds16(g, dst);
        } else {
          *(half8*)(dst + lane * 16) = *(const half8*)g;
        }
      }
    }
  };

  // ---- fragment read offsets (bytes inside a stage) ---------------------------------------
  const int fi = lane & 15;        // MFMA row (weights) / column (pixels) index of this lane
  const int fq = lane >> 4;        // k-slice of this lane
  const int fkey = lane & 7;       // == key of every row this lane reads
  int w_off[MI], x_off[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = wch * MI * 16 + (fi >> 2) * 4 * MI + i * 4 + (fi & 3);
    w_off[i] = BM * 128 + row * 128;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) x_off[j] = (wpx * NJ * 16 + j * 16 + fi) * 128;
  const int c0_off = ((fq) ^ fkey) * 16;        // kk = 0
  const int c1_off = ((4 + fq) ^ fkey) * 16;    // kk = 1

  int kh = 0, kw = 0, c0 = 0;   // position of the NEXT K step to stage
  // K-step order: 64-channel slice OUTER, filter taps INNER.  All KH*KW taps of one channel slice touch the
  // same ~(rows+2) x W x 128 B of the input, so a tile's live footprint between re-reads is 1/(Cin/64) of the
  // taps-outer order and stays L2-resident (256-ch 3x3 @200x200: 85 KB instead of 338 KB per tile, 32 tiles
  // per XCD against 4 MB of L2).  The weight row is addressed by (tap, slice), so its memory layout is unchanged.
  auto next_off = [&]() {
    if (c0 >= p.Cin) {               // second source (1x1 taps): only reached when p.in2 is set (nk counts its steps)
      if (c0 == p.Cin) setup_acts(true);     // the first source's pointers are dead from here on: reuse the registers
      const int off = c0 - p.Cin;
      w_koff = p.KH * p.KW * p.Cin + off;
      c0 += KST;
      return off;
    }
    const int off = (kh * p.in_Wp + kw) * p.in_Cs + c0;
    w_koff = (kh * p.KW + kw) * p.Cin + c0;
    if (++kw == p.KW) {
      kw = 0;
      if (++kh == p.KH) { kh = 0; c0 += KST; }
    }
    return off;
  };

  setup_tile(q);
  stage(0, 0, SMALLC ? 0 : next_off());
  int gs = 0;                    // running K-step count: LDS buffer of step gs is (gs & 1) when double buffered
  constexpr int STORES_F16 = MI % 2 == 0 ? MI / 2 : MI;   // store instructions per pixel of the fp16 epilogue
  constexpr int STORES_F32 = MI;                          // ... of the fp32-out epilogue
  int stores_in_flight = 0;      // stores this wave issued in the previous tile's epilogue (0 = unknown / first tile)
  // (256x256 tile: issuing the next step's LDS-DMA pieces between the MFMA groups with sched_group_barrier was
  //  tried and measured 1-6 % slower than issuing them right after the barrier.  Fetching the residual tile
  //  together with the operands was tried too: no gain, 32-64 VGPRs.)
  while (true) {
    const int m0c = m0, n0c = n0;      // the tile being computed (setup_tile() below moves m0/n0 to the next one)
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < nk; ++t) {
      if (PERSIST && t == 0 && stores_in_flight) {
        // Follow-on tile of a persistent workgroup: the previous tile's epilogue stores are the YOUNGEST
        // vector-memory operations of this wave and need not be waited for; everything older (the prefetched
        // operands of this tile) must have landed.  vmcnt counts loads, stores and LDS-DMA in issue order, so a
        // counted wait does exactly that.  __syncthreads() would drain the stores (its fence waits vmcnt(0)),
        // hence the raw barrier.
        if (stores_in_flight == NJ * STORES_F16) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ * STORES_F16) : "memory");
        else if (stores_in_flight == NJ * STORES_F32) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ * STORES_F32) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (nst == 2) {
        if (t + 1 < nk) {
          stage((gs + 1) & 1, t + 1, SMALLC ? 0 : next_off());
        } else if (PERSIST && q + G < ntiles) {     // last K step of this tile: start on the next tile
          setup_tile(q + G);
          kh = 0; kw = 0; c0 = 0;
          stage((gs + 1) & 1, 0, SMALLC ? 0 : next_off());
        }
      }
      const char* sb = smem + (nst == 2 ? (gs & 1) : 0) * T::STAGE;
      if constexpr (SPLIT && !SMALLC) {
        // chunks 0-3 of a row = the hi halfs, 4-7 = the lo halfs of the step's 32 channels: c0_off reads this lane's hi fragment, c1_off its lo one
        half8 wh[MI], wl[MI], xh[NJ], xl[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) { wh[i] = *(const half8*)(sb + w_off[i] + c0_off); wl[i] = *(const half8*)(sb + w_off[i] + c1_off); }
#pragma unroll
        for (int j = 0; j < NJ; ++j) { xh[j] = *(const half8*)(sb + x_off[j] + c0_off); xl[j] = *(const half8*)(sb + x_off[j] + c1_off); }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh[j], acc[i][j], 0, 0, 0);
      } else if constexpr (NJ == 8) {
        // 8 waves in lockstep behind one barrier: all of them read fragments at the same time, so the second
        // half-step's fragments are requested before the first half-step's MFMAs instead of after them.
        half8 wf0[MI], xf0[NJ], wf1[MI], xf1[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) wf0[i] = *(const half8*)(sb + w_off[i] + c0_off);
#pragma unroll
        for (int j = 0; j < NJ; ++j) xf0[j] = *(const half8*)(sb + x_off[j] + c0_off);
#pragma unroll
        for (int i = 0; i < MI; ++i) wf1[i] = *(const half8*)(sb + w_off[i] + c1_off);
#pragma unroll
        for (int j = 0; j < NJ; ++j) xf1[j] = *(const half8*)(sb + x_off[j] + c1_off);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[i], xf0[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[i], xf1[j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int co = kk ? c1_off : c0_off;
          half8 wf[MI], xf[NJ];
#pragma unroll
          for (int i = 0; i < MI; ++i) wf[i] = *(const half8*)(sb + w_off[i] + co);
#pragma unroll
          for (int j = 0; j < NJ; ++j) xf[j] = *(const half8*)(sb + x_off[j] + co);
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
      }
      if (nst == 1 && t + 1 < nk) {
        __syncthreads();                               // everyone done reading the only buffer
        stage(0, t + 1, SMALLC ? 0 : next_off());
      }
      ++gs;
    }

    // ---- epilogue: lane holds channels cb .. cb+4*MI-1 of pixel (nj, fi) ------------------------
    const int ch_local = wch * MI * 16 + fq * 4 * MI;
    const int crow = n0c + ch_local;                 // row in the (possibly 4x grouped) weight matrix
    int g = 0, cb = crow;
    if (p.mode != 0) { g = crow / p.Cout; cb = crow % p.Cout; }
    float bias[4 * MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const f32x4 b4 = *(const f32x4*)(p.bias + crow + i * 4);
      bias[i * 4 + 0] = b4[0]; bias[i * 4 + 1] = b4[1]; bias[i * 4 + 2] = b4[2]; bias[i * 4 + 3] = b4[3];
    }
    float wsc[SPLIT ? 4 * MI : 1];
    if constexpr (SPLIT) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const f32x4 s4 = *(const f32x4*)(p.wscale + crow + i * 4);
        wsc[i * 4 + 0] = s4[0]; wsc[i * 4 + 1] = s4[1]; wsc[i * 4 + 2] = s4[2]; wsc[i * 4 + 3] = s4[3];
      }
    }
    float dotp[NJ];
    long long dot_idx[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { dotp[j] = 0.f; dot_idx[j] = -1; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int m = m0c + wpx * NJ * 16 + j * 16 + fi;
      if (m >= M) continue;
      const int x = m % p.Wo;
      const int t = m / p.Wo;
      const int y = t % p.Ho;
      const int n = t / p.Ho;
      int oy = y, ox = x;
      if (p.mode != 0) { oy = 2 * y + (g >> 1); ox = 2 * x + (g & 1); }
      else if (TRAIN && p.out_stride > 1) { oy = y * p.out_stride; ox = x * p.out_stride; }
      const long long opix = (long long)(n * p.out_Hp + oy + p.out_pad) * p.out_Wp + ox + p.out_pad;
      float v[4 * MI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (SPLIT) v[i * 4 + r] = acc[i][j][r] * wsc[i * 4 + r] + bias[i * 4 + r];
          else v[i * 4 + r] = acc[i][j][r] + bias[i * 4 + r];
        }
      if (p.res) {
        const half_t* rp = p.res + opix * p.out_Cs + cb;
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
        const half_t* up = p.up + upix * p.up_Cs + cb;
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
      if (TRAIN && p.down) {      // backward of the nearest 2x upsample: add the 2x2 block of the finer gradient map
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
          const long long dpix = (long long)(n * p.down_Hp + 2 * y + (dd >> 1) + p.down_pad) * p.down_Wp + 2 * x + (dd & 1) + p.down_pad;
          const half_t* dp = p.down + dpix * p.down_Cs + cb;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const half4 h = *(const half4*)(dp + i * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i * 4 + r] += (float)h[r];
          }
        }
      }
      if (TRAIN && p.res32) {
        const float* rp = p.res32 + opix * p.out_Cs + cb;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const f32x4 h = *(const f32x4*)(rp + i * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i * 4 + r] += h[r];
        }
      }
      if (TRAIN && p.mask) {      // ReLU backward
        const half_t* mp = p.mask + opix * p.out_Cs + cb;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const half4 h = *(const half4*)(mp + i * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i * 4 + r] = (float)h[r] > 0.f ? v[i * 4 + r] : 0.f;
        }
      }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 4 * MI; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      }
      if (p.mode == 2) {
        // fused mask predictor: partial dot product of this lane's 4*MI channels with the predicted class' weights
        const int slot = p.dot_slot[n];
        const float* wv = p.dot_w + (long long)p.dot_cls[slot] * p.Cout + cb;
        float sdot = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const f32x4 w4 = *(const f32x4*)(wv + i * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) sdot += v[i * 4 + r] * w4[r];
        }
        dotp[j] = sdot;
        dot_idx[j] = (long long)slot * (4 * p.Ho * p.Wo) + (long long)oy * (2 * p.Wo) + ox;
      } else if (p.out_f32) {
        float* op = (float*)p.out + opix * p.out_Cs + cb;
#pragma unroll
        for (int i = 0; i < MI; ++i) *(f32x4*)(op + i * 4) = f32x4{v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]};
      } else {
        half_t* op = (half_t*)p.out + opix * p.out_Cs + cb;
        if constexpr (MI % 2 == 0) {
#pragma unroll
          for (int i = 0; i < MI; i += 2) {          // one 16-byte store per 8 channels (STORES_F16 per pixel)
            half8 h, l;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              float f = v[i * 4 + r];
              f = f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f);
              h[r] = (half_t)f;
              if constexpr (SPLIT) l[r] = (half_t)(f - (float)h[r]);
            }
            *(half8*)(op + i * 4) = h;
            if constexpr (SPLIT) *(half8*)(op + p.out_lo + i * 4) = l;
          }
        } else {
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            half4 h, l;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float f = v[i * 4 + r];
              f = f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f);
              h[r] = (half_t)f;
              if constexpr (SPLIT) l[r] = (half_t)(f - (float)h[r]);
            }
            *(half4*)(op + i * 4) = h;
            if constexpr (SPLIT) *(half4*)(op + p.out_lo + i * 4) = l;
          }
        }
      }
    }
    if (p.mode == 2) {
      // Reduce the per-lane partials to one value per output pixel in a FIXED order (bitwise reproducible):
      // lanes of one pixel (k-slices fq = 0..3) by shuffles, the WCH channel waves through LDS; the two
      // channel tiles of a (dy,dx) group live in different workgroups and meet in one float atomicAdd each
      // on a zero-initialised word -- two addends commute exactly.  (mode 2 is launched one workgroup per
      // tile, so no next-tile prefetch is in flight into the LDS words used here.)
      float* red = (float*)smem;      // [WCH-1][WPX][NJ][16]
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float v0 = dotp[j];
        v0 += __shfl_xor(v0, 16);
        v0 += __shfl_xor(v0, 32);
        dotp[j] = v0;
        if (wch > 0 && fq == 0) red[(((wch - 1) * WPX + wpx) * NJ + j) * 16 + fi] = v0;
      }
      __syncthreads();
      if (wch == 0 && fq == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float v0 = dotp[j];
#pragma unroll
          for (int c = 1; c < WCH; ++c) v0 += red[(((c - 1) * WPX + wpx) * NJ + j) * 16 + fi];
          if (dot_idx[j] >= 0) atomicAdd(p.dot_out + dot_idx[j], v0);
        }
      }
    }

    // every pixel row of this wave was valid (no `continue` above) => the store count is exact
    stores_in_flight = (p.mode != 2 && m0c + BM <= M) ? NJ * (p.out_f32 ? STORES_F32 : STORES_F16) : 0;
    if constexpr (!PERSIST) break;
    q += G;
    if (q >= ntiles) break;
    if (nst == 1) {                    // single buffer: no cross-tile prefetch; restart on the next tile
      setup_tile(q);
      kh = 0; kw = 0; c0 = 0;
      __syncthreads();
      stage(0, 0, SMALLC ? 0 : next_off());
    }
  }
}

template <int WPX, int WCH, int MI, int NJ, bool SMALLC, bool CAN_PERSIST>
int launch_variant(const ConvParams& p, hipStream_t stream, int use_glds) {
  using T = Tile<WPX, WCH, MI, NJ>;
  const int rows = p.Cout * (p.mode != 0 ? 4 : 1);
  const int tiles_n = cdiv(rows, T::BN);
  const int tiles_m = cdiv(p.M, T::BM);
  const long long nblk = (long long)tiles_n * tiles_m;
  RS_CHECK(nblk > 0 && nblk < (1ll << 31), RS_ERR_ARG, "conv: bad grid %lld", nblk);
  RS_CHECK(rows % T::BN == 0 || T::BN <= 16, RS_ERR_ARG, "conv: Cout rows %d not a multiple of tile %d", rows, T::BN);
  const int lds = (p.stages == 1 ? 1 : 2) * T::STAGE + 1024;
  // Persistent launch (p.persist = workgroups per CU, -1 = as many as the LDS admits): only with double
  // buffering, never for the fused-dot mode, and only for the variants instantiated with PERSIST.
  int bpc = CAN_PERSIST ? p.persist : 0;
  if (bpc < 0) bpc = (160 * 1024) / lds < 1 ? 1 : (160 * 1024) / lds;
  const bool persistent = bpc > 0 && p.stages != 1 && p.mode != 2 && use_glds && nblk > 256ll * bpc &&
                          !(p.down || p.res32 || p.mask || p.out_stride > 1);
  const bool train = p.down || p.res32 || p.mask || p.out_stride > 1;
  const void* k;
  if (p.split) {
    RS_CHECK(use_glds > 0 && !train && p.wscale, RS_ERR_UNSUPPORTED, "conv: the split-operand mode needs LDS-DMA staging, the row scales and no training epilogue");
    k = (const void*)conv_igemm_kernel<WPX, WCH, MI, NJ, SMALLC, true, false, false, true>;
  } else if (train) {
    RS_CHECK(use_glds && !SMALLC, RS_ERR_UNSUPPORTED, "conv: training epilogue options need LDS-DMA staging and Cin >= 64");
    if constexpr (!SMALLC) k = (const void*)conv_igemm_kernel<WPX, WCH, MI, NJ, false, true, false, true>;
    else k = nullptr;
  } else if constexpr (CAN_PERSIST) {
    k = persistent ? (const void*)conv_igemm_kernel<WPX, WCH, MI, NJ, SMALLC, true, true>
                   : (use_glds ? (const void*)conv_igemm_kernel<WPX, WCH, MI, NJ, SMALLC, true, false>
                               : (const void*)conv_igemm_kernel<WPX, WCH, MI, NJ, SMALLC, false, false>);
  } else {
    k = use_glds ? (const void*)conv_igemm_kernel<WPX, WCH, MI, NJ, SMALLC, true, false>
                 : (const void*)conv_igemm_kernel<WPX, WCH, MI, NJ, SMALLC, false, false>;
  }
  static bool attr[5] = {false, false, false, false, false};
  const int ai = p.split ? 4 : (train ? 3 : (persistent ? 2 : (use_glds ? 1 : 0)));
  if (!attr[ai]) {
    RS_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS));
    attr[ai] = true;
  }
  const unsigned grid = (persistent && !p.split) ? 256u * (unsigned)bpc : (unsigned)nblk;
  ConvParams pc = p;
  void* args[] = {&pc};
  RS_HIP(hipLaunchKernel(k, dim3(grid), dim3(T::NT), args, lds, stream));
  return RS_OK;
}

}  // namespace

// Tile variant picked by the most recent launch_conv() on this thread (-1 = fp32 validation kernel);
// the engine tags its stages with it so that bench.py can attribute time and FLOPs per kernel symbol.
thread_local int g_last_conv_variant = -2;

// Tile variants: 0 = 128x128, 1 = 256x64, 2 = 256x16 fp32-out (small heads), 3 = 256x128, 4 = 256x256 (conv_igemm),
// 5 = small-Cin stem (reported only), 7 = 64x128, 8 = 128x64, 9 = 32x128, 10 = 64x256, 12 = conv_deep 256x256, 14 = 128x256.
//
// conv_choose_variant() is the WHOLE dispatch rule -- a pure function of the layer shape (M = batch * Ho * Wo, so of the
// batch size too) and of the debug switches; launch_conv() only validates and launches what it returns.  It is exported as
// rs_op_conv_variant so that tests can enumerate the choice per layer and batch size without a GPU
// (tests/test_host_cpu.py::test_conv_variant_table).  Choices come from the offline sweep over every layer shape of the
// batch-16 forward (tools/ubench/tune_conv.py, profiles/r01/conv_tile_sweep.txt).
int conv_choose_variant(ConvParams& p, int force_variant, int use_glds) {
  const RsDebug& D = rs_debug();
  const int nk2 = p.in2 ? (p.Cin2 >> 6) : 0;   // K steps of the second source
  const bool smallc = p.Cin < 64;
  // split-operand mode: K steps of 32 channels x (hi, lo) with three MFMA blocks each = three times the matrix work of an fp16 layer over
  // twice the steps; the tile rule treats it as a layer of three times the depth
  const int nk = (p.split ? 3 : 1) * (smallc ? (p.Kpad >> 6) : p.KH * p.KW * (p.Cin >> 6) + nk2);
  if (p.stages == 0) {
    // Shallow-K layers (1x1 convs of res2/res3, laterals) are HBM-bound and gain nothing from a second LDS buffer; a
    // single buffer halves the LDS footprint so 4 workgroups fit per CU and their loads/epilogues overlap each other.
    // (RS_CONV_PERSIST >= 1: persistent + double buffered instead -- correct, measured not faster: 0.221 vs 0.225 ms on
    // res2 conv3, slower on conv1; these layers sit at ~3.3 TB/s either way.)
    // (split-operand mode, nk = three times the fp16 layer's: res2.x.conv3 has two K steps of 32 channels x (hi, lo) and would take one buffer by its
    // step count; measured slower on one buffer than on two, 339 against 329 us, res3.x.conv3 213 against 203: the rule stays on the tripled count)
    const bool shallow = nk <= D.conv_single_stage_nk;
    if (D.conv_persist >= 1 && shallow && p.mode != 2 && !p.split) { p.stages = 2; p.persist = 2; }
    else { p.stages = shallow ? 1 : 2; p.persist = D.conv_persist >= 2 && p.mode != 2 ? -1 : 0; }
  }
  if (force_variant >= 0) return force_variant;
  const int rows = p.Cout * (p.mode != 0 ? 4 : 1);
  const int nkd = smallc ? 0 : nk;
  const long long tiles0 = (long long)cdiv(p.M, 128) * (rows / 128 > 0 ? rows / 128 : 1);
  const long long tiles4 = (long long)cdiv(p.M, 256) * (rows / 256 > 0 ? rows / 256 : 1);
  if (rows <= 16) return 2;
  if (!D.conv_tuned || p.mode != 0 || smallc) return rows % 128 == 0 ? 0 : 1;
  if (rows % 128 != 0) return 8;                                          // Cout = 64: 128x64 beats 256x64 everywhere
  // Deep K on a map whose pixels fill the 256 CUs badly in 256-pixel tiles (one workgroup per CU, so a launch runs in whole rounds):
  // conv_deep with 160 / 192 / 224-pixel tiles when that saves a round or shortens the only one.  A round costs about (pixels / 32 + 15)
  // units -- per K step the barrier chain pays the LDS-DMA latency whatever the tile height (tools/ubench/conv_shapes.py: 50 x 50 x 16
  // images, 3x3 256 -> 256: 58.8 us on the 64x128 tile, 52.8 in 157 tiles of 256 pixels, 46.0 in 250 tiles of 160).
  if (rows % 256 == 0 && nkd >= 8 && D.conv_deep && D.deep_tile_px && use_glds > 0 && !p.in2 && !p.m_count && !p.down && !p.res32 && !p.mask && p.out_stride <= 1) {
    const int ncu = 256;
    auto cost = [&](int nj) {
      const long long t = (long long)cdiv(p.M, 32 * nj) * (rows / 256);
      const long long r = t % ncu;
      const double rounds = (double)(t / ncu) + (r == 0 ? 0.0 : (nj == 8 && D.deep_tail && 2 * r <= ncu ? 0.75 : 1.0));   // 256-pixel tiles split a short last round
      return rounds * (nj + 15);
    };
    int best = 8;
    double cb = cost(8);
    for (int nj = 7; nj >= 5; --nj) {
      const long long t = (long long)cdiv(p.M, 32 * nj) * (rows / 256);
      if (4 * t >= 3 * ncu && cost(nj) < 0.95 * cb && cost(nj) < (best == 8 ? 1e30 : cost(best))) best = nj;
    }
    if (best != 8) return 10 + best;                                      // 15 / 16 / 17
  }
  // Split-operand mode, deep K on few pixels (res5 at batch 16: 80-320 tiles of 256 x 256): three MFMA blocks per staged byte make conv_deep's big tile
  // the fastest from a quarter of the chip up, where the fp16 layers are bound by the weight traffic of every tile shape (tools/ubench/split_shapes.py,
  // profiles/r04/split_shapes_b16.txt: res5.x.conv2 156 us against 195 on the 64 x 128 tile, res5.x.conv1 77 against 91)
  if (p.split && rows % 256 == 0 && nkd >= 8 && tiles4 >= 64 && tiles4 < 240 && D.conv_deep && use_glds > 0 && !p.in2 && p.out_stride <= 1) return 12;
  // conv3 + projection shortcut over two K sources (conv_igemm only).  tools/ubench/dual_shapes.py, batch 16: res3.0 (1250 tiles of
  // 256x256) 106 us on 256x256 vs 129 on 128x128; res4.0 (628) 88 vs 88; res5.0 (320) 91.5 vs 79
  if (p.in2 && rows % 256 == 0 && nkd >= 6) return tiles4 >= 600 ? 4 : (tiles0 < 1250 ? 7 : 0);
  if (rows % 256 == 0 && nkd >= 8 && tiles4 >= 240)                       // deep K, many rows: 256x256 halves the L2->LDS bytes per FLOP
    return (D.conv_deep && use_glds > 0 && !p.in2 && p.out_stride <= 1) ? 12 : 4;   // conv_deep incl. the backward epilogue (down / res32 / mask)
  // HBM-bound 1x1 layers on big maps: all 256 channels per workgroup, so every activation row is read once; 128 pixels per
  // workgroup restage the 128 KB weight matrix half as often as 64 (fpn_lateral2 0.241 -> 0.210 ms, fused deconv 0.338 -> 0.267 ms)
  // (res4.x.conv3 at batch 16, 40 000 pixels: 38.9 us against 44.0 on the 64x128 tile, tools/ubench/conv_shapes.py)
  // ... and where a workgroup walks about five tiles or more, the persistent form with the weights in registers (conv_wreg.hip: 32-pixel
  // tiles, two independent workgroups per CU).  tools/ubench/wreg_shapes.py, layers with their epilogue operands, us against the best
  // conv_igemm tile: fpn_lateral2 at batch 16 / 8 / 3 / 2 / 1: 152 / 82 / 34.1 / 26.2 / 19.6 against 210 / 108 / 34.5 / 26.3 / 16.4;
  // res4.x.conv3 at batch 16 / 8 / 4: 40.0 / 26.1 / 20.0 against 45.4 / 27.0 / 16.8 -- the break-even is a walk of ~4.8 tiles per workgroup
  // (its prologue loads 128 KB of weights into registers)
  if (D.conv_wreg && !p.split && use_glds > 0 && conv_wreg_ok(p) && (long long)cdiv(p.M, 32) * (p.Cout >> 8) * 5 >= 48ll * rs_device_cu_count()) return 22;
  if (rows % 256 == 0 && nkd <= 4 && p.M >= 40000) return D.conv_wide_px == 64 ? 10 : 14;
  if (nkd <= 4 || tiles0 < 1250) return 7;                                // few tiles or shallow K: 64x128 keeps more workgroups in flight
  return 0;
}

int launch_conv(const ConvParams& p_in, hipStream_t stream, int force_variant, int use_glds) {
  if (use_glds < 0) {                   // fp32 reference-precision mode (ref_f32.hip)
    RS_CHECK(!p_in.in2, RS_ERR_UNSUPPORTED, "conv: the fp32 kernel has no second K source");
    g_last_conv_variant = -1;
    return launch_conv_f32(p_in, stream, use_glds == -2, force_variant);      // -2: the VALU cross-check kernel
  }
  ConvParams p = p_in;
  const int nk2 = p.in2 ? (p.Cin2 >> 6) : 0;
  const bool train_opts = p.down || p.res32 || p.mask || p.out_stride > 1;   // only conv_igemm_kernel / conv_deep implement these
  RS_CHECK(p.M > 0, RS_ERR_ARG, "conv: M=%d", p.M);
  RS_CHECK(p.Kpad % 64 == 0, RS_ERR_ARG, "conv: Kpad %d not a multiple of 64", p.Kpad);
  const bool smallc = p.Cin < 64;
  if (smallc) {
    RS_CHECK((p.Cin == 8 || p.Cin == 4) && p.koff != nullptr, RS_ERR_ARG, "conv: small-Cin path needs Cin 4 or 8 and a koff table");
  } else {
    RS_CHECK(p.Cin % 64 == 0, RS_ERR_ARG, "conv: Cin %d not a multiple of 64", p.Cin);
    RS_CHECK(p.KH * p.KW * p.Cin + nk2 * 64 <= p.Kpad, RS_ERR_ARG, "conv: K exceeds Kpad");
  }
  if (p.in2) RS_CHECK(!smallc && p.mode == 0 && p.Cin2 % 64 == 0 && p.Cin2 > 0 && p.stride2 >= 1, RS_ERR_ARG, "conv: bad second K source (Cin2 %d)", p.Cin2);
  const int rows = p.Cout * (p.mode != 0 ? 4 : 1);
  int v = conv_choose_variant(p, force_variant, use_glds);
  if (v == 22 && p.mode == 2 && !conv_wreg_ok(p)) v = 14;     // the fused mask predictor beyond conv_wreg's LDS budget (thousands of entries): conv_igemm's tile
  RS_CHECK(!(train_opts && (p.mode != 0 || (v == 12 && p.out_stride > 1) || (v >= 15 && v <= 20))), RS_ERR_UNSUPPORTED, "conv: training epilogue options need mode 0 (and no scatter on conv_deep)");
  if (v == 22 || v == 23 || v == 25 || v == 26) {   // persistent 1x1 with the weights in registers (conv_wreg.hip): 22 = the form that ships (RS_WREG_WAVES,
    // default 32-pixel tiles with two workgroups per CU), 25 = that form by name, 26 / 23 = 64-pixel tiles with four / eight waves
    RS_CHECK(conv_wreg_ok(p) && !train_opts, RS_ERR_UNSUPPORTED, "conv: variant %d takes 1x1 / Cin 256 / Cout %% 256 == 0 inference layers only", v);
    g_last_conv_variant = v;
    return launch_conv_wreg(p, stream, v == 23 ? 8 : (v == 25 ? 2 : (v == 26 ? 4 : 0)));
  }
  if (v == 12 || (v >= 15 && v <= 20)) {     // 256x256 with 3 activation stages / 2 weight stages (conv_deep.hip); 15 / 16 / 17: 160 / 192 / 224 pixels, 18 / 19 / 20: 64 / 96 / 128
    RS_CHECK(!p.in2, RS_ERR_UNSUPPORTED, "conv: variant %d has no second K source", v);
    g_last_conv_variant = v;
    return launch_conv_deep(p, stream, v == 12 ? 256 : (v <= 17 ? 160 + 32 * (v - 15) : 64 + 32 * (v - 18)));
  }
  RS_CHECK(!(p.mode != 0 && p.Cout % 128 != 0), RS_ERR_ARG, "deconv needs Cout %% 128 == 0");
  RS_CHECK(!(p.mode == 2 && !(p.dot_w && p.dot_cls && p.dot_slot && p.dot_out && (v == 0 || v == 10 || v == 14))), RS_ERR_ARG, "fused mask predictor needs its pointers and the 128x128, 64x256 or 128x256 tile");
  g_last_conv_variant = smallc ? 5 : v;
  if (smallc) {
    RS_CHECK(v == 1, RS_ERR_ARG, "conv: small-Cin path is built for the 256x64 tile only (Cout=%d)", p.Cout);
    if (rs_debug().stem_small_tile) return launch_variant<4, 1, 4, 2, true, false>(p, stream, use_glds);     // 128x64
    return launch_variant<4, 1, 4, 4, true, false>(p, stream, use_glds);
  }
  switch (v) {
    case 0:
      RS_CHECK(rows % 128 == 0, RS_ERR_ARG, "conv: variant 0 needs Cout %% 128 == 0");
      return launch_variant<2, 2, 4, 4, false, true>(p, stream, use_glds);
    case 1:
      RS_CHECK(rows % 64 == 0, RS_ERR_ARG, "conv: variant 1 needs Cout %% 64 == 0");
      return launch_variant<4, 1, 4, 4, false, true>(p, stream, use_glds);
    case 2:
      RS_CHECK(rows % 16 == 0 && p.out_f32, RS_ERR_ARG, "conv: variant 2 is the 16-channel-tile fp32-out head kernel");
      return launch_variant<4, 1, 1, 4, false, true>(p, stream, use_glds);
    case 7:   // 64 px x 128 ch tile (wave tile 32x64): half the accumulators, more workgroups per CU
      RS_CHECK(rows % 128 == 0, RS_ERR_ARG, "conv: variant 7 needs Cout %% 128 == 0");
      return launch_variant<2, 2, 4, 2, false, false>(p, stream, use_glds);
    case 9:   // 32 px x 128 ch tile (wave tile 16x64)
      RS_CHECK(rows % 128 == 0, RS_ERR_ARG, "conv: variant 9 needs Cout %% 128 == 0");
      return launch_variant<2, 2, 4, 1, false, false>(p, stream, use_glds);
    case 10:  // 64 px x 256 ch tile, 8 waves (wave tile 32x64): every activation row read once
      RS_CHECK(rows % 256 == 0, RS_ERR_ARG, "conv: variant 10 needs Cout %% 256 == 0");
      return launch_variant<2, 4, 4, 2, false, false>(p, stream, use_glds);
    case 14:  // 128 px x 256 ch tile, 8 waves (wave tile 64x64): half the weight restaging per pixel of the 64x256 tile
      RS_CHECK(rows % 256 == 0, RS_ERR_ARG, "conv: variant 14 needs Cout %% 256 == 0");
      return launch_variant<2, 4, 4, 4, false, false>(p, stream, use_glds);
    case 8:   // 128 px x 64 ch tile (wave tile 32x64, 4 px-waves)
      RS_CHECK(rows % 64 == 0, RS_ERR_ARG, "conv: variant 8 needs Cout %% 64 == 0");
      return launch_variant<4, 1, 4, 2, false, false>(p, stream, use_glds);
    case 3:
      RS_CHECK(rows % 128 == 0, RS_ERR_ARG, "conv: variant 3 needs Cout %% 128 == 0");
      return launch_variant<4, 2, 4, 4, false, false>(p, stream, use_glds);
    case 4:
      RS_CHECK(rows % 256 == 0, RS_ERR_ARG, "conv: variant 4 needs Cout %% 256 == 0");
      return launch_variant<2, 4, 4, 8, false, false>(p, stream, use_glds);
    default:
      rs_set_error("conv: unknown variant %d", v);
      return RS_ERR_ARG;
  }
}
