// 1x1 convolution 256 -> 256 g (g = Cout / 256) for the shallow-K layers on big maps -- fpn_lateral2, res4.x.conv3 (HBM-bound) and the mask
// head's 2x2 transposed convolution + predictor (four groups of 256 rows, EPI 3): PERSISTENT workgroups with the weights in REGISTERS and the
// whole LDS given to operand tiles in flight.  Variant 22.
//
// What bounds these layers in conv_igemm's 128 x 256 tile (variant 14): K = 256 is four K steps; a workgroup stages 16 KB of
// activations (HBM) and 32 KB of weights (L2) per step into its single 48 KB buffer, three workgroups per CU -- so at most 48 KB of
// HBM reads are in flight per CU (12 MB on the chip), two thirds of every staged byte are weights that every tile stages again, and
// a tile's epilogue reads (residual / top-down addend: ordinary loads, waited for with nothing else in flight) overlap nothing of
// their own workgroup.  Little's law at the ~3.4 us these loads take under load gives the 3.2-3.6 TB/s the stage table shows.
//
// Here a workgroup (4 waves, one per SIMD; two such workgroups per CU in the form that ships) owns ONE 256-channel group for its whole life:
//   * its wave's weight fragments -- 64 channels x 256 K as 32 MFMA A operands, 128 VGPRs -- are loaded once (after the first two
//     tiles' pieces are on their way);
//   * LDS holds two operand tiles of BM pixels: the 256-channel activation rows (BM x 512 B) AND the epilogue operand of the same pixels
//     (residual, or the coarser map's rows of the FPN top-down add: as much again), both staged by LDS-DMA in the 128-byte XOR-swizzled rows
//     of conv_igemm -- no register is ever the target of a load in flight, so nothing here depends on where hipcc puts a copy;
//   * tile k is computed while k+1 is in flight and k+2 is issued as soon as every wave has read tile k: 64-128 KB of HBM reads in
//     flight per CU;
//   * one counted s_waitcnt per tile: vmcnt counts LDS-DMA pieces and stores in issue order, and everything a wave issues after tile
//     k's pieces is known (the stores of tile k-2, tile k+1's pieces, the stores of tile k-1);
//   * with ONE wave per SIMD nothing hides a wave's own scalar work, so there is none to speak of: pixel coordinates advance by
//     carries (pix_add; the kernel's only divisions are in its prologue), bias / addend are packed adds, the clamp is one v_med3.
// The arithmetic of an output element is conv_igemm's (same fragments, same K order, same epilogue order: bias, residual, top-down
// add, ReLU, clamp, fp16 rounding): results are bit-identical to variant 14 for every finite value
// (tests/test_gpu_conv.py::test_conv1x1_register_weights_is_bit_identical_to_the_tiled_kernel); a NaN accumulator, which variant 14
// stores as NaN, is stored as -65504 here (v_med3_f32).
//
// The form that ships has 32-PIXEL tiles and TWO such workgroups per CU (template BM = 32: 64 KB of LDS and <= 256 registers each): the
// phases of a workgroup -- wait for the tile, MFMAs, epilogue -- run one after the other (see the ablations below), but two independent
// workgroups on a CU are out of phase with each other, so one's epilogue and waits run under the other's MFMAs.  fpn_lateral2 with its
// top-down add at batch 16: 210 us (variant 14) -> 165.5 (64-pixel tiles, one workgroup per CU) -> 152 us (4.86 TB/s of algorithmic
// bytes); res4.x.conv3 (1024 channels, residual) 45.4 -> 40.0 us.
//
// Measured first with 64-pixel tiles (tools/ubench/wreg_shapes.py, one box, batch 16): fpn_lateral2 with its top-down add 216 -> 167 us (3.42 -> 4.42 TB/s of
// algorithmic bytes), without the add 182 -> 148 us.  RS_WREG_DBG ablations of the 167 us: without its stores 103, without its loads
// 122, without both 63 -- the three add up, i.e. the wave's phases (wait for the tile, MFMAs, epilogue) still run one after the other;
// an eight-wave form (variant 23: two waves per SIMD, 32 pixels of the tile each) measures the same 163-166 us because one barrier
// pair per tile keeps all waves in the same phase.  Short walks lose to the prologue (128 KB of weights per workgroup): see the
// dispatch rule in conv_choose_variant.
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int MI = 4;

__device__ __forceinline__ void glds16(const half_t* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Pixel m = m0 + r as (x, y, n) without a division: (x0, y0, n0) are the coordinates of m0 and (rx, ry, rn) those of the offset r
// (rx < Wo, ry < Ho), so one carry per digit suffices.
struct Pix { int x, y, n; };
__device__ __forceinline__ Pix pix_add(int x0, int y0, int n0, int rx, int ry, int rn, int Wo, int Ho) {
  int x = x0 + rx;
  const int cx = x >= Wo;
  x -= cx ? Wo : 0;
  int y = y0 + ry + cx;
  const int cy = y >= Ho;
  y -= cy ? Ho : 0;
  return Pix{x, y, n0 + rn + cy};
}

// EPI: 0 = bias (+ ReLU) only, 1 = residual (same geometry as the output), 2 = coarser map added at (y/2, x/2) (FPN top-down),
//      3 = the mask head's 2x2 stride-2 transposed convolution + ReLU + predictor 1x1 of the predicted class (ConvParams::mode 2: the four
//          (dy, dx) groups of 256 rows are the channel groups; nothing but one float per output pixel is written)
// NWPX: 1 = four waves (one per SIMD, all 64 pixels of the tile each), 2 = eight waves (two per SIMD, 32 pixels each)
// BM: pixels per tile: 64 (one workgroup per CU), or 32 with NWPX = 1 (half the LDS and a quarter fewer registers: TWO independent workgroups
// per CU, whose phases -- wait for the tile, MFMAs, epilogue -- then overlap each other's)
template <int EPI, int NWPX, int BM>
__global__ __launch_bounds__(256 * NWPX, BM == 32 ? 2 : 1) void conv1x1_wreg_kernel(const ConvParams p, const int dbg) {
  constexpr int SLICE = BM * 128;                 // one 64-channel slice of a tile
  constexpr int HALF = 4 * SLICE;                 // 256 channels
  constexpr int NWAVE = 4 * NWPX, NJ = BM / 16 / NWPX;
  constexpr int PASSES = BM / (NWAVE * 8);        // LDS-DMA passes of NWAVE * 8 rows per tile
  constexpr bool OPND = EPI == 1 || EPI == 2;     // an epilogue operand is staged next to the activations
  constexpr int TILE = OPND ? 2 * HALF : HALF;    // activations [+ epilogue operand]
  constexpr int OPS = (OPND ? 8 : 4) * PASSES;    // LDS-DMA pieces a wave issues per tile
  constexpr int STS = 2 * NJ;                     // stores a wave issues per (full) tile
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * TILE
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wch = wave & 3, wpx = wave >> 2;      // wave = 64 channels x (64 / NWPX) pixels
  int Mv = p.M;
  if (p.m_count) { const long long mc = (long long)(*p.m_count) * p.m_mul; if (mc < Mv) Mv = (int)mc; }    // rows bounded by a device counter (mask head)
  const int M = Mv, Wo = p.Wo, Ho = p.Ho;
  const int tiles_m = (M + BM - 1) / BM;
  const int groups = EPI == 3 ? 4 : p.Cout >> 8;
  // XCD-aware order (conv_igemm.hip): workgroups b, b + 8, ... share an XCD; the `groups` workgroups that read one activation tile get
  // consecutive logical ids, i.e. the same XCD's L2
  int L;
  {
    const int G = gridDim.x, qn = G >> 3, r = G & 7, x = blockIdx.x & 7;
    L = (x < r ? x * (qn + 1) : r * (qn + 1) + (x - r) * qn) + (blockIdx.x >> 3);
  }
  const int grp = L % groups, first = L / groups, step = gridDim.x / groups;
  const int ntl = first < tiles_m ? (tiles_m - 1 - first) / step + 1 : 0;     // this workgroup's tiles: first, first + step, ...
  if (ntl == 0) return;
  const int n0 = grp * 256;
  const int fi = lane & 15, fq = lane >> 4, fkey = lane & 7;
  const int lrow = lane >> 3, lchk = lane & 7;

  // ---- pixel coordinates: the only divisions of the kernel (uniform ones for the tile walk, per-lane ones for the lane's fixed offsets)
  const int HW = Ho * Wo;
  int tx, ty, tn;                                 // coordinates of the first pixel of the tile being ISSUED (uniform)
  { const int m = first * BM; tn = m / HW; const int r = m - tn * HW; ty = r / Wo; tx = r - ty * Wo; }
  int dx, dy, dn;                                 // ... advance by `step` tiles
  { const int m = step * BM; dn = m / HW; const int r = m - dn * HW; dy = r / Wo; dx = r - dy * Wo; }
  int ex = tx, ey = ty, en = tn;                  // the same walk for the tile being STORED
  int irx[PASSES], iry[PASSES], irn[PASSES];      // this lane's rows in the LDS-DMA passes
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int r = ps * NWAVE * 8 + wave * 8 + lrow;
    irn[ps] = r / HW; const int q = r - irn[ps] * HW; iry[ps] = q / Wo; irx[ps] = q - iry[ps] * Wo;
  }
  int orx[NJ], ory[NJ], orn[NJ];                  // ... and its pixels of the output fragment
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int r = (wpx * NJ + j) * 16 + fi;
    orn[j] = r / HW; const int q = r - orn[j] * HW; ory[j] = q / Wo; orx[j] = q - ory[j] * Wo;
  }
  const int c0_off = (fq ^ fkey) * 16, c1_off = ((4 + fq) ^ fkey) * 16;
  const int ec0 = ((2 * fq) ^ fkey) * 16, ec1 = ((2 * fq + 1) ^ fkey) * 16;
  const int sw8 = (lchk ^ lrow) * 8;
  const int crow = n0 + wch * 64 + fq * 16;                       // lane holds channels crow .. crow + 15 of pixel (j, fi)

  // the LDS-DMA pieces of the tile at (tx, ty, tn), first pixel m0: activation rows, then the epilogue operand's rows
  auto issue_tile = [&](int m0, int buf) {
    if (dbg & 4) return;
    char* base = smem + buf * TILE;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int r = ps * NWAVE * 8 + wave * 8 + lrow;
      Pix q = pix_add(tx, ty, tn, irx[ps], iry[ps], irn[ps], Wo, Ho);
      if (m0 + r >= M) q = Pix{tx, ty, tn};                     // rows past the end of the map (ragged last tile): any valid address
      const half_t* src = p.in + ((long long)(q.n * p.in_Hp + q.y + p.in_off) * p.in_Wp + q.x + p.in_off) * p.in_Cs + sw8;
      char* dst = base + (ps * NWAVE * 8 + wave * 8) * 128;
#pragma unroll
      for (int s = 0; s < 4; ++s) glds16(src + s * 64, dst + s * SLICE);
      if (OPND) {
        const half_t* es;
        if (EPI == 1) es = p.res + ((long long)(q.n * p.out_Hp + q.y + p.out_pad) * p.out_Wp + q.x + p.out_pad) * p.out_Cs + n0;
        else es = p.up + ((long long)(q.n * p.up_Hp + (q.y >> 1) + p.up_pad) * p.up_Wp + (q.x >> 1) + p.up_pad) * p.up_Cs + n0;
        es += sw8;
#pragma unroll
        for (int s = 0; s < 4; ++s) glds16(es + s * 64, dst + HALF + s * SLICE);
      }
    }
    const Pix nx = pix_add(tx, ty, tn, dx, dy, dn, Wo, Ho);     // the walk goes on to this workgroup's next tile
    tx = nx.x; ty = nx.y; tn = nx.n;
  };

  issue_tile(first * BM, 0);                                     // the first two tiles are on their way before the weights are asked for
  if (ntl > 1) issue_tile((first + step) * BM, 1);

  // ---- weights: the A fragments conv_igemm reads from LDS (row permutation and k chunks of its w_off / c0_off / c1_off), once
  half8 wreg[4][2][MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = n0 + wch * 64 + (fi >> 2) * 16 + i * 4 + (fi & 3);
    const half_t* wr = p.w + (long long)row * p.Kpad + fq * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      wreg[s][0][i] = *(const half8*)(wr + s * 64);
      wreg[s][1][i] = *(const half8*)(wr + s * 64 + 32);
    }
  }
  f32x2 bias[8];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const f32x4 b4 = *(const f32x4*)(p.bias + crow + i * 4);
    bias[i * 2] = f32x2{b4[0], b4[1]}; bias[i * 2 + 1] = f32x2{b4[2], b4[3]};
  }
  // EPI 3: everything the epilogue looks up -- the predictor's K x 256 weights, and per mask-head entry its slot and predicted class -- goes
  // into LDS once, so that no ordinary global load (whose compiler-made wait would drain the operand tiles in flight) is left in the loop
  float* const red = (float*)(smem + 2 * TILE);                   // [3 channel waves][NJ][16] partial sums
  float* const wcls = red + 256;                                  // [K][256]
  int* const ent_slot = (int*)(wcls + (EPI == 3 ? p.dot_k : 0) * 256);
  const int n_ent = EPI == 3 ? p.M / (Ho * Wo) : 0;               // entry capacity
  int* const ent_cls = ent_slot + n_ent;
  if constexpr (EPI == 3) {
    for (int i = tid; i < p.dot_k * 256; i += 256 * NWPX) wcls[i] = p.dot_w[i];
    const int n_valid = (M + Ho * Wo - 1) / (Ho * Wo);
    for (int e = tid; e < n_ent; e += 256 * NWPX) {
      int sl = 0, cl = 0;
      if (e < n_valid) { sl = p.dot_slot[e]; cl = p.dot_cls[sl]; }
      ent_slot[e] = sl; ent_cls[e] = cl;
    }
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // weights, biases and the first two tiles have landed
  const float lo = -65504.f;
  int x_off[NJ], e_off[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    x_off[j] = ((wpx * NJ + j) * 16 + fi) * 128;
    // epilogue operand of this lane: channels 16 fq .. 16 fq + 15 of its wave's 64-channel slice = chunks 2 fq, 2 fq + 1 of row (j, fi)
    e_off[j] = HALF + wch * SLICE + ((wpx * NJ + j) * 16 + fi) * 128;
  }

  for (int k = 0; k < ntl; ++k) {
    // Tile k's pieces have landed once at most `newer` YOUNGER operations of this wave are outstanding.  Issue order from tile 2 on: tile
    // k's pieces [iteration k-2], the stores of tile k-2, tile k+1's pieces, the stores of tile k-1 (every tile but a workgroup's last
    // is a full one: the ragged tile is the last of the whole map, hence the last of its workgroup).  Tiles 0 and 1 landed in the prologue.
    if (k >= 2) {
      if (EPI == 3 && wch != 0) {               // the predictor sums leave through channel wave 0 alone (NJ atomics per tile): the others store nothing
        if (k + 1 < ntl) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        constexpr int ST = EPI == 3 ? NJ : STS;
        if (k + 1 < ntl) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS + 2 * ST) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ST) : "memory");
      }
    }
    __builtin_amdgcn_s_barrier();           // every wave's pieces of tile k are in LDS
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* sb = smem + (k & 1) * TILE;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int co = kk ? c1_off : c0_off;
        half8 xf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) xf[j] = *(const half8*)(sb + s * SLICE + x_off[j] + co);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[s][kk][i], xf[j], acc[i][j], 0, 0, 0);
      }
    }
    half8 e0[NJ], e1[NJ];
    if (OPND) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        e0[j] = *(const half8*)(sb + e_off[j] + ec0);
        e1[j] = *(const half8*)(sb + e_off[j] + ec1);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave has read everything it needs of tile k's buffer
    __builtin_amdgcn_s_barrier();                            // ... and so has every other wave: the buffer is free
    __builtin_amdgcn_sched_barrier(0);
    if (k + 2 < ntl) issue_tile((first + (k + 2) * step) * BM, k & 1);
    __builtin_amdgcn_sched_barrier(0);
    const int m0 = (first + k * step) * BM;
    if constexpr (EPI == 3) {
      // ---- conv_igemm.hip's mode-2 epilogue, operation for operation (same per-lane dot order, the same shuffles, the same order over the
      // channel waves, one atomicAdd per output pixel onto the zeroed map): bias, ReLU, dot with the predicted class' 256 weights
      float dotp[NJ];
      long long dot_idx[NJ];
      const int g = grp, cb = wch * 64 + fq * 16;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        dotp[j] = 0.f; dot_idx[j] = -1;
        const int m = m0 + (wpx * NJ + j) * 16 + fi;
        if (m >= M) continue;
        const Pix q = pix_add(ex, ey, en, orx[j], ory[j], orn[j], Wo, Ho);
        const int oy = 2 * q.y + (g >> 1), ox = 2 * q.x + (g & 1);
        float v[16];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r] + bias[i * 2 + (r >> 1)][r & 1];
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        const int slot = ent_slot[q.n];
        const float* wv = wcls + ent_cls[q.n] * 256 + cb;
        float sdot = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const f32x4 w4 = *(const f32x4*)(wv + i * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) sdot += v[i * 4 + r] * w4[r];
        }
        dotp[j] = sdot;
        dot_idx[j] = (long long)slot * (4 * Ho * Wo) + (long long)oy * (2 * Wo) + ox;
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float v0 = dotp[j];
        v0 += __shfl_xor(v0, 16);
        v0 += __shfl_xor(v0, 32);
        dotp[j] = v0;
        if (wch > 0 && fq == 0) red[((wch - 1) * NJ + j) * 16 + fi] = v0;
      }
      __syncthreads();                               // (the next tile's sums are written two barriers later)
      if (wch == 0 && fq == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float v0 = dotp[j];
#pragma unroll
          for (int c = 1; c < 4; ++c) v0 += red[((c - 1) * NJ + j) * 16 + fi];
          if (dot_idx[j] >= 0) atomicAdd(p.dot_out + dot_idx[j], v0);
        }
      }
      { const Pix nx = pix_add(ex, ey, en, dx, dy, dn, Wo, Ho); ex = nx.x; ey = nx.y; en = nx.n; }
      continue;
    }
    // ---- epilogue (conv_igemm.hip's order: bias, residual, top-down add, ReLU, clamp, fp16)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int m = m0 + (wpx * NJ + j) * 16 + fi;
      if (m >= M || ((dbg & 1) && acc[0][j][0] != 12345.f)) continue;
      const Pix q = pix_add(ex, ey, en, orx[j], ory[j], orn[j], Wo, Ho);
      const long long opix = (long long)(q.n * p.out_Hp + q.y + p.out_pad) * p.out_Wp + q.x + p.out_pad;
      f32x2 v[8];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        v[i * 2] = f32x2{acc[i][j][0], acc[i][j][1]} + bias[i * 2];
        v[i * 2 + 1] = f32x2{acc[i][j][2], acc[i][j][3]} + bias[i * 2 + 1];
      }
      if (OPND) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] += f32x2{(float)e0[j][2 * r], (float)e0[j][2 * r + 1]};
          v[4 + r] += f32x2{(float)e1[j][2 * r], (float)e1[j][2 * r + 1]};
        }
      }
      half_t* op = (half_t*)p.out + opix * p.out_Cs + crow;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        half8 h;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float f = v[i * 4 + (r >> 1)][r & 1];
          if (p.relu) f = f > 0.f ? f : 0.f;
          // conv_igemm's clamp (f > 65504 ? 65504 : f < -65504 ? -65504 : f) as one median: the same value for every f but NaN
          f = __builtin_amdgcn_fmed3f(f, lo, 65504.f);
          h[r] = (half_t)f;
        }
        *(half8*)(op + i * 8) = h;
      }
    }
    { const Pix nx = pix_add(ex, ey, en, dx, dy, dn, Wo, Ho); ex = nx.x; ey = nx.y; en = nx.n; }
  }
}

}  // namespace

// Shapes this kernel takes (conv_choose_variant returns 22 for them): 1x1, stride 1, Cin = 256 = the row pitch, Cout a multiple of 256,
// fp16 output, at most one of residual / top-down addend, nothing else of ConvParams' options.
bool conv_wreg_ok(const ConvParams& p) {
  const bool common = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.Cin == 256 && p.in_Cs == 256 && p.Kpad >= 256 && !p.out_f32 && !p.in2 && !p.koff &&
                      !p.down && !p.res32 && !p.mask && p.out_stride <= 1 && !p.head_w && p.nseg == 0;
  if (p.mode == 2)      // the mask head's fused deconv + predictor: four (dy, dx) groups of 256 rows
    return common && p.Cout == 256 && !p.res && !p.up && p.dot_w && p.dot_cls && p.dot_slot && p.dot_out && p.dot_k > 0 && p.Ho * p.Wo > 0 &&
           (long long)p.dot_k * 1024 + (long long)(p.M / (p.Ho * p.Wo)) * 8 <= 48 * 1024;
  return common && p.mode == 0 && p.Cout % 256 == 0 && !p.m_count && !(p.res && p.up) &&
         (!p.up || (p.up_Cs % 8 == 0 && p.up_Cs >= p.Cout)) && p.out_Cs % 8 == 0 && p.out_Cs >= p.Cout;
}

template <int EPI, int NWPX, int BM>
static int launch_wreg(const ConvParams& p, hipStream_t stream) {
  constexpr int lds_max = ((EPI == 1 || EPI == 2) ? 4 : 2) * 4 * BM * 128 + (EPI == 3 ? 1024 + 48 * 1024 : 0);
  static bool done = false;
  if (!done) {
    RS_HIP(hipFuncSetAttribute((const void*)conv1x1_wreg_kernel<EPI, NWPX, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    done = true;
  }
  int lds = ((EPI == 1 || EPI == 2) ? 4 : 2) * 4 * BM * 128;
  if (EPI == 3) lds += 1024 + p.dot_k * 1024 + (p.M / (p.Ho * p.Wo)) * 8;      // partial sums, class weights, slot + class per entry
  const int groups = EPI == 3 ? 4 : p.Cout >> 8, tiles_m = cdiv(p.M, BM);
  int per_group = rs_device_cu_count() * (64 / BM) / groups;     // one workgroup per CU (two of the 32-pixel form)
  if (per_group < 1) per_group = 1;
  if (per_group > tiles_m) per_group = tiles_m;
  hipLaunchKernelGGL((conv1x1_wreg_kernel<EPI, NWPX, BM>), dim3((unsigned)(per_group * groups)), dim3(256 * NWPX), lds, stream, p, rs_debug().wreg_dbg);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

// waves: 4 / 8 per workgroup of the 64-pixel form, 2 = the 32-pixel form with two workgroups per CU, 0 = RS_WREG_WAVES
int launch_conv_wreg(const ConvParams& p, hipStream_t stream, int waves) {
  RS_CHECK(conv_wreg_ok(p) && p.M > 0, RS_ERR_ARG, "conv_wreg: shape outside the kernel's rules");
  const int w = waves ? waves : rs_debug().wreg_waves;
  if (p.mode == 2) return w == 4 ? launch_wreg<3, 1, 64>(p, stream) : launch_wreg<3, 1, 32>(p, stream);      // the mask head's deconv + predictor
  if (w == 2) {
    if (p.res) return launch_wreg<1, 1, 32>(p, stream);
    if (p.up) return launch_wreg<2, 1, 32>(p, stream);
    return launch_wreg<0, 1, 32>(p, stream);
  }
  const bool w8 = w == 8;
  if (p.res) return w8 ? launch_wreg<1, 2, 64>(p, stream) : launch_wreg<1, 1, 64>(p, stream);
  if (p.up) return w8 ? launch_wreg<2, 2, 64>(p, stream) : launch_wreg<2, 1, 64>(p, stream);
  return w8 ? launch_wreg<0, 2, 64>(p, stream) : launch_wreg<0, 1, 64>(p, stream);
}
