// Weight gradient of a convolution / linear layer for gfx950 (MI355X): fp16 operands, fp32 accumulate on MFMA.
//
// Training path: what autograd reaches through cuDNN's backward-filter / cuBLAS for
// every trainable Conv2d / Linear of the detectron2 model ([EXT d2: layers/wrappers.py Conv2d; modeling/backbone/
// {resnet,fpn}.py; modeling/proposal_generator/rpn.py; modeling/roi_heads/{box_head,fast_rcnn,mask_head}.py];
// trainable set fixed by FREEZE_AT 2, R:config/detectron2_config_3bands.yaml:58).
//
//   dW[co][(kh,kw,ci)] = sum over output pixels m = (n,y,x) of  dY[m][co] * X[n][y*s+kh-pad][x*s+kw-pad][ci]
//
// As a GEMM the reduction runs over PIXELS, and both operands are stored pixel-major (NHWC): every MFMA operand
// is "k-strided" in memory.  The tiles are therefore staged row-major ([pixel][64 channels], 128-byte rows, LDS-DMA
// exactly like the forward kernel's activation tile, zero halo = no bounds checks) and read back TRANSPOSED with
// ds_read_b64_tr_b16: a 16-lane group reads a 4-pixel x 16-channel block and each lane receives the 4 pixels of
// its channel, two reads = the 8 consecutive k of a v_mfma_f32_16x16x32_f16 operand.
//   * swizzle: LDS slot s of row r holds source chunk s ^ key(r), key(r) = ((r>>1)&1)<<1 | ((r>>3)&1)<<2 -- the 8 rows
//     a 32-lane half touches in one transposed read ({0..3, 8..11} or {4..7, 12..15} of a 16-row group) land on 8
//     disjoint 8-bank spans (checked by simulation, tools/ubench/lds_tr_banks.py).
//   * workgroup tile: CB x 64 output channels x 128 K columns (= two (tap, 64-channel slice) pairs), wave tile 64x64, K step
//     = 64 pixels.  CB = 2 (Cout <= 128): 4 waves, double buffered, two workgroups per CU.  CB = 4: 8 waves, 256 x 128,
//     THREE stages of 48 KB with the loads issued two steps ahead (counted vmcnt): a 64-pixel step is only ~0.25 us of MFMA
//     work per wave, less than an L2 round trip, so the double-buffered form waits for memory every step (630-690 TFLOP/s on
//     the 3x3 256->256 shapes), and the wider tile moves 25 % fewer L2 bytes per FLOP.
//   * the pixel range is split over gridDim.z; every split writes its own fp32 partial tile and wgrad_reduce_kernel
//     adds the partials in a fixed order (bitwise reproducible; no float atomics), applies the per-output-channel
//     FrozenBN scale (the trainable tensor is the UNFOLDED weight) and accumulates into the gradient.
//   * rows of dY beyond M read a zero row (WgradParams::zeros), so the tail contributes nothing.
#include "common.h"

namespace {

typedef _Float16 half4v __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
constexpr int SUB = BK * 128;                 // one [64 px][64 ch] sub-tile: 8 KB
template <int CB> struct WgCfg {
  static constexpr int BM = CB * 64, NT = CB * 128, WAVES = CB * 2;
  static constexpr int SUBS = CB + 2;                       // dY blocks 0..CB-1, X half 0, X half 1
  static constexpr int STAGE = SUBS * SUB;
  static constexpr int NST = CB == 4 ? 3 : 2;
  static constexpr int LDS_BYTES = NST * STAGE;             // 64 KB (CB 2) / 144 KB (CB 4)
  static constexpr int PCS = 8 / WAVES;                     // 8-row pieces of a sub-tile staged per wave
  static constexpr int LOADS = PCS * SUBS;                  // LDS-DMA pieces a lane issues per stage
};

__device__ __forceinline__ void glds16(const half_t* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int row_key(int r) { return (((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2); }

template <int CB>
__global__ __launch_bounds__(WgCfg<CB>::NT) void conv_wgrad_kernel(const WgradParams p) {
  using C = WgCfg<CB>;
  constexpr int BM = C::BM, STAGE = C::STAGE, NST = C::NST;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;      // wave tile: channels wr*64.., K columns wc*64..

  const int co0 = blockIdx.x * BM;
  int M = p.M;
  if (p.m_count) {
    const long long mc = (long long)(*p.m_count) * p.m_mul;
    if (mc < M) M = (int)mc;
  }
  // K columns of this workgroup: two 64-wide halves, each one (tap, channel slice)
  const int slices = p.Cin >> 6;                // 64-channel slices per tap
  int x_off[2], k_col[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int u = blockIdx.y * 2 + h;                 // (tap, slice) unit index, tap-major
    const int units = p.KH * p.KW * slices;
    if (u >= units) u = units - 1;              // odd unit count: the second half repeats the last unit (not stored)
    const int tap = u / slices, sl = u - tap * slices;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    x_off[h] = (kh * p.in_Wp + kw) * p.in_Cs + sl * 64;
    k_col[h] = tap * p.Cin + sl * 64;
  }
  const bool second_valid = (blockIdx.y * 2 + 1) < p.KH * p.KW * slices;

  // pixel range of this split (multiples of BK)
  const int steps_total = (M + BK - 1) / BK;
  const int per = (steps_total + gridDim.z - 1) / gridDim.z;
  const int s0 = blockIdx.z * per;
  int s1 = s0 + per;
  if (s1 > steps_total) s1 = steps_total;

  // ---- staging: a sub-tile is 8 pieces of 8 rows; wave w stages pieces w*PCS .. of each of the CB + 2 sub-tiles
  const int lrow = lane >> 3, lchk = lane & 7;
  auto stage = [&](int buf, int step) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int pc = 0; pc < C::PCS; ++pc) {
      const int piece = wave * C::PCS + pc;
      const int r = piece * 8 + lrow;                          // row inside the 64-pixel K step
      const int m = step * BK + r;
      const int src_chunk = (lchk ^ row_key(r)) * 8;
      const half_t *gy, *gx;
      if (m < M) {
        const int x = m % p.Wo;
        const int t = m / p.Wo;
        const int y = t % p.Ho;
        const int n = t / p.Ho;
        gy = p.dy + ((long long)(n * p.dy_Hp + y + p.dy_pad) * p.dy_Wp + x + p.dy_pad) * p.dy_Cs + co0 + src_chunk;
        gx = p.x + ((long long)(n * p.in_Hp + y * p.stride + p.in_off) * p.in_Wp + x * p.stride + p.in_off) * p.in_Cs + src_chunk;
      } else {
        gy = p.zeros + co0 + src_chunk;                        // zero row: the tail contributes nothing
        gx = p.x + src_chunk;                                  // any valid address; multiplied by zero
      }
      char* dst = base + piece * 1024;
      // channel chunks past the gradient buffer's row (narrow heads: 16 outputs stored 64 wide) read the zero row
#pragma unroll
      for (int h = 0; h < CB; ++h) glds16((co0 + h * 64 + src_chunk < p.dy_Cs) ? gy + h * 64 : p.zeros + src_chunk, dst + h * SUB);
      glds16(gx + (m < M ? x_off[0] : 0), dst + CB * SUB);
      glds16(gx + (m < M ? x_off[1] : 0), dst + (CB + 1) * SUB);
    }
  };

  // ---- transposed fragment reads
  // lane l: group g = l>>4 supplies k rows 8g..8g+7 of a 32-pixel MFMA step; inside the group q = (l&15)>>2 is the
  // row of the 4-row block, pp = l&3 the 4-element column piece.  Block columns = 16 channels = chunks 2cb, 2cb+1.
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // byte offset (inside a sub-tile) of this lane's address for channel block cb, MFMA step ks (0/1), half hh (0/1)
  auto tr_addr = [&](int cb, int ks, int hh) {
    const int r = ks * 32 + g * 8 + hh * 4 + q;
    const int c = 2 * cb + (pp >> 1);
    return (unsigned)(r * 128 + ((c ^ row_key(r)) << 4) + 8 * (pp & 1));
  };
  unsigned a_addr[4][2][2], b_addr[4][2][2];     // [block][ks][hh]; wave's dY sub-tile = wr, X sub-tile = CB + wc
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        a_addr[i][ks][hh] = lds0 + wr * SUB + tr_addr(i, ks, hh);
        b_addr[i][ks][hh] = lds0 + (CB + wc) * SUB + tr_addr(i, ks, hh);
      }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define RS_TR(dst, addr) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr))
  if (s0 < s1) {
    stage(0, s0);
    if (NST == 3 && s0 + 1 < s1) stage(1, s0 + 1);
    int buf = 0;
    for (int s = s0; s < s1; ++s) {
      // step s has landed: everything but the (NST - 2) younger stages' pieces
      if (NST == 3 && s + 1 < s1) {
        static_assert(CB != 4 || C::LOADS == 6, "vmcnt immediate below");
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();                                         // ... for every wave; and everyone is done reading step s-1
      if (s + NST - 1 < s1) stage(buf == 0 ? NST - 1 : buf - 1, s + NST - 1);   // into the buffer step s-1 used
      const unsigned bo = (unsigned)(buf * STAGE);
      // the 16 transposed reads of the first 32-pixel half and 12 of the second are issued before the first MFMA (lgkmcnt is a
      // 4-bit counter: at most 15 may be left in flight); the first 16 MFMAs wait only for the older 16, so the second half's
      // LDS latency hides behind them
      half4v a_lo[2][4], a_hi[2][4], b_lo[2][4], b_hi[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        RS_TR(a_lo[0][i], a_addr[i][0][0] + bo);
        RS_TR(a_hi[0][i], a_addr[i][0][1] + bo);
        RS_TR(b_lo[0][i], b_addr[i][0][0] + bo);
        RS_TR(b_hi[0][i], b_addr[i][0][1] + bo);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        RS_TR(a_lo[1][i], a_addr[i][1][0] + bo);
        RS_TR(a_hi[1][i], a_addr[i][1][1] + bo);
        RS_TR(b_lo[1][i], b_addr[i][1][0] + bo);
        RS_TR(b_hi[1][i], b_addr[i][1][1] + bo);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) {
          asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
          RS_TR(a_lo[1][3], a_addr[3][1][0] + bo);                 // the last four of the second half go out behind the wait
          RS_TR(a_hi[1][3], a_addr[3][1][1] + bo);
          RS_TR(b_lo[1][3], b_addr[3][1][0] + bo);
          RS_TR(b_hi[1][3], b_addr[3][1][1] + bo);
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        half8 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          af[i] = __builtin_shufflevector(a_lo[ks][i], a_hi[ks][i], 0, 1, 2, 3, 4, 5, 6, 7);
          bf[i] = __builtin_shufflevector(b_lo[ks][i], b_hi[ks][i], 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      buf = buf + 1 == NST ? 0 : buf + 1;
    }
  }
#undef RS_TR

  // ---- store the partial tile: D[row = 4*(lane>>4) + e][col = lane&15] of block (i, j)
  if (wc == 1 && !second_valid) return;
  // one split: the tile IS the gradient (same arithmetic as wgrad_reduce_kernel with splits == 1, without its launch)
  const bool direct = gridDim.z == 1 && p.KH * p.KW * p.Cin == p.Kpad;
  float* out = direct ? p.grad : p.partial + (long long)blockIdx.z * p.Cout * p.Kpad;
  const int kc = k_col[wc];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = co0 + wr * 64 + i * 16 + (lane >> 4) * 4 + e;
      if (co >= p.Cout) continue;
      float* row = out + (long long)co * p.Kpad + kc + (lane & 15);
      if (direct) {
        const float sc = p.scale ? p.scale[co] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = 0.f + acc[i][j][e];
          if (p.scale) v *= sc;
          row[j * 16] = p.accumulate ? row[j * 16] + v : v;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) row[j * 16] = acc[i][j][e];
      }
    }
  }
}

// grad[co][k] (+)= scale[co] * sum_z partial[z][co][k].  A block owns 64 consecutive elements; wave g sums the planes
// z = g, g+4, ... (four independent load chains per element instead of one `splits`-long one: the kernel is pure latency
// at these sizes), then the four sums are added in the fixed order ((g0+g1)+g2)+g3 -- deterministic.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, int splits, long long n_el, int Kpad,
                                                           const float* scale, float* grad, int accumulate) {
  __shared__ float part[4][64];
  const int g = threadIdx.x >> 6, e = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 64 + e;
  float s = 0.f;
  if (i < n_el) {
    int z = g;
    for (; z + 12 < splits; z += 16) {
      const float a = partial[(long long)z * n_el + i], b = partial[(long long)(z + 4) * n_el + i];
      const float c = partial[(long long)(z + 8) * n_el + i], d = partial[(long long)(z + 12) * n_el + i];
      s += a; s += b; s += c; s += d;
    }
    for (; z < splits; z += 4) s += partial[(long long)z * n_el + i];
  }
  part[g][e] = s;
  __syncthreads();
  if (g != 0 || i >= n_el) return;
  s = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
  if (scale) s *= scale[i / Kpad];
  grad[i] = accumulate ? grad[i] + s : s;
}


// ------------------------------------------------------------------------------------------------ reference precision (fp32)
// Weight gradient with fp32 operands on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate) for the reference-precision
// trainer (rs_spec.precision = 1; the reference trains in fp32, R:config/detectron2_config_3bands.yaml:268-305 has no AMP key).
// One wave = a 64-channel x 64-K-column tile of dW, reduction over pixels four at a time: lane (c = lane & 15, q = lane >> 4) loads
// 16 bytes of pixel m + q from each operand -- channels co0 + 4c .. 4c+3 of dY, columns ci0 + 4c .. 4c+3 of X -- and MFMA (t, j) multiplies
// element t of the first with element j of the second, i.e. accumulator (t, j) holds rows co0 + 4r + t x columns ci0 + 4c + j: both
// operands are read straight from global memory in 256-byte runs, no LDS and no transposition.  The four waves of a workgroup take
// four adjacent K-column tiles (same dY rows: L1 hits).  Pixel range split over gridDim.z into the same fp32 partial planes as the
// fp16 kernel, summed by wgrad_reduce_kernel in a fixed order.
__global__ __launch_bounds__(256) void conv_wgrad_f32_kernel(const WgradParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slices = p.Cin >> 6;
  const int units = p.KH * p.KW * slices;
  const int unit = blockIdx.y * 4 + wave;
  if (unit >= units) return;
  int M = p.M;
  if (p.m_count) {
    const long long mc = (long long)(*p.m_count) * p.m_mul;
    if (mc < M) M = (int)mc;
  }
  const int co0 = blockIdx.x * 64;
  const int tap = unit / slices, ci0 = (unit - tap * slices) * 64, kh = tap / p.KW, kw = tap - kh * p.KW;
  const int steps = (M + 3) >> 2;
  const int per = (steps + (int)gridDim.z - 1) / (int)gridDim.z;
  const int s0 = blockIdx.z * per;
  const int s1 = s0 + per < steps ? s0 + per : steps;
  const int lq = lane >> 4, lc = lane & 15;
  const float* dy = (const float*)p.dy;
  const float* x = (const float*)p.x;
  const bool cok = co0 + 4 * lc + 4 <= p.dy_Cs;
  f32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Pixel m = 4 s + lq of this lane, advanced by 4 per step: (xx, row) = (x, n * Ho + y).  The loop body is STRAIGHT-LINE code: loads
  // from clamped (always valid) addresses, the tail / channel masks applied by selects at the point of use -- with a branch around a
  // load hipcc waits vmcnt(0) before the first MFMA of every iteration (seen in the ISA) and the prefetch below overlaps nothing.
  int m = s0 * 4 + lq;
  int xx = m % p.Wo, y, n;
  { const int row = m / p.Wo; y = row % p.Ho; n = row / p.Ho; }
  const int q4 = 4 / p.Wo, r4 = 4 - q4 * p.Wo;          // a step advances the pixel by 4 = q4 rows + r4 columns (launcher: q4 < Ho)
  const int last_row = (M - 1) / p.Wo, last_x = (M - 1) - last_row * p.Wo, last_n = last_row / p.Ho, last_y = last_row - last_n * p.Ho;
  const int coff = cok ? co0 + 4 * lc : 0;
  auto load = [&](f32x4& a, f32x4& b, bool& ok) {
    ok = m < M;
    const int yc = ok ? y : last_y, nc = ok ? n : last_n, xc = ok ? xx : last_x;
    a = *(const f32x4*)(dy + ((long long)(nc * p.dy_Hp + yc + p.dy_pad) * p.dy_Wp + xc + p.dy_pad) * p.dy_Cs + coff);
    b = *(const f32x4*)(x + ((long long)(nc * p.in_Hp + yc * p.stride + kh + p.in_off) * p.in_Wp + xc * p.stride + kw + p.in_off) * p.in_Cs + ci0 + 4 * lc);
    m += 4;
    xx += r4;
    const bool wx = xx >= p.Wo;
    xx = wx ? xx - p.Wo : xx;
    y += q4 + (wx ? 1 : 0);
    const bool wy = y >= p.Ho;
    y = wy ? y - p.Ho : y;
    n += wy ? 1 : 0;
  };
  auto mfma16 = [&](f32x4 a, f32x4 b, bool ok) {
    const bool oka = ok && cok;
#pragma unroll
    for (int t = 0; t < 4; ++t) { a[t] = oka ? a[t] : 0.f; b[t] = ok ? b[t] : 0.f; }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[j], acc[t][j], 0, 0, 0);
  };
  // Two steps per half iteration: the four loads of the NEXT pair are in flight under the 32 MFMAs of this one.  The two pairs
  // ping-pong between two register sets WITHOUT copies (a copy of a just-loaded register is a use: it would wait for the load at once),
  // and a sched_barrier pins the issue order (hipcc otherwise sinks the loads below the MFMAs).
  f32x4 a0, b0, a1, b1, a2, b2, a3, b3;
  bool k0 = false, k1 = false, k2 = false, k3 = false;
  int s = s0;
  if (s + 1 < s1) {
    load(a0, b0, k0);
    load(a1, b1, k1);
    for (; s + 5 < s1; s += 4) {             // steps s, s+1 are loaded; s+2 .. s+5 exist
      load(a2, b2, k2);
      load(a3, b3, k3);
      __builtin_amdgcn_sched_barrier(0);
      mfma16(a0, b0, k0);
      mfma16(a1, b1, k1);
      __builtin_amdgcn_sched_barrier(0);
      load(a0, b0, k0);
      load(a1, b1, k1);
      __builtin_amdgcn_sched_barrier(0);
      mfma16(a2, b2, k2);
      mfma16(a3, b3, k3);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma16(a0, b0, k0);
    mfma16(a1, b1, k1);
    s += 2;
  }
  for (; s < s1; ++s) {                      // at most four steps left
    load(a0, b0, k0);
    mfma16(a0, b0, k0);
  }
  float* out = p.partial + (long long)blockIdx.z * p.Cout * p.Kpad;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = co0 + 4 * (4 * lq + e) + t;
      if (co < p.Cout)
        *(f32x4*)(out + (long long)co * p.Kpad + unit * 64 + 4 * lc) = f32x4{acc[t][0][e], acc[t][1][e], acc[t][2][e], acc[t][3][e]};
    }
}

}  // namespace

// 256-wide tile (CB 4) where it pays: more than 128 output channels, at least 4 K-column tiles, and a pixel range per
// workgroup long enough to amortise its deeper pipeline (measured on the layer shapes of the batch-8 step,
// tools/ubench/wgrad_shapes.py: +14 % on the 3x3 256->256 maps of p2/p3, -8 % on the short 1x1 bottleneck shapes).
static inline int wgrad_cb(const WgradParams& p) {
  const int units = p.KH * p.KW * (p.Cin >> 6);
  if (p.Cout <= 128 || units < 8) return 2;
  const long long out_tiles = (long long)cdiv(p.Cout, 256) * cdiv(units, 2);
  const int steps = cdiv(p.M, BK);
  long long s = 512 / out_tiles;
  if (s > steps / 4) s = steps / 4;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return steps / s >= 8 ? 4 : 2;
}

int wgrad_splits(const WgradParams& p) {
  if (p.f32) {
    // conv_wgrad_f32_kernel: a workgroup is 64 channels x 4 K-column tiles of 64, its waves are independent and hide their own load
    // latency only through the waves beside them -- aim at ~5 waves per SIMD (1280 workgroups), at least 8 four-pixel steps per split
    const long long out_tiles = (long long)cdiv(p.Cout, 64) * cdiv(p.KH * p.KW * (p.Cin >> 6), 4);
    const int steps = cdiv(p.M, 4);
    long long s = cdiv(1280, out_tiles);
    if (s > steps / 8) s = steps / 8;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return (int)s;
  }
  const int cb = wgrad_cb(p);
  const long long out_tiles = (long long)cdiv(p.Cout, cb * 64) * cdiv(p.KH * p.KW * (p.Cin >> 6), 2);
  const int steps = cdiv(p.M, BK);
  int target = rs_debug().wgrad_target;
  if (target < 1 || target > 1024) target = 1024;   // the trainer sizes its scratch for <= 1024
  // ~2 workgroups per CU: more splits only add partial-tile traffic (measured 1024 -> 512: -2 % step time).  The 256-wide
  // tile runs one workgroup per CU: round DOWN so that the grid is at most two full rounds of the 256 CUs.
  long long s = cb == 4 ? (target < 512 ? target : 512) / out_tiles : cdiv(target, out_tiles);
  if (s > steps / 4) s = steps / 4;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

int launch_conv_wgrad(const WgradParams& p, hipStream_t stream) {
  if (p.f32) {
    RS_CHECK(p.dy && p.x && p.partial && p.grad, RS_ERR_ARG, "wgrad: null pointer");
    const int units = p.KH * p.KW * (p.Cin >> 6);
    RS_CHECK(p.M > 0 && p.Cin % 64 == 0 && p.dy_Cs % 4 == 0 && p.in_Cs % 4 == 0 && p.Cout >= 1 && units * 64 == p.Kpad && p.splits >= 1, RS_ERR_ARG,
             "wgrad (fp32): Cin %d must be a multiple of 64 and K = %d unpadded", p.Cin, p.Kpad);
    RS_CHECK(4 / p.Wo + 1 <= p.Ho, RS_ERR_UNSUPPORTED, "wgrad (fp32): map %d x %d too small for the 4-pixel step", p.Ho, p.Wo);
    dim3 grid(cdiv(p.Cout, 64), cdiv(units, 4), p.splits);
    hipLaunchKernelGGL(conv_wgrad_f32_kernel, grid, dim3(256), 0, stream, p);
    RS_HIP(hipGetLastError());
    const long long n_el = (long long)p.Cout * p.Kpad;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv(n_el, 64)), dim3(256), 0, stream, p.partial, p.splits, n_el, p.Kpad,
                       p.scale, p.grad, p.accumulate);
    RS_HIP(hipGetLastError());
    return RS_OK;
  }
  RS_CHECK(p.dy && p.x && p.partial && p.grad && p.zeros, RS_ERR_ARG, "wgrad: null pointer");
  RS_CHECK(p.M > 0 && p.Cin % 64 == 0 && p.dy_Cs % 8 == 0 && p.Cout >= 1 && p.Cout <= p.dy_Cs, RS_ERR_ARG, "wgrad: Cin %d must be a multiple of 64, Cout %d <= gradient row width %d", p.Cin, p.Cout, p.dy_Cs);
  RS_CHECK(p.KH * p.KW * p.Cin <= p.Kpad && p.splits >= 1, RS_ERR_ARG, "wgrad: K exceeds Kpad");
  static bool done = false;
  if (!done) {
    RS_HIP(hipFuncSetAttribute((const void*)conv_wgrad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, WgCfg<2>::LDS_BYTES));
    RS_HIP(hipFuncSetAttribute((const void*)conv_wgrad_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, WgCfg<4>::LDS_BYTES));
    done = true;
  }
  const int units = p.KH * p.KW * (p.Cin >> 6);
  const int force_cb = rs_debug().wgrad_cb;
  const int cb = force_cb == 2 || force_cb == 4 ? force_cb : wgrad_cb(p);
  dim3 grid(cdiv(p.Cout, cb * 64), cdiv(units, 2), p.splits);
  if (cb == 4) hipLaunchKernelGGL(conv_wgrad_kernel<4>, grid, dim3(WgCfg<4>::NT), WgCfg<4>::LDS_BYTES, stream, p);
  else hipLaunchKernelGGL(conv_wgrad_kernel<2>, grid, dim3(WgCfg<2>::NT), WgCfg<2>::LDS_BYTES, stream, p);
  RS_HIP(hipGetLastError());
  if (p.splits == 1 && units * 64 == p.Kpad) return RS_OK;       // stored by the kernel itself (no padding columns to define)
  const long long n_el = (long long)p.Cout * p.Kpad;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv(n_el, 64)), dim3(256), 0, stream, p.partial, p.splits, n_el, p.Kpad,
                     p.scale, p.grad, p.accumulate);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
