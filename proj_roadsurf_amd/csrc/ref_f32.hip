// Reference-precision (fp32) path of the conv / linear stages (rs_spec.precision == 1) on the matrix cores.
//
// The reference computes everything in fp32 (no AMP key in R:config/detectron2_config_3bands.yaml; SURVEY.md §8
// preamble).  The production path uses fp16 operands; this file is the SAME layer semantics (NHWC + zero halo, bias /
// residual / FPN top-down add / ReLU epilogue, 2x2 transposed conv as 4 GEMMs + pixel shuffle, device-side row count)
// with fp32 activations and weights on `v_mfma_f32_16x16x4_f32`: exact fp32 (each product rounded once, fp32
// accumulate -- bitwise a k-ordered fmaf chain, /opt/skills/guides/cdna_hip_programming.md "FP32-input MFMA"), at the
// fp32 matrix rate (157 TFLOP/s peak = 1/16 of the fp16 rate).  It is the like-for-like mode against the reference's
// arithmetic and the strict-parity mode of the tests (tests/test_gpu_engine.py::test_fp32_mode_*,
// ::test_config1_batch16_of_512_tiles); it is not a CPU fallback.
//
// Formulation = conv_igemm.hip's: D[channel][pixel] = sum_k W[channel][k] * X[pixel][k]; weights are the MFMA A
// operand, activations the B operand, so a lane ends up with 4*MI consecutive output channels of one pixel and stores
// 16-byte pieces straight from registers.  Both operands are staged global -> LDS by LDS-DMA (global_load_lds_dwordx4)
// into 128-byte rows = one 32-float K step, XOR-swizzled on the source chunk index and on the ds_read_b128 (the byte-level
// access pattern of the fp16 kernel, so the same conflict-free keys apply), double buffered.
// K order inside a 32-float step: lane (row i, quarter q) reads floats 4q..4q+3 and 16+4q..16+4q+3 of its row with two
// ds_read_b128; MFMA number t of a half uses register t of every lane, i.e. k = {t, 4+t, 8+t, 12+t}: a fixed permutation
// of the summation order, the same for every output element (results do not depend on tile shape or batch size).
// The old VALU kernel (64x64 LDS-tiled SGEMM, ~10 TFLOP/s) is kept as `conv_f32_valu_kernel` for operands that do not
// meet the MFMA kernel's shape rules (none on this network) and as the cross-check of tests/test_gpu_conv.py.
#include "common.h"

namespace {

__device__ __forceinline__ void glds16f(const float* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int WPX, int WCH, int MI, int NJ>
struct TileF {
  static constexpr int NW = WPX * WCH;
  static constexpr int NT = NW * 64;
  static constexpr int BM = WPX * NJ * 16;   // pixels
  static constexpr int BN = WCH * MI * 16;   // channels
  static constexpr int STAGE = (BM + BN) * 128;
  static constexpr int LDS = 2 * STAGE;
};

// TRAIN: instantiation with the backward-epilogue options of the input-gradient convolutions (ConvParams::down / res32 / mask /
// out_stride -- every tensor fp32 here), compiled out of the inference instantiation.
template <int WPX, int WCH, int MI, int NJ, bool TRAIN = false>
__global__ __launch_bounds__(WPX* WCH * 64) void conv_f32_mfma_kernel(const ConvParams p) {
  using T = TileF<WPX, WCH, MI, NJ>;
  constexpr int NW = T::NW, BM = T::BM, BN = T::BN;
  constexpr int PA = BM / (NW * 8);
  constexpr int PW = (BN + NW * 8 - 1) / (NW * 8);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpx = wave / WCH, wch = wave % WCH;

  int M = p.M;
  if (p.m_count) {
    long long mc = (long long)(*p.m_count) * p.m_mul;
    if (mc < M) M = (int)mc;
  }
  const int rows = p.Cout * (p.mode != 0 ? 4 : 1);
  const int tiles_n = (rows + BN - 1) / BN;
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
  const float* in = (const float*)p.in;
  const float* w = (const float*)p.w;
  const int lrow = lane >> 3, lchk = lane & 7;
  const float* aptr[PA];
  const float* wptr[PW];
#pragma unroll
  for (int ps = 0; ps < PA; ++ps) {
    int m = m0 + ps * NW * 8 + wave * 8 + lrow;
    if (m >= M) m = M - 1;
    const int x2 = m % p.Wo, t = m / p.Wo, y = t % p.Ho, n = t / p.Ho;
    const long long base = ((long long)(n * p.in_Hp + y * p.stride + p.in_off) * p.in_Wp + x2 * p.stride + p.in_off) * p.in_Cs;
    aptr[ps] = in + base + (lchk ^ lrow) * 4;
  }
#pragma unroll
  for (int ps = 0; ps < PW; ++ps) {
    const int row = ps * NW * 8 + wave * 8 + lrow;
    const int key = (row & 3) | (((row / (4 * MI)) & 1) << 2);
    int rr = row < BN ? row : BN - 1;
    if (n0 + rr >= rows) rr = rows - 1 - n0;
    wptr[ps] = w + (long long)(n0 + rr) * p.Kpad + (lchk ^ key) * 4;
  }
  // K steps of 32 floats.  Cin >= 32: (32-channel slice outer, taps inner) like the fp16 kernel.  The stem (Cin = 4, tap
  // rows padded to KW = 8 -> 32 contiguous floats per kernel row): one step per kernel row.
  const bool rowmode = p.Cin < 32;
  const int nk = rowmode ? (p.Kpad >> 5) : p.KH * p.KW * (p.Cin >> 5);
  int kh = 0, kw = 0, c0 = 0, w_koff = 0, tstep = 0;
  auto next_off = [&]() {
    if (rowmode) {
      const int off = tstep * p.in_Wp * p.in_Cs;
      w_koff = tstep * 32;
      ++tstep;
      return off;
    }
    const int off = (kh * p.in_Wp + kw) * p.in_Cs + c0;
    w_koff = (kh * p.KW + kw) * p.Cin + c0;
    if (++kw == p.KW) { kw = 0; if (++kh == p.KH) { kh = 0; c0 += 32; } }
    return off;
  };
  auto stage = [&](int buf, int a_off) {
    char* abase = smem + buf * T::STAGE;
    char* wbase = abase + BM * 128;
#pragma unroll
    for (int ps = 0; ps < PA; ++ps) glds16f(aptr[ps] + a_off, abase + (ps * NW * 8 + wave * 8) * 128);
#pragma unroll
    for (int ps = 0; ps < PW; ++ps)
      if (ps * NW * 8 + wave * 8 < BN) glds16f(wptr[ps] + w_koff, wbase + (ps * NW * 8 + wave * 8) * 128);
  };

  const int fi = lane & 15, fq = lane >> 4, fkey = lane & 7;
  int w_off[MI], x_off[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) w_off[i] = BM * 128 + (wch * MI * 16 + (fi >> 2) * 4 * MI + i * 4 + (fi & 3)) * 128;
#pragma unroll
  for (int j = 0; j < NJ; ++j) x_off[j] = (wpx * NJ * 16 + j * 16 + fi) * 128;
  const int c_off[2] = {(fq ^ fkey) * 16, ((4 + fq) ^ fkey) * 16};

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage(0, next_off());
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nk) stage((t + 1) & 1, next_off());
    const char* sb = smem + (t & 1) * T::STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 wf[MI], xf[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) wf[i] = *(const f32x4*)(sb + w_off[i] + c_off[s]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) xf[j] = *(const f32x4*)(sb + x_off[j] + c_off[s]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][r], xf[j][r], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds channels crow .. crow+4*MI-1 of pixel (j, fi)
  const int crow = n0 + wch * MI * 16 + fq * 4 * MI;
  if (crow >= rows) return;
  int g = 0, cb = crow;
  if (p.mode != 0) { g = crow / p.Cout; cb = crow % p.Cout; }
  float bias[4 * MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const f32x4 b4 = *(const f32x4*)(p.bias + crow + i * 4);
    bias[i * 4 + 0] = b4[0]; bias[i * 4 + 1] = b4[1]; bias[i * 4 + 2] = b4[2]; bias[i * 4 + 3] = b4[3];
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int m = m0 + wpx * NJ * 16 + j * 16 + fi;
    if (m >= M) continue;
    const int x = m % p.Wo, t = m / p.Wo, y = t % p.Ho, n = t / p.Ho;
    int oy = y, ox = x;
    if (p.mode != 0) { oy = 2 * y + (g >> 1); ox = 2 * x + (g & 1); }
    else if (TRAIN && p.out_stride > 1) { oy = y * p.out_stride; ox = x * p.out_stride; }
    const long long opix = (long long)(n * p.out_Hp + oy + p.out_pad) * p.out_Wp + ox + p.out_pad;
    float* op = (float*)p.out + opix * p.out_Cs + cb;
    const float* rp = p.res ? (const float*)p.res + opix * p.out_Cs + cb : nullptr;
    const float* up = nullptr;
    if (p.up) {
      const long long upix = (long long)(n * p.up_Hp + (y >> 1) + p.up_pad) * p.up_Wp + (x >> 1) + p.up_pad;
      up = (const float*)p.up + upix * p.up_Cs + cb;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bias[i * 4 + r];
      if (rp) { const f32x4 h = *(const f32x4*)(rp + i * 4); v += h; }
      if (up) { const f32x4 h = *(const f32x4*)(up + i * 4); v += h; }
      if (TRAIN && p.down) {      // backward of the nearest 2x upsample: add the 2x2 block of the finer gradient map
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
          const long long dpix = (long long)(n * p.down_Hp + 2 * y + (dd >> 1) + p.down_pad) * p.down_Wp + 2 * x + (dd & 1) + p.down_pad;
          v += *(const f32x4*)((const float*)p.down + dpix * p.down_Cs + cb + i * 4);
        }
      }
      if (TRAIN && p.res32) v += *(const f32x4*)(p.res32 + opix * p.out_Cs + cb + i * 4);
      if (TRAIN && p.mask) {      // ReLU backward: zero where the saved forward activation is not positive
        const f32x4 h = *(const f32x4*)((const float*)p.mask + opix * p.out_Cs + cb + i * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = h[r] > 0.f ? v[r] : 0.f;
      }
      if (p.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      }
      *(f32x4*)(op + i * 4) = v;
    }
  }
}

template <int WPX, int WCH, int MI, int NJ, bool TRAIN = false>
int launch_f32_variant(const ConvParams& p, hipStream_t stream) {
  using T = TileF<WPX, WCH, MI, NJ>;
  const int rows = p.Cout * (p.mode != 0 ? 4 : 1);
  const long long nblk = (long long)cdiv(rows, T::BN) * cdiv(p.M, T::BM);
  RS_CHECK(nblk > 0 && nblk < (1ll << 31), RS_ERR_ARG, "conv_f32: bad grid %lld", nblk);
  const void* k = (const void*)conv_f32_mfma_kernel<WPX, WCH, MI, NJ, TRAIN>;
  static bool attr = false;
  if (!attr) {
    RS_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS));
    attr = true;
  }
  ConvParams pc = p;
  void* args[] = {&pc};
  RS_HIP(hipLaunchKernel(k, dim3((unsigned)nblk), dim3(T::NT), args, T::LDS, stream));
  return RS_OK;
}

// ------------------------------------------------------------------------------------------------ VALU cross-check kernel
constexpr int TM = 64, TN = 64, TK = 16;

__global__ __launch_bounds__(256) void conv_f32_valu_kernel(const ConvParams p) {
  __shared__ float As[TK][TM + 4];   // [k][pixel]
  __shared__ float Ws[TK][TN + 4];   // [k][channel]
  const int tid = threadIdx.x;
  int M = p.M;
  if (p.m_count) {
    long long mc = (long long)(*p.m_count) * p.m_mul;
    if (mc < M) M = (int)mc;
  }
  const int rows = p.Cout * (p.mode != 0 ? 4 : 1);
  const int tiles_n = (rows + TN - 1) / TN;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int m0 = tile_m * TM, n0 = tile_n * TN;
  if (m0 >= M) return;
  const float* in = (const float*)p.in;
  const float* w = (const float*)p.w;
  const int K = p.KH * p.KW * p.Cin;

  // loader mapping: 256 threads load 64 rows x 16 k (4 per thread along k)
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  long long abase;
  {
    int m = m0 + lr;
    if (m >= M) m = M - 1;
    const int x = m % p.Wo, t = m / p.Wo, y = t % p.Ho, n = t / p.Ho;
    abase = ((long long)(n * p.in_Hp + y * p.stride + p.in_off) * p.in_Wp + x * p.stride + p.in_off) * p.in_Cs;
  }
  const int wrow = n0 + lr < rows ? n0 + lr : rows - 1;
  const int ty = tid >> 4, tx = tid & 15;     // 16x16 threads, 4x4 outputs each
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < K; k0 += TK) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + lk + u;
      float av = 0.f, wv = 0.f;
      if (k < K) {
        const int tap = k / p.Cin, c = k - tap * p.Cin;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        av = in[abase + (long long)(kh * p.in_Wp + kw) * p.in_Cs + c];
        wv = w[(long long)wrow * p.Kpad + k];
      }
      As[lk + u][lr] = av;
      Ws[lk + u][lr] = wv;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TK; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Ws[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
    const int x = m % p.Wo, t = m / p.Wo, y = t % p.Ho, n = t / p.Ho;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int crow = n0 + tx * 4 + j;
      if (crow >= rows) continue;
      int g = 0, cb = crow;
      if (p.mode != 0) { g = crow / p.Cout; cb = crow % p.Cout; }
      int oy = y, ox = x;
      if (p.mode != 0) { oy = 2 * y + (g >> 1); ox = 2 * x + (g & 1); }
      const long long opix = (long long)(n * p.out_Hp + oy + p.out_pad) * p.out_Wp + ox + p.out_pad;
      float v = acc[i][j] + p.bias[crow];
      if (p.res) v += ((const float*)p.res)[opix * p.out_Cs + cb];
      if (p.up) {
        const long long upix = (long long)(n * p.up_Hp + (y >> 1) + p.up_pad) * p.up_Wp + (x >> 1) + p.up_pad;
        v += ((const float*)p.up)[upix * p.up_Cs + cb];
      }
      if (p.relu) v = v > 0.f ? v : 0.f;
      ((float*)p.out)[opix * p.out_Cs + cb] = v;
    }
  }
}

}  // namespace

// ConvParams with every tensor pointer (in, w, out, res, up) referring to fp32 data.  force_valu: the VALU cross-check kernel.
int launch_conv_f32(const ConvParams& p, hipStream_t stream, int force_valu, int tile) {
  RS_CHECK(p.M > 0 && p.mode != 2, RS_ERR_ARG, "conv_f32: bad arguments");
  const int rows = p.Cout * (p.mode != 0 ? 4 : 1);
  // shape rules of the MFMA kernel: 128-byte K steps (Cin % 32 == 0, or the stem's 32-float kernel rows), 16-byte aligned
  // channel groups on both sides
  const bool rowmode = p.Cin < 32 && p.Cin * p.KW == 32 && p.in_Cs == p.Cin && p.Kpad % 32 == 0;
  const bool mfma_ok = !force_valu && (p.Cin % 32 == 0 || rowmode) && p.Kpad % 32 == 0 && p.in_Cs % 4 == 0 && p.out_Cs % 4 == 0 &&
                       rows % 16 == 0 && (p.mode == 0 || p.Cout % 16 == 0) && (!p.up || p.up_Cs % 4 == 0);
  const bool train = p.down || p.res32 || p.mask || p.out_stride > 1;
  RS_CHECK(!train || (mfma_ok && p.mode == 0), RS_ERR_UNSUPPORTED, "conv_f32: the backward epilogue needs the MFMA kernel's shape rules and mode 0");
  if (mfma_ok) {
    // Few 128 x 128 tiles (res4 / res5 and the p4 / p5 levels: 300-1 260 of them on 256 CUs x 2) fill the chip for one full round and a
    // ragged second one: the 128 x 64 tile (3 workgroups per CU) runs res4.x.conv2 in 403 instead of 578 us, res5.x.conv2 in 482 instead of
    // 575 (tools/ubench/f32_tiles.py); from ~1 300 tiles on the wide tile wins (fpn_output3: 1 466 vs 1 590 us).  The K order inside an
    // output element does not depend on the tile, so the choice changes no bit.
    const bool wide = rows % 128 == 0 && (long long)cdiv(p.M, 128) * (rows / 128) >= 5ll * rs_device_cu_count();
    if (train) {
      if (wide) return launch_f32_variant<2, 2, 4, 4, true>(p, stream);
      if (rows % 64 == 0) return launch_f32_variant<4, 1, 4, 2, true>(p, stream);
      return launch_f32_variant<4, 1, 1, 4, true>(p, stream);
    }
    // tile forced by the operator interface (rs_op_conv2d variant 30 / 31, tools/ubench/f32_tiles.py)
    if (tile == 30 && rows % 128 == 0) return launch_f32_variant<2, 2, 4, 4>(p, stream);
    if (tile == 31 && rows % 64 == 0) return launch_f32_variant<4, 1, 4, 2>(p, stream);
    RS_CHECK(tile != 30 && tile != 31, RS_ERR_ARG, "conv_f32: tile %d does not divide %d rows", tile, rows);
    if (wide) return launch_f32_variant<2, 2, 4, 4>(p, stream);               // 128 px x 128 ch
    if (rows % 64 == 0) return launch_f32_variant<4, 1, 4, 2>(p, stream);     // 128 px x 64 ch
    return launch_f32_variant<4, 1, 1, 4>(p, stream);                          // 256 px x 16 ch (heads)
  }
  const long long nblk = (long long)cdiv(rows, TN) * cdiv(p.M, TM);
  RS_CHECK(nblk > 0 && nblk < (1ll << 31), RS_ERR_ARG, "conv_f32: bad grid");
  hipLaunchKernelGGL(conv_f32_valu_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
