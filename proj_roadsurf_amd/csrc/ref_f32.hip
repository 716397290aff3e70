// fp32 VALIDATION path of the conv/GEMM stages (rs_spec.precision == 1).
//
// The production path computes every conv/linear layer with fp16 operands on MFMA.  Detection
// pipelines amplify tiny feature differences through discrete decisions (top-k, greedy NMS,
// thresholds), so end-to-end agreement with the fp32 oracle is statistical in fp16.  This file is
// the same layer semantics (NHWC + zero halo, bias / residual / FPN upsample-add / ReLU epilogue,
// 2x2 transposed conv as 4 GEMMs, device-side row count) in plain fp32 FMAs, used only to show that
// the engine's LOGIC reproduces the oracle end to end (tests/test_gpu_engine.py::test_fp32_mode_*).
// It is not a CPU fallback and not tuned: 64x64 LDS-tiled SGEMM, ~10 TFLOP/s.
#include "common.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvParams p) {
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

// ConvParams with every tensor pointer (in, w, out, res, up) referring to fp32 data.
int launch_conv_f32(const ConvParams& p, hipStream_t stream) {
  RS_CHECK(p.M > 0 && p.mode != 2, RS_ERR_ARG, "conv_f32: bad arguments");
  const int rows = p.Cout * (p.mode != 0 ? 4 : 1);
  const long long nblk = (long long)cdiv(rows, TN) * cdiv(p.M, TM);
  RS_CHECK(nblk > 0 && nblk < (1ll << 31), RS_ERR_ARG, "conv_f32: bad grid");
  hipLaunchKernelGGL(conv_f32_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
