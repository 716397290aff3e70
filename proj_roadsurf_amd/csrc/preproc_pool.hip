// Tile ingest and the two pooling ops of the backbone (HBM-bound byte/half kernels).
//
//  * preprocess_kernel: DefaultPredictor.__call__ + GeneralizedRCNN.preprocess_image fused
//    ([EXT d2: engine/defaults.py, data/transforms/transform.py ResizeTransform,
//      modeling/meta_arch/rcnn.py]; R:config/detectron2_config_3bands.yaml:26-30,81-88):
//    BGR->RGB flip, Pillow's 2-pass fixed-point bilinear resize (uint8 rounding between the
//    passes, antialiased when shrinking -- coefficient tables come from the host, bit-exact with
//    Pillow's precompute_coeffs/normalize_coeffs_8bpc), (x-mean)/std, fp16 NHWC with the channel
//    dim padded to 4 (8 bytes per pixel) and a 3-pixel zero halo for the 7x7 stem.
//  * maxpool3x3s2_kernel: stem max_pool2d(3, 2, 1) [EXT d2: modeling/backbone/resnet.py BasicStem].
//    The input is post-ReLU (>= 0), so the zero halo is equivalent to -inf padding.
//  * subsample2_kernel: LastLevelMaxPool = max_pool2d(k=1, s=2) [EXT d2: modeling/backbone/fpn.py].
#include "common.h"

__global__ __launch_bounds__(256) void preprocess_kernel(const PreprocParams p) {
  // block = 64 output columns x 4 rows of one tile (no integer divisions)
  const int X = blockIdx.x * 64 + (threadIdx.x & 63);
  const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.z;
  if (X >= p.new_w || Y >= p.new_h) return;
  int ymin = Y, ny = 1, xmin = X, nx = 1;
  if (p.need_v) { ymin = p.vb[Y * 2]; ny = p.vb[Y * 2 + 1]; }
  if (p.need_h) { xmin = p.hb[X * 2]; nx = p.hb[X * 2 + 1]; }
  const uint8_t* img = p.tiles + (long long)n * p.H * p.W * p.C;
  const int C = p.C;
  // all channels of a source pixel are adjacent bytes: walk the taps once and carry one accumulator per channel
  // (same integer arithmetic per channel as Pillow's ImagingResampleHorizontal_8bpc / Vertical_8bpc).
  int vacc[4] = {1 << 21, 1 << 21, 1 << 21, 1 << 21};
  int val[4] = {0, 0, 0, 0};
  const int* hk = p.hk + (long long)X * p.ksh;
  const int* vk = p.vk + (long long)Y * p.ksv;
  for (int ty = 0; ty < ny; ++ty) {
    const uint8_t* row = img + ((long long)(ymin + ty) * p.W) * C;
    int h[4] = {0, 0, 0, 0};
    if (p.need_h) {
      int ss[4] = {1 << 21, 1 << 21, 1 << 21, 1 << 21};
      for (int tx = 0; tx < nx; ++tx) {
        const uint8_t* px = row + (xmin + tx) * C;
        const int k = hk[tx];
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < C) ss[c] += (int)px[c] * k;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) { const int v = ss[c] >> 22; h[c] = v < 0 ? 0 : (v > 255 ? 255 : v); }
    } else {
      const uint8_t* px = row + X * C;
#pragma unroll
      for (int c = 0; c < 4; ++c) if (c < C) h[c] = px[c];
    }
    if (p.need_v) {
      const int k = vk[ty];
#pragma unroll
      for (int c = 0; c < 4; ++c) vacc[c] += h[c] * k;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) val[c] = h[c];
    }
  }
  half4 o;
  float of[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    o[c] = (half_t)0.f;
    of[c] = 0.f;
    if (c < C) {
      const int cs = p.flip ? (C - 1 - c) : c;      // model channel c reads source channel cs
      int v = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) if (q == cs) v = p.need_v ? (vacc[q] >> 22) : val[q];
      if (p.need_v) v = v < 0 ? 0 : (v > 255 ? 255 : v);
      const float f = ((float)v - p.mean[c]) / p.stdv[c];
      o[c] = (half_t)f;
      of[c] = f;
    }
  }
  const long long oidx = (((long long)n * p.out_Hp + Y + 3) * p.out_Wp + X + 3) * 4;
  if (p.out_f32) {
    *(f32x4*)((float*)p.out + oidx) = f32x4{of[0], of[1], of[2], of[3]};
  } else {
    *(half4*)(p.out + oidx) = o;
  }
}

int launch_preprocess(const PreprocParams& p, hipStream_t s) {
  hipLaunchKernelGGL(preprocess_kernel, dim3(cdiv(p.new_w, 64), cdiv(p.new_h, 4), p.N), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

// in: [N][Hi+2][Wi+2][C] (halo 1), out: [N][Ho+2][Wo+2][C] (halo 1); window 3x3 stride 2 pad 1.
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const half_t* in, half_t* out, int N, int Hi, int Wi, int Ho, int Wo, int C) {
  const int cv = C >> 3;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * Ho * Wo * cv;
  if (gid >= total) return;
  const int c8 = (int)(gid % cv);
  long long t = gid / cv;
  const int x = (int)(t % Wo); t /= Wo;
  const int y = (int)(t % Ho);
  const int n = (int)(t / Ho);
  const int Hip = Hi + 2, Wip = Wi + 2;
  // output (y,x) covers input rows 2y-1..2y+1 -> halo-buffer rows 2y..2y+2
  half8 m;
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = (half_t)0.f;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const half8 v = *(const half8*)(in + (((long long)n * Hip + 2 * y + dy) * Wip + 2 * x + dx) * C + c8 * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) m[i] = v[i] > m[i] ? v[i] : m[i];
    }
  *(half8*)(out + (((long long)n * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * C + c8 * 8) = m;
}

int launch_maxpool(const half_t* in, half_t* out, int N, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s) {
  const long long total = (long long)N * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, in, out, N, Hi, Wi, Ho, Wo, C);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

// out[n][y][x][:] = in[n][2y][2x][:]; both with halo 1.
__global__ __launch_bounds__(256) void subsample2_kernel(const half_t* in, half_t* out, int N, int Hi, int Wi, int Ho, int Wo, int C) {
  const int cv = C >> 3;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * Ho * Wo * cv;
  if (gid >= total) return;
  const int c8 = (int)(gid % cv);
  long long t = gid / cv;
  const int x = (int)(t % Wo); t /= Wo;
  const int y = (int)(t % Ho);
  const int n = (int)(t / Ho);
  const half8 v = *(const half8*)(in + (((long long)n * (Hi + 2) + 2 * y + 1) * (Wi + 2) + 2 * x + 1) * C + c8 * 8);
  *(half8*)(out + (((long long)n * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * C + c8 * 8) = v;
}

int launch_subsample2(const half_t* in, half_t* out, int N, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s) {
  const long long total = (long long)N * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(subsample2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, in, out, N, Hi, Wi, Ho, Wo, C);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

// ---- fp32 validation variants (rs_spec.precision == 1): same indexing, float storage ----
__global__ __launch_bounds__(256) void maxpool3x3s2_f32_kernel(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, int C) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * Ho * Wo * C;
  if (gid >= total) return;
  const int c = (int)(gid % C);
  long long t = gid / C;
  const int x = (int)(t % Wo); t /= Wo;
  const int y = (int)(t % Ho);
  const int n = (int)(t / Ho);
  float m = 0.f;
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx) {
      const float v = in[(((long long)n * (Hi + 2) + 2 * y + dy) * (Wi + 2) + 2 * x + dx) * C + c];
      m = v > m ? v : m;
    }
  out[(((long long)n * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * C + c] = m;
}
__global__ __launch_bounds__(256) void subsample2_f32_kernel(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, int C) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * Ho * Wo * C;
  if (gid >= total) return;
  const int c = (int)(gid % C);
  long long t = gid / C;
  const int x = (int)(t % Wo); t /= Wo;
  const int y = (int)(t % Ho);
  const int n = (int)(t / Ho);
  out[(((long long)n * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * C + c] = in[(((long long)n * (Hi + 2) + 2 * y + 1) * (Wi + 2) + 2 * x + 1) * C + c];
}
int launch_maxpool_f32(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s) {
  const long long total = (long long)N * Ho * Wo * C;
  hipLaunchKernelGGL(maxpool3x3s2_f32_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, in, out, N, Hi, Wi, Ho, Wo, C);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
int launch_subsample2_f32(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s) {
  const long long total = (long long)N * Ho * Wo * C;
  hipLaunchKernelGGL(subsample2_f32_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, in, out, N, Hi, Wi, Ho, Wo, C);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
