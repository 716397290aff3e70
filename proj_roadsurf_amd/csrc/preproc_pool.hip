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
      const float f = rs_fdiv((float)v - p.mean[c], p.stdv[c]);
      o[c] = (half_t)f;
      of[c] = f;
    }
  }
  const long long oidx = (((long long)n * p.out_Hp + Y + 3) * p.out_Wp + X + 3) * 4;
  if (p.out_f32 == 1) {
    *(f32x4*)((float*)p.out + oidx) = f32x4{of[0], of[1], of[2], of[3]};
  } else {
    *(half4*)(p.out + oidx) = o;
    if (p.out_f32 == 2) {      // split-operand mode: the lo plane holds what fp16 rounding left behind
      half4 lo;
#pragma unroll
      for (int c = 0; c < 4; ++c) lo[c] = (half_t)(of[c] - (float)o[c]);
      *(half4*)(p.out + p.out_lo + oidx) = lo;
    }
  }
}

// Tile form of the same arithmetic: a block = 64 output columns x 16 output rows of one tile.  Pillow's tap ranges are monotone in the
// output coordinate, so the source pixels a block needs are a rectangle; it is staged ONCE into LDS with aligned dword loads and every
// thread then walks its taps there.  The per-pixel kernel above issues ny * nx * C byte loads from global memory per output pixel and is
// bound by the texture-address path (0.136 ms per batch of 16 512 -> 800 tiles, 0.7 TB/s); it stays as the fallback for tap tables or
// resize ratios whose rectangle does not fit the LDS budget below.
constexpr int PP_ROWS = 16, PP_LDS_BYTES = 40 * 1024, PP_KMAX = 8;

// C (bands), NH / NV (horizontal / vertical pass needed) are compile-time: with run-time values every per-channel statement is a branch
template <int C, bool NH, bool NV>
__global__ __launch_bounds__(256) void preprocess_tile_kernel(const PreprocParams p, const long long src_dwords) {
  __shared__ uint32_t src[PP_LDS_BYTES / 4];
  const int tid = threadIdx.x, lx = tid & 63;
  const int ly = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int X0 = blockIdx.x * 64, Y0 = blockIdx.y * PP_ROWS, n = blockIdx.z;
  const int Xl = min(X0 + 63, p.new_w - 1), Yl = min(Y0 + PP_ROWS - 1, p.new_h - 1);
  // source rectangle of the block (block-uniform)
  const int cx0 = NH ? p.hb[X0 * 2] : X0;
  const int cx1 = NH ? p.hb[Xl * 2] + p.hb[Xl * 2 + 1] : Xl + 1;
  const int ry0 = NV ? p.vb[Y0 * 2] : Y0;
  const int ry1 = NV ? p.vb[Yl * 2] + p.vb[Yl * 2 + 1] : Yl + 1;
  const int pitch_dw = (((cx1 - cx0) * C + 3) >> 2) + 1;      // + 1: the row segment starts at any byte of its first dword
  const int nrows = ry1 - ry0;
  const long long img0 = (long long)n * p.H * p.W * C;        // byte offset of the tile in the batch buffer (4-byte aligned base)
  const uint32_t* g32 = (const uint32_t*)p.tiles;
  for (int i = tid; i < nrows * pitch_dw; i += 256) {
    const int r = i / pitch_dw, d = i - r * pitch_dw;
    long long dw = ((img0 + ((long long)(ry0 + r) * p.W + cx0) * C) >> 2) + d;
    if (dw >= src_dwords) dw = src_dwords - 1;                // slack dwords past the last pixel are never used
    src[i] = g32[dw];
  }
  __syncthreads();
  const int X = X0 + lx;
  if (X >= p.new_w) return;
  int xmin = X, nx = 1;
  int hkr[PP_KMAX];
  if (NH) {
    xmin = p.hb[X * 2]; nx = p.hb[X * 2 + 1];
    const int* hk = p.hk + (long long)X * p.ksh;
#pragma unroll
    for (int t = 0; t < PP_KMAX; ++t) hkr[t] = t < p.ksh ? hk[t] : 0;
  }
  const uint8_t* sb = (const uint8_t*)src;
  const int xoff = (xmin - cx0) * C;
#pragma unroll 1
  for (int it = 0; it < PP_ROWS / 4; ++it) {
    const int Y = Y0 + it * 4 + ly;                           // wave-uniform
    if (Y >= p.new_h) break;
    int ymin = Y, ny = 1;
    if (NV) { ymin = p.vb[Y * 2]; ny = p.vb[Y * 2 + 1]; }
    const int* vk = p.vk + (long long)Y * p.ksv;
    int vacc[4] = {1 << 21, 1 << 21, 1 << 21, 1 << 21};
    int val[4] = {0, 0, 0, 0};
    for (int ty = 0; ty < ny; ++ty) {
      const int r = ymin + ty - ry0;
      const int off = (int)((img0 + ((long long)(ymin + ty) * p.W + cx0) * C) & 3);
      const uint8_t* row = sb + r * pitch_dw * 4 + off + xoff;
      int h[4] = {0, 0, 0, 0};
      if (NH) {
        int ss[4] = {1 << 21, 1 << 21, 1 << 21, 1 << 21};
#pragma unroll
        for (int tx = 0; tx < PP_KMAX; ++tx) {
          if (tx >= p.ksh) break;                               // uniform: the tap table's width (3 when enlarging)
          if (tx < nx) {
            const uint8_t* px = row + tx * C;
            const int k = hkr[tx];
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < C) ss[c] += __mul24((int)px[c], k);   // 8-bit pixel x 23-bit Pillow coefficient: exact in 24-bit operands
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int v = ss[c] >> 22; h[c] = v < 0 ? 0 : (v > 255 ? 255 : v); }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < C) h[c] = row[c];
      }
      if (NV) {
        const int k = vk[ty];
#pragma unroll
        for (int c = 0; c < 4; ++c) vacc[c] += __mul24(h[c], k);
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
        for (int q = 0; q < 4; ++q) if (q == cs) v = NV ? (vacc[q] >> 22) : val[q];
        if (NV) v = v < 0 ? 0 : (v > 255 ? 255 : v);
        const float f = rs_fdiv((float)v - p.mean[c], p.stdv[c]);
        o[c] = (half_t)f;
        of[c] = f;
      }
    }
    const long long oidx = (((long long)n * p.out_Hp + Y + 3) * p.out_Wp + X + 3) * 4;
    if (p.out_f32 == 1) *(f32x4*)((float*)p.out + oidx) = f32x4{of[0], of[1], of[2], of[3]};
    else {
      *(half4*)(p.out + oidx) = o;
      if (p.out_f32 == 2) {
        half4 lo;
#pragma unroll
        for (int c = 0; c < 4; ++c) lo[c] = (half_t)(of[c] - (float)o[c]);
        *(half4*)(p.out + p.out_lo + oidx) = lo;
      }
    }
  }
}

int launch_preprocess(const PreprocParams& p, hipStream_t s) {
  // bound of the block's source rectangle: (output extent) * (input / output) + the tap table's width + rounding
  const long long rows = ((long long)PP_ROWS * p.H + p.new_h - 1) / p.new_h + (p.need_v ? p.ksv : 0) + 2;
  const long long cols = (64ll * p.W + p.new_w - 1) / p.new_w + (p.need_h ? p.ksh : 0) + 2;
  const long long lds = rows * (((cols * p.C + 3) >> 2) + 1) * 4;
  if (lds <= PP_LDS_BYTES && (!p.need_h || p.ksh <= PP_KMAX) && ((uintptr_t)p.tiles & 3) == 0) {
    // N tiles of the batch live in one allocation rounded up to 256 bytes (rs_engine::alloc): whole dwords up to its end exist
    const long long src_dwords = ((long long)p.N * p.H * p.W * p.C + 3) >> 2;
    const dim3 grid(cdiv(p.new_w, 64), cdiv(p.new_h, PP_ROWS), p.N);
#define RS_PP(Cc, H_, V_) hipLaunchKernelGGL((preprocess_tile_kernel<Cc, H_, V_>), grid, dim3(256), 0, s, p, src_dwords)
#define RS_PP_C(Cc) do { if (p.need_h) { if (p.need_v) RS_PP(Cc, true, true); else RS_PP(Cc, true, false); } \
                         else { if (p.need_v) RS_PP(Cc, false, true); else RS_PP(Cc, false, false); } } while (0)
    switch (p.C) {
      case 1: RS_PP_C(1); break;
      case 2: RS_PP_C(2); break;
      case 3: RS_PP_C(3); break;
      default: RS_PP_C(4); break;
    }
#undef RS_PP_C
#undef RS_PP
  } else {
    hipLaunchKernelGGL(preprocess_kernel, dim3(cdiv(p.new_w, 64), cdiv(p.new_h, 4), p.N), dim3(256), 0, s, p);
  }
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

// ---- split-operand variants (rs_spec.precision == 2): a value is hi + lo on two fp16 planes.  fp32(hi) + fp32(lo) is exact (22 significand
// bits), so the maximum is taken on the sums and the winner's two halves are copied: no re-rounding ----
__global__ __launch_bounds__(256) void maxpool3x3s2_split_kernel(const half_t* in, long long in_lo, half_t* out, long long out_lo, int N, int Hi, int Wi, int Ho, int Wo, int C) {
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
  half8 mh, ml;
  float m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { mh[i] = (half_t)0.f; ml[i] = (half_t)0.f; m[i] = 0.f; }
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const long long idx = (((long long)n * Hip + 2 * y + dy) * Wip + 2 * x + dx) * C + c8 * 8;
      const half8 vh = *(const half8*)(in + idx), vl = *(const half8*)(in + in_lo + idx);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = (float)vh[i] + (float)vl[i];
        if (v > m[i]) { m[i] = v; mh[i] = vh[i]; ml[i] = vl[i]; }
      }
    }
  const long long o = (((long long)n * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * C + c8 * 8;
  *(half8*)(out + o) = mh;
  *(half8*)(out + out_lo + o) = ml;
}
__global__ __launch_bounds__(256) void subsample2_split_kernel(const half_t* in, long long in_lo, half_t* out, long long out_lo, int N, int Hi, int Wi, int Ho, int Wo, int C) {
  const int cv = C >> 3;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * Ho * Wo * cv;
  if (gid >= total) return;
  const int c8 = (int)(gid % cv);
  long long t = gid / cv;
  const int x = (int)(t % Wo); t /= Wo;
  const int y = (int)(t % Ho);
  const int n = (int)(t / Ho);
  const long long i = (((long long)n * (Hi + 2) + 2 * y + 1) * (Wi + 2) + 2 * x + 1) * C + c8 * 8;
  const long long o = (((long long)n * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * C + c8 * 8;
  *(half8*)(out + o) = *(const half8*)(in + i);
  *(half8*)(out + out_lo + o) = *(const half8*)(in + in_lo + i);
}
int launch_maxpool_split(const half_t* in, long long in_lo, half_t* out, long long out_lo, int N, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s) {
  const long long total = (long long)N * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(maxpool3x3s2_split_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, in, in_lo, out, out_lo, N, Hi, Wi, Ho, Wo, C);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
int launch_subsample2_split(const half_t* in, long long in_lo, half_t* out, long long out_lo, int N, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s) {
  const long long total = (long long)N * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(subsample2_split_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, in, in_lo, out, out_lo, N, Hi, Wi, Ho, Wo, C);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
