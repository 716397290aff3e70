// Does a small VALU kernel (the arithmetic of paste_masks_kernel: fp32 divisions, floors, bilinear sums, a threshold) compute the same bits while a matrix-core
// kernel of another stream shares the CUs?  Stream A repeats the small kernel on constant inputs into alternating output buffers and compares every result with
// the first one; stream B (optional) loops an MFMA + LDS kernel.  Build: hipcc -O3 --offload-arch=gfx950 coexec_probe.hip -o coexec_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Correctly rounded a / b for normal-range operands without v_div_scale / v_div_fmas / v_div_fixup: reciprocal refined once, quotient refined with the
// exact remainder (Markstein); the compiler's own expansion is the same iteration wrapped in a scaling for extreme exponents
__device__ __forceinline__ float div_nr(float a, float b) {
  float y = __builtin_amdgcn_rcpf(b);
  y = __builtin_fmaf(__builtin_fmaf(-b, y, 1.0f), y, y);
  float q = a * y;
  const float r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}

// which of the three instructions?  3 = the software iteration followed by v_div_fixup_f32 alone; 4 = v_div_scale_f32 on both operands (a scale of 1 for these
// magnitudes), the software iteration, no v_div_fmas_f32 / v_div_fixup_f32; 5 = v_div_scale + the iteration with v_div_fmas_f32 as its last step, no fixup
__device__ __forceinline__ float div_var(float a, float b, int which) {
  if (which == 3) return __builtin_amdgcn_div_fixupf(div_nr(a, b), b, a);
  if (which == 4) {
    bool f0, f1;
    const float bs = __builtin_amdgcn_div_scalef(a, b, false, &f0), as = __builtin_amdgcn_div_scalef(a, b, true, &f1);
    return div_nr(as, bs);
  }
  bool f0, f1;
  const float bs = __builtin_amdgcn_div_scalef(a, b, false, &f0), as = __builtin_amdgcn_div_scalef(a, b, true, &f1);
  float y = __builtin_amdgcn_rcpf(bs);
  y = __builtin_fmaf(__builtin_fmaf(-bs, y, 1.0f), y, y);
  const float q0 = as * y;
  const float q1 = __builtin_fmaf(__builtin_fmaf(-bs, q0, as), y, q0);
  return __builtin_amdgcn_div_fmasf(__builtin_fmaf(-bs, q1, as), y, q1, f1);
}

template <int DIV>
__global__ __launch_bounds__(256) void paste_like(const float* probs, const float* boxes, unsigned* out, int n_det, int H, int Ww, int S, float thr) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)H * Ww;
  if (gid >= (long long)n_det * per) return;
  const int det = (int)(gid / per);
  const int rem = (int)(gid - (long long)det * per);
  const int y = rem / Ww, xw = rem - y * Ww;
  const float* bx = boxes + (long long)det * 4;
  const float x0 = bx[0], y0 = bx[1], x1 = bx[2], y1 = bx[3];
  const float* m = probs + (long long)det * S * S;
  unsigned word = 0;
  const float gy = (DIV == 0 ? ((float)y + 0.5f - y0) / (y1 - y0) : DIV == 2 || DIV >= 6 ? div_nr((float)y + 0.5f - y0, y1 - y0) : DIV >= 3 ? div_var((float)y + 0.5f - y0, y1 - y0, DIV) : ((float)y + 0.5f - y0) * __builtin_amdgcn_rcpf(y1 - y0)) * 2.f - 1.f;
  const float iy = ((gy + 1.f) * (float)S - 1.f) * 0.5f;
  const float fy = floorf(iy);
  const int iy0 = (int)fy, iy1 = iy0 + 1;
  const float wy1 = iy - fy, wy0 = 1.f - wy1;
  if (iy1 >= 0 && iy0 < S) {
    for (int b = 0; b < 32; ++b) {
      const int x = xw * 32 + b;
      const float gx = (DIV == 0 ? ((float)x + 0.5f - x0) / (x1 - x0) : DIV == 2 || DIV >= 6 ? div_nr((float)x + 0.5f - x0, x1 - x0) : DIV >= 3 ? div_var((float)x + 0.5f - x0, x1 - x0, DIV) : ((float)x + 0.5f - x0) * __builtin_amdgcn_rcpf(x1 - x0)) * 2.f - 1.f;
      const float ix = ((gx + 1.f) * (float)S - 1.f) * 0.5f;
      const float fx = floorf(ix);
      const int ix0 = (int)fx, ix1 = ix0 + 1;
      if (ix1 < 0 || ix0 >= S) continue;
      const float wx1 = ix - fx, wx0 = 1.f - wx1;
      float v = 0.f;
      if (iy0 >= 0 && ix0 >= 0) v += m[iy0 * S + ix0] * (wx0 * wy0);
      if (iy0 >= 0 && ix1 < S) v += m[iy0 * S + ix1] * (wx1 * wy0);
      if (iy1 < S && ix0 >= 0) v += m[iy1 * S + ix0] * (wx0 * wy1);
      if (iy1 < S && ix1 < S) v += m[iy1 * S + ix1] * (wx1 * wy1);
      if (v >= thr) word |= 1u << b;
      // flavours 6-9: other compiler-expanded functions, their low result bits mixed into the word (division by the written-out iteration)
      if (DIV == 6) word ^= (__float_as_uint(sqrtf(fabsf(gx) + 1.f)) & 1u) << b;
      if (DIV == 7) word ^= (__float_as_uint(expf(gx * 0.25f)) & 1u) << b;
      if (DIV == 8) word ^= (__float_as_uint(logf(fabsf(gx) + 0.5f)) & 1u) << b;
      if (DIV == 9) word ^= ((unsigned)((long long)(gx * 1000.f) / (long long)(b + 3)) & 1u) << b;
      if (DIV == 10) word ^= ((unsigned)__double2loint((double)gx / ((double)fabsf(x1 - x0) + 1.0)) & 1u) << b;      // the compiler's fp64 division
    }
  }
  out[gid] = word;
}

__global__ __launch_bounds__(512) void mfma_busy(float* sink, int iters) {
  extern __shared__ char lds[];
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  half8* l = (half8*)lds;
  l[threadIdx.x] = a;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    const half8 x = l[(threadIdx.x + it) & 511];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
}

// a kernel shaped like the repository's convolutions: global loads -> LDS, barrier, fragment reads, MFMAs, fp16 stores; mode bits: 1 = LDS-DMA staging,
// 4 = no MFMA (one scalar product instead), 8 = no output stores, 16 = no global loads / LDS staging  (command line: 2 + mode)
template <int mode>
__global__ __launch_bounds__(256) void gemm_like(const _Float16* in, const _Float16* w, _Float16* out, int tiles, int ksteps) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    f32x4 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < ksteps; ++k) {
      __syncthreads();
      const _Float16* src = in + ((long long)t * ksteps + k) * 128 * 64 + tid * 8;
      for (int r = 0; r < 4 && !(mode & 16); ++r) {
        if (mode & 1) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + r * 2048),
                                           (__attribute__((address_space(3))) void*)(lds + r * 4096 + wave * 1024), 16, 0, 0);
        } else {
          *(half8*)(lds + r * 4096 + tid * 16) = *(const half8*)(src + r * 2048);
        }
        *(half8*)(lds + 16384 + r * 4096 + tid * 16) = *(const half8*)(w + (long long)(k & 7) * 128 * 64 + r * 2048 + tid * 8);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      half8 xf[2], wf[4];
      for (int j = 0; j < 2; ++j) xf[j] = *(const half8*)(lds + ((wave * 32 + j * 16 + (lane & 15)) * 128 + (((lane >> 4) ^ (lane & 7)) * 16)) % 16384);
      for (int i = 0; i < 4; ++i) wf[i] = *(const half8*)(lds + 16384 + ((i * 16 + (lane & 15)) * 128 + (((lane >> 4) ^ (lane & 7)) * 16)) % 16384);
      if (!(mode & 4)) {
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      } else {
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j][0] += (float)wf[i][0] * (float)xf[j][0];
      }
    }
    if (mode & 8) { if (acc[0][0][0] == 123.456f) out[0] = (_Float16)1.f; continue; }
    for (int j = 0; j < 2; ++j) {
      _Float16* op = out + ((long long)t * 128 + wave * 32 + j * 16 + (lane & 15)) * 64 + (lane >> 4) * 16;
      for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 4; ++r) {
          float f = acc[i][j][r];
          f = f > 0.f ? f : 0.f;
          op[i * 4 + r] = (_Float16)f;
        }
      }
    }
  }
}

int main(int argc, char** argv) {
  const int with_b = argc > 1 ? atoi(argv[1]) : 1;
  const int rounds = argc > 2 ? atoi(argv[2]) : 3000;
  const int n_det = 300, H = 256, Ww = 8, S = 28;
  const long long words = (long long)n_det * H * Ww;
  std::vector<float> hp((size_t)n_det * S * S), hb((size_t)n_det * 4);
  srand(5);
  for (auto& v : hp) v = (float)rand() / RAND_MAX;
  for (int d = 0; d < n_det; ++d) {
    const float cx = 20.f + 216.f * rand() / RAND_MAX, cy = 20.f + 216.f * rand() / RAND_MAX, w = 8.f + 100.f * rand() / RAND_MAX, h = 8.f + 100.f * rand() / RAND_MAX;
    hb[d * 4] = cx - w / 2; hb[d * 4 + 1] = cy - h / 2; hb[d * 4 + 2] = cx + w / 2; hb[d * 4 + 3] = cy + h / 2;
  }
  float *probs, *boxes, *sink; unsigned *out[2];
  CK(hipMalloc(&probs, hp.size() * 4)); CK(hipMalloc(&boxes, hb.size() * 4)); CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&out[0], words * 4)); CK(hipMalloc(&out[1], words * 4));
  CK(hipMemcpy(probs, hp.data(), hp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(boxes, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  _Float16 *gin, *gw, *gout;
  CK(hipMalloc(&gin, 4096ll * 8 * 128 * 64 * 2)); CK(hipMalloc(&gw, 8ll * 128 * 64 * 2)); CK(hipMalloc(&gout, 4096ll * 128 * 64 * 2));
  CK(hipMemset(gin, 0x11, 4096ll * 8 * 128 * 64 * 2)); CK(hipMemset(gw, 0x12, 8ll * 128 * 64 * 2));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  CK(hipFuncSetAttribute((const void*)mfma_busy, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  const int grid = (int)((words + 255) / 256);
  std::vector<unsigned> ref(words), got(words);
  const int divk = argc > 3 ? atoi(argv[3]) : 0;
  auto launch_a = [&](unsigned* o) {
    if (divk == 1) hipLaunchKernelGGL(paste_like<1>, dim3(grid), dim3(256), 0, a, probs, boxes, o, n_det, H, Ww, S, 0.5f);
    else if (divk == 2) hipLaunchKernelGGL(paste_like<2>, dim3(grid), dim3(256), 0, a, probs, boxes, o, n_det, H, Ww, S, 0.5f);
    else if (divk == 3) hipLaunchKernelGGL(paste_like<3>, dim3(grid), dim3(256), 0, a, probs, boxes, o, n_det, H, Ww, S, 0.5f);
    else if (divk == 4) hipLaunchKernelGGL(paste_like<4>, dim3(grid), dim3(256), 0, a, probs, boxes, o, n_det, H, Ww, S, 0.5f);
    else if (divk == 5) hipLaunchKernelGGL(paste_like<5>, dim3(grid), dim3(256), 0, a, probs, boxes, o, n_det, H, Ww, S, 0.5f);
    else if (divk == 6) hipLaunchKernelGGL(paste_like<6>, dim3(grid), dim3(256), 0, a, probs, boxes, o, n_det, H, Ww, S, 0.5f);
    else if (divk == 7) hipLaunchKernelGGL(paste_like<7>, dim3(grid), dim3(256), 0, a, probs, boxes, o, n_det, H, Ww, S, 0.5f);
    else if (divk == 8) hipLaunchKernelGGL(paste_like<8>, dim3(grid), dim3(256), 0, a, probs, boxes, o, n_det, H, Ww, S, 0.5f);
    else if (divk == 9) hipLaunchKernelGGL(paste_like<9>, dim3(grid), dim3(256), 0, a, probs, boxes, o, n_det, H, Ww, S, 0.5f);
    else if (divk == 10) hipLaunchKernelGGL(paste_like<10>, dim3(grid), dim3(256), 0, a, probs, boxes, o, n_det, H, Ww, S, 0.5f);
    else hipLaunchKernelGGL(paste_like<0>, dim3(grid), dim3(256), 0, a, probs, boxes, o, n_det, H, Ww, S, 0.5f);
  };
  launch_a(out[0]);
  CK(hipStreamSynchronize(a));
  CK(hipMemcpy(ref.data(), out[0], words * 4, hipMemcpyDeviceToHost));
  {
    std::vector<unsigned> ieee(words);
    hipLaunchKernelGGL(paste_like<0>, dim3(grid), dim3(256), 0, a, probs, boxes, out[1], n_det, H, Ww, S, 0.5f);
    CK(hipStreamSynchronize(a));
    CK(hipMemcpy(ieee.data(), out[1], words * 4, hipMemcpyDeviceToHost));
    long long d = 0;
    for (long long i = 0; i < words; ++i) d += ieee[i] != ref[i];
    printf("division flavour %d against the compiler's IEEE division, both alone: %lld of %lld words differ\n", divk, d, words);
  }
  long long bad_rounds = 0, bad_words = 0;
  for (int r = 0; r < rounds; ++r) {
    if (with_b == 1) for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(mfma_busy, dim3(256), dim3(512), 150 * 1024, b, sink, 3000);
    if (with_b >= 2) for (int k = 0; k < 6; ++k) {
      switch (with_b - 2) {
#define RS_CASE(M) case M: hipLaunchKernelGGL(gemm_like<M>, dim3(1024), dim3(256), 32768, b, gin, gw, gout, 4096, 8); break;
        RS_CASE(0) RS_CASE(1) RS_CASE(4) RS_CASE(5) RS_CASE(8) RS_CASE(9) RS_CASE(12) RS_CASE(13) RS_CASE(16) RS_CASE(20) RS_CASE(24) RS_CASE(28)
        default: printf("mode not built\n"); return 1;
      }
    }
    unsigned* o = out[r & 1];
    CK(hipMemsetAsync(o, 0xA5, words * 4, a));
    launch_a(o);
    CK(hipMemcpyAsync(got.data(), o, words * 4, hipMemcpyDeviceToHost, a));
    CK(hipStreamSynchronize(a));
    long long nb = 0;
    for (long long i = 0; i < words; ++i) nb += got[i] != ref[i];
    bad_words += nb; bad_rounds += nb != 0;
    if (with_b) CK(hipStreamSynchronize(b));
  }
  printf("matrix-core kernel on a second stream: %d; %d rounds of the small VALU kernel on constant inputs: %lld rounds differ from the first result (%lld words)\n",
         with_b, rounds, bad_rounds, bad_words);
  return 0;
}
