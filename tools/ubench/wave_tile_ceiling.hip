// LDS + MFMA ceiling of three wave-tile structures for the 256 x 256 x 64 K step of conv_deep (no global traffic, no epilogue;
// results are meaningless numbers -- only the rate counts).  Answers the question DESIGN.md section 3.1 "Round 3" leaves open: is a
// 128 x 128 wave tile (4 waves per workgroup, one per SIMD, 256 accumulator registers) worth a new kernel?
//
//   A   8 waves, wave tile 128 px x 64 ch, v_mfma_f32_16x16x32_f16   (conv_deep today: 24 fragment reads / 64 MFMAs per K step and wave)
//   B   8 waves, wave tile 128 px x 64 ch, v_mfma_f32_32x32x16_f16   (same LDS bytes, half the MFMA instructions)
//   C   4 waves, wave tile 128 px x 128 ch, v_mfma_f32_16x16x32_f16  (32 reads / 128 MFMAs: 2/3 of the LDS bytes per FLOP)
//   D   4 waves, wave tile 128 px x 128 ch, v_mfma_f32_32x32x16_f16
//
// Every variant software-pipelines half K steps the way conv_deep does (reads of the next half in flight under the MFMAs of this
// one), one workgroup per CU (160 KB of LDS claimed), fragment addresses follow conv_deep's 128-byte XOR-swizzled rows and change with
// the K step (three activation stages, two weight stages) so nothing is hoisted.
//
//   hipcc --offload-arch=gfx950 -O3 -o wave_tile_ceiling tools/ubench/wave_tile_ceiling.hip && ./wave_tile_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ASTAGE = 256 * 128, WSTAGE = 256 * 128, W_BASE = 3 * ASTAGE, LDS_BYTES = 3 * ASTAGE + 2 * WSTAGE;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MI = 16-channel blocks per wave (4: 64 ch, 8: 128 ch); 8 pixel blocks of 16 per wave; WCH = channel waves
template <int MI, int WCH>
__global__ __launch_bounds__(WCH * 2 * 64) void k16(float* out, int nk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wpx = wave / WCH, wch = wave % WCH;
  for (int i = tid; i < LDS_BYTES / 4; i += blockDim.x) ((float*)smem)[i] = 0.0001f * (float)(i & 1023);
  __syncthreads();
  const int fi = lane & 15, fq = lane >> 4, fkey = lane & 7;
  const int wrow = W_BASE + (wch * MI * 16 + (fi >> 2) * 4 * MI + (fi & 3)) * 128;
  const int xrow = (wpx * 128 + fi) * 128;
  const int c0 = (fq ^ fkey) * 16, c1 = ((4 + fq) ^ fkey) * 16;
  f32x4 acc[MI][8];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  half8 wf0[MI], xf0[8], wf1[MI], xf1[8];
  auto reads = [&](half8* wf, half8* xf, int ab, int wb, int c) {
#pragma unroll
    for (int i = 0; i < MI; ++i) wf[i] = *(const half8*)(smem + wrow + wb * WSTAGE + i * 512 + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) xf[j] = *(const half8*)(smem + xrow + ab * ASTAGE + j * 2048 + c);
  };
  reads(wf0, xf0, 0, 0, c0);
  int ab = 0;
  for (int t = 0; t < nk; ++t) {
    reads(wf1, xf1, ab, t & 1, c1);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[i], xf0[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_barrier();
    const int an = ab == 2 ? 0 : ab + 1;
    reads(wf0, xf0, an, (t + 1) & 1, c0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[i], xf1[j], acc[i][j], 0, 0, 0);
    ab = an;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

// 32x32x16: MB = 32-channel blocks per wave (2: 64 ch, 4: 128 ch); 4 pixel blocks of 32 per wave.  A fragment: lane (r = lane & 31,
// h = lane >> 5) reads 8 halfs = k 8h .. 8h+7 of row r for each of the two 16-deep steps of a 32-deep half K step.
template <int MB, int WCH>
__global__ __launch_bounds__(WCH * 2 * 64) void k32(float* out, int nk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wpx = wave / WCH, wch = wave % WCH;
  for (int i = tid; i < LDS_BYTES / 4; i += blockDim.x) ((float*)smem)[i] = 0.0001f * (float)(i & 1023);
  __syncthreads();
  const int r = lane & 31, h = lane >> 5, fkey = lane & 7;
  const int wrow = W_BASE + (wch * MB * 32 + r) * 128;
  const int xrow = (wpx * 128 + r) * 128;
  // a 64-deep K step = 4 MFMA steps of 16: step s reads chunk 2s + h (16 bytes) of the 128-byte row, XOR-swizzled
  int cs[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) cs[s] = ((2 * s + h) ^ fkey) * 16;
  f32x16 acc[MB][4];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  half8 wf0[2][MB], xf0[2][4], wf1[2][MB], xf1[2][4];
  auto reads = [&](half8 (*wf)[MB], half8 (*xf)[4], int ab, int wb, int half_) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < MB; ++i) wf[s][i] = *(const half8*)(smem + wrow + wb * WSTAGE + i * 4096 + cs[half_ * 2 + s]);
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[s][j] = *(const half8*)(smem + xrow + ab * ASTAGE + j * 4096 + cs[half_ * 2 + s]);
    }
  };
  reads(wf0, xf0, 0, 0, 0);
  int ab = 0;
  for (int t = 0; t < nk; ++t) {
    reads(wf1, xf1, ab, t & 1, 1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf0[s][i], xf0[s][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_barrier();
    const int an = ab == 2 ? 0 : ab + 1;
    reads(wf0, xf0, an, (t + 1) & 1, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf1[s][i], xf1[s][j], acc[i][j], 0, 0, 0);
    ab = an;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

template <typename K>
static void run(const char* name, K kern, int threads, int nk, int grid, float* out) {
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), LDS_BYTES, 0, out, nk);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), LDS_BYTES, 0, out, nk);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = (double)grid * reps * nk * 2.0 * 256 * 256 * 64;       // one 256 x 256 x 64 K step per workgroup and iteration
  printf("%-58s %8.3f ms  %7.1f TFLOP/s\n", name, ms / reps, flop / (ms * 1e-3) / 1e12);
}

int main() {
  float* out = nullptr;
  CHECK(hipMalloc(&out, (size_t)4096 * 512 * 4));
  const int nk = 36 * 8, grid = 256 * 4;                                     // 8 tiles' worth of a K = 2304 loop per workgroup, 4 rounds
  for (int rep = 0; rep < 2; ++rep) {
    run("A  8 waves 128x64   16x16x32 (conv_deep today)", k16<4, 4>, 512, nk, grid, out);
    run("B  8 waves 128x64   32x32x16", k32<2, 4>, 512, nk, grid, out);
    run("C  4 waves 128x128  16x16x32 (acc in AGPRs)", k16<8, 2>, 256, nk, grid, out);
    run("D  4 waves 128x128  32x32x16 (acc in AGPRs)", k32<4, 2>, 256, nk, grid, out);
  }
  return 0;
}
