// Micro-benchmark: HBM write/read rate of the conv epilogue's access pattern vs a fully coalesced one.
// pattern 0: each wave-instruction stores 64 lanes x 16 B fully contiguous (1 KiB).
// pattern 1: conv epilogue as shipped: lane (p = l&15, q = l>>4) stores 2 x 16 B at pixel p, bytes q*32 + {0,16} of a
//            128-byte channel slab (pixels are 512 B apart) -> each instruction writes 16-B pieces at 32-B stride.
// pattern 2: "64-byte run" variant: instruction k stores bytes k*64 + q*16 of the slab.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int PAT, bool WITH_READ>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, char* __restrict__ dst, long long npix) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // block = 128 pixels x 256 B (128 channels); 4 waves as 2 (px) x 2 (ch); wave = 64 px x 128 B
  const long long blk = blockIdx.x;
  const long long tile_m = blk >> 1; const int tile_n = blk & 1;
  const int wpx = wave >> 1, wch = wave & 1;
  const int p = lane & 15, q = lane >> 4;
  for (int j = 0; j < 4; ++j) {
    const long long pix = tile_m * 128 + wpx * 64 + j * 16 + p;
    if (pix >= npix) continue;
    const long long base = pix * 512 + tile_n * 256 + wch * 128;
    if (PAT == 0) {
      // same bytes, but addressed so that a wave-instruction is 1 KiB contiguous: treat the wave's 64px x 128B as linear
      const long long lin = ((tile_m * 2 + tile_n) * 4 + wave) * 8192ll + j * 2048;
      f4 a = {1, 2, 3, 4}, b = a;
      if (WITH_READ) { a = *(const f4*)(src + lin + lane * 16); b = *(const f4*)(src + lin + 1024 + lane * 16); }
      *(f4*)(dst + lin + lane * 16) = a;
      *(f4*)(dst + lin + 1024 + lane * 16) = b;
    } else if (PAT == 1) {
      f4 a = {1, 2, 3, 4}, b = a;
      if (WITH_READ) { a = *(const f4*)(src + base + q * 32); b = *(const f4*)(src + base + q * 32 + 16); }
      *(f4*)(dst + base + q * 32) = a;
      *(f4*)(dst + base + q * 32 + 16) = b;
    } else {
      f4 a = {1, 2, 3, 4}, b = a;
      if (WITH_READ) { a = *(const f4*)(src + base + q * 16); b = *(const f4*)(src + base + 64 + q * 16); }
      *(f4*)(dst + base + q * 16) = a;
      *(f4*)(dst + base + 64 + q * 16) = b;
    }
  }
}
template <int PAT, bool R> void run(const char* name, char* s, char* d, long long npix) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nb = (int)((npix + 127) / 128) * 2;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PAT, R>), dim3(nb), dim3(256), 0, 0, s, d, npix);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<PAT, R>), dim3(nb), dim3(256), 0, 0, s, d, npix);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  const double bytes = (double)npix * 512 * (R ? 2 : 1);
  printf("%-34s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
}
int main() {
  const long long npix = 640000;
  char *s, *d; hipMalloc(&s, npix * 512 + 65536); hipMalloc(&d, npix * 512 + 65536);
  hipMemset(s, 1, npix * 512); hipMemset(d, 0, npix * 512);
  run<0, false>("linear store only", s, d, npix);
  run<1, false>("epilogue(32B/lane) store only", s, d, npix);
  run<2, false>("64B-run store only", s, d, npix);
  run<0, true>("linear read+store", s, d, npix);
  run<1, true>("epilogue(32B/lane) read+store", s, d, npix);
  run<2, true>("64B-run read+store", s, d, npix);
  return 0;
}
