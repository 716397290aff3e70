// Does a kernel see the stores of the kernel launched before it ON THE SAME STREAM while another stream keeps the chip busy?
// Stream A: W(i) writes word pattern i over a buffer (many small workgroups), then V(i) checks it and counts what it still sees of i - 1.
// Stream B (optional): a long-running kernel loop that occupies the CUs.  Build: hipcc -O2 --offload-arch=gfx950 stream_order_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void write_k(unsigned* buf, long long n, unsigned v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = v ^ (unsigned)i;
}
__global__ void verify_k(const unsigned* buf, long long n, unsigned v, int* bad) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && buf[i] != (v ^ (unsigned)i)) atomicAdd(bad, 1);
}
__global__ void busy_k(float* x, int iters) {
  float a = x[blockIdx.x * blockDim.x + threadIdx.x];
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  x[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main(int argc, char** argv) {
  const int with_b = argc > 1 ? atoi(argv[1]) : 1;
  const int rounds = argc > 2 ? atoi(argv[2]) : 20000;
  const long long n = argc > 3 ? atoll(argv[3]) : 3 * 100 * 256 * 8;      // words of a masks buffer of 3 tiles of 256 x 256
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  unsigned* buf; int* bad; float* x;
  CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&x, 256 * 1024 * 4));
  CK(hipMemset(bad, 0, 4)); CK(hipMemset(x, 0, 256 * 1024 * 4)); CK(hipMemset(buf, 0, n * 4));
  const int grid = (int)((n + 255) / 256);
  for (int r = 0; r < rounds; ++r) {
    if (with_b && (r % 4) == 0) hipLaunchKernelGGL(busy_k, dim3(1024), dim3(256), 0, b, x, 20000);
    hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, a, buf, n, (unsigned)r * 2654435761u);
    hipLaunchKernelGGL(verify_k, dim3(grid), dim3(256), 0, a, buf, n, (unsigned)r * 2654435761u, bad);
    if ((r & 255) == 255) CK(hipStreamSynchronize(a));
  }
  CK(hipDeviceSynchronize());
  // second experiment: write -> copy to the host on the same stream -> check on the host, every round (what a fetch of the detection masks does)
  {
    unsigned* hbuf;
    CK(hipHostMalloc(&hbuf, n * 4));
    long long stale_words = 0, stale_rounds = 0;
    const int r2 = rounds / 10;
    for (int r = 0; r < r2; ++r) {
      const unsigned v = (unsigned)(r + 77) * 2654435761u;
      if (with_b) hipLaunchKernelGGL(busy_k, dim3(1024), dim3(256), 0, b, x, 20000);
      hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, a, buf, n, v);
      CK(hipMemcpyAsync(hbuf, buf, n * 4, hipMemcpyDeviceToHost, a));
      CK(hipStreamSynchronize(a));
      long long bad_here = 0;
      for (long long i = 0; i < n; ++i) bad_here += hbuf[i] != (v ^ (unsigned)i);
      stale_words += bad_here;
      stale_rounds += bad_here != 0;
    }
    printf("second stream busy: %d, %d rounds of write -> D2H copy -> host check: %lld stale words in %lld rounds\n", with_b, r2, stale_words, stale_rounds);
    CK(hipHostFree(hbuf));
  }
  int h = 0;
  CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
  printf("second stream busy: %d, %d rounds of write -> verify on one stream, %lld words: %d stale words seen by the verifying kernel\n", with_b, rounds, n, h);
  return 0;
}
