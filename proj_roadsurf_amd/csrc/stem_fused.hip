// BasicStem in one launch: conv 7x7 stride 2 pad 3 (FrozenBN folded) + ReLU + max_pool2d(3, 2, 1)
// [EXT d2: modeling/backbone/resnet.py BasicStem; R:config/detectron2_config_3bands.yaml:101,110].
//
// Layer by layer the stem moves, per batch of 16 800x800 inputs, 82 MB in, the 400x400x64 conv output 327 MB out and 327 MB back in, and
// 82 MB of pooled output: 0.187 + 0.087 ms, both bound by that 327 MB map.  Here a workgroup owns an 8x8 patch of POOLED pixels: it
// computes the 17x17 conv outputs the patch's 3x3 windows cover (13 % of them twice, once per neighbouring patch), keeps them in LDS and
// writes only the pooled values: 82 MB in, 82 MB out.
//
// The convolution is the same implicit GEMM as the stand-alone stem (conv_igemm.hip, small-Cin path): weights [64][256] with
// k = kh*32 + kw*4 + c (tap rows padded to 8 taps, 4 channels per pixel; weights.py STEM_KW_PAD), v_mfma_f32_16x16x32_f16 with the
// weights as the A operand, one 32-deep K step per filter row kh, steps in the same order -- so every conv output, and hence every
// pooled value, is BIT-identical to the two-kernel path.  What differs is where the B operand comes from: the 39x39 input pixels of the
// patch are staged ONCE into LDS (8 bytes per pixel) and a lane's 8 k-values of a step -- two horizontally adjacent taps x 4 channels --
// are 16 contiguous bytes of that image, read straight into the fragment; no im2col copy exists anywhere.
#include "common.h"

namespace {

constexpr int PR = 8, PC = 8;                    // pooled patch
constexpr int CR = 2 * PR + 1, CC = 2 * PC + 1;   // conv patch: 17 x 17
constexpr int IR = 2 * CR + 5, IP = 40;          // input patch: 39 rows x 39 columns, row pitch 40 pixels (the zero-weight 8th tap reads column 39)
constexpr int NPX = CR * CC;                      // 289 conv pixels
constexpr int NBLK = (NPX + 15) / 16;             // 19 blocks of 16 pixels
constexpr int NT = 256, NW = 4, BPW = (NBLK + NW - 1) / NW;   // blocks per wave: 5
constexpr int W_BYTES = 7 * 4 * 64 * 16;          // weight fragments [kh 7][mi 4][lane 64][16 B] = 28 KB
constexpr int IN_BYTES = IR * IP * 8;             // 12 480 B
constexpr int OP = 144;                           // conv-output pixel pitch: 128 B of channels + 16: stores (8 B, pixel-strided) and the pooling's 16 B
                                                  // reads are then spread over the banks (pitch 128: 16-way conflicts, 0.06 ms of a 0.26 ms kernel)
constexpr int OUT_BYTES = NPX * OP;               // conv outputs [289][64] fp16, aliasing weights + input after the GEMM
constexpr int LDS_BYTES = OUT_BYTES > W_BYTES + IN_BYTES ? OUT_BYTES : W_BYTES + IN_BYTES;     // 41 616 B: three workgroups per CU

template <int DBG>
__global__ __launch_bounds__(NT) void stem_pool_kernel(const StemPoolParams p) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a0 = blockIdx.y * PR, b0 = blockIdx.x * PC, n = blockIdx.z;
  char* wl = smem;
  char* il = smem + W_BYTES;

  // ---- weights -> LDS: already in fragment order in memory (weights.py "stem.conv1f": fragment (kh, mi) = 1 KB, lane l holds
  // W[mi*16 + (l & 15)][kh*32 + (l >> 4)*8 .. +7]), so this is a linear 28 KB copy: 7 LDS-DMA pieces per wave
  if (!(DBG & 1)) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int piece = i * NW + wave;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.wf + (long long)piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(wl + piece * 1024), 16, 0, 0);
    }
  }
  // ---- input patch -> LDS: buffer rows 4 a0 - 2 .. +38 of the halo-3 image (conv row cy reads rows 2 cy - 3 .. 2 cy + 3), zeros outside
  {
    const int by0 = 4 * a0 - 2, bx0 = 4 * b0 - 2;              // buffer coordinates (halo included) of patch pixel (0, 0)
    const half_t* img = p.in + (long long)n * p.in_Hp * p.in_Wp * 4;
    // 16-byte chunks = pixel pairs; bx0 and in_Wp are even, so a pair is inside the row or outside it as a whole (column 39 only meets
    // the zero weights of the 8th tap: any finite value will do)
    if (!(DBG & 2))
    for (int t = tid; t < IR * (IP / 2); t += NT) {
      const int r = t / (IP / 2), c = (t - r * (IP / 2)) * 2;
      const int by = by0 + r, bx = bx0 + c;
      half8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)0.f;
      if (by >= 0 && by < p.in_Hp && bx >= 0 && bx + 1 < p.in_Wp) v = *(const half8*)(img + ((long long)by * p.in_Wp + bx) * 4);
      *(half8*)(il + (r * IP + c) * 8) = v;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA pieces of the weights
  __syncthreads();

  // ---- implicit GEMM: D[channel][pixel], 7 K steps (one filter row each)
  const int fi = lane & 15, fq = lane >> 4;
  int boff[BPW];                                 // LDS byte offset of this lane's pixel (tap row 0, tap 2 fq) per block
#pragma unroll
  for (int b = 0; b < BPW; ++b) {
    const int blk = wave + b * NW;
    int px = blk * 16 + fi;
    if (px >= NPX) px = NPX - 1;                 // slots past the patch (the last block's tail, wave 3's fifth block) compute a valid
                                                 // pixel again and are never stored: every wave runs the same branch-free loop
    const int i = px / CC, j = px - i * CC;
    boff[b] = ((2 * i) * IP + 2 * j + 2 * fq) * 8;
  }
  f32x4 bv[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) bv[mi] = *(const f32x4*)(p.bias + mi * 16 + fq * 4);
  f32x4 acc[BPW][4];
#pragma unroll
  for (int b = 0; b < BPW; ++b)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[b][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kh = 0; kh < ((DBG & 4) ? 1 : 7); ++kh) {
    half8 wf[4], xf[BPW];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) wf[mi] = *(const half8*)(wl + ((kh * 4 + mi) * 64 + lane) * 16);
#pragma unroll
    for (int b = 0; b < BPW; ++b) xf[b] = *(const half8*)(il + boff[b] + kh * IP * 8);
#pragma unroll
    for (int b = 0; b < BPW; ++b)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[b][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mi], xf[b], acc[b][mi], 0, 0, 0);
  }
  __syncthreads();                               // every wave is done with the weights and the input: their space becomes the conv-output patch

  // ---- bias + ReLU, fp16 exactly as the stand-alone conv stores it; conv pixels outside the map are the pooling's padding (the map is
  // >= 0 after ReLU, so 0 stands for -inf as in maxpool3x3s2_kernel)
  char* ol = smem;
#pragma unroll
  for (int b = 0; b < BPW; ++b) {
    const int blk = wave + b * NW;
    const int px = blk * 16 + fi;
    if (blk >= NBLK || px >= NPX) continue;
    const int i = px / CC, j = px - i * CC;
    const int cy = 2 * a0 - 1 + i, cx = 2 * b0 - 1 + j;
    const bool inside = cy >= 0 && cy < p.Hc && cx >= 0 && cx < p.Wc;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      half4 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float f = acc[b][mi][r] + bv[mi][r];
        f = f > 0.f ? f : 0.f;
        f = f > 65504.f ? 65504.f : f;
        h[r] = inside ? (half_t)f : (half_t)0.f;
      }
      if (!(DBG & 8)) *(half4*)(ol + px * OP + (mi * 16 + fq * 4) * 2) = h;
    }
  }
  __syncthreads();

  // ---- 3x3 stride-2 max over the patch: thread = (pooled pixel, 16 channels)
  {
    const int pp = tid >> 2, cg = tid & 3;
    const int a = pp / PC, b = pp - a * PC;
    const int py = a0 + a, pxo = b0 + b;
    if (py < p.Hq && pxo < p.Wq && !(DBG & 16)) {
      half8 m0, m1;
#pragma unroll
      for (int e = 0; e < 8; ++e) { m0[e] = (half_t)0.f; m1[e] = (half_t)0.f; }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const char* q = ol + ((2 * a + dy) * CC + 2 * b + dx) * OP + cg * 32;
          const half8 v0 = *(const half8*)q, v1 = *(const half8*)(q + 16);
#pragma unroll
          for (int e = 0; e < 8; ++e) { m0[e] = v0[e] > m0[e] ? v0[e] : m0[e]; m1[e] = v1[e] > m1[e] ? v1[e] : m1[e]; }
        }
      half_t* op = p.out + (((long long)n * (p.Hq + 2) + py + 1) * (p.Wq + 2) + pxo + 1) * 64 + cg * 16;
      *(half8*)op = m0;
      *(half8*)(op + 8) = m1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Split-operand mode (rs_spec.precision == 2): input planes hi + lo, weights hi + lo of the row-scaled matrix, three products per real product in the
// stand-alone split stem's order -- W_hi.X_hi over the seven filter rows, then W_hi.X_lo, then W_lo.X_hi (conv_igemm.hip, SMALLC: pass outermost) -- so
// every conv value and every pooled (hi, lo) pair is bit-identical to that path followed by maxpool3x3s2_split_kernel.  Twice the bytes per value do not
// fit three ways, so: the conv-output patch goes through LDS in four quarters of 16 channels ([289 pixels][16 hi | 16 lo], 23 KB beside the 25 KB of the
// two input planes = 48 KB, three workgroups per CU), and the weight fragments -- 56 KB, the same for every workgroup -- are read straight from L1 / L2
// into registers (through LDS they would add 4 fragment reads to the 5 input-fragment reads per 20 matrix instructions, and the LDS port is what bounds
// this GEMM).  Measured at batch 16 (ablation by switching phases off): 0.35 ms = GEMM 0.16 (its matrix work alone: 0.125) + input staging 0.05 + 10 000
// workgroups of barriers and epilogue arithmetic 0.085 + pooling 0.05 (0.12 before the key trick below); the two launches it replaces: 0.45 + 0.19 ms.
// A non-negative value v as the pair hi = fp16(v), lo = fp16(v - hi) orders like the pair itself, lexicographically: rounding is monotone, so v1 > v2
// gives hi1 >= hi2, and for equal hi the remainders decide (equal remainders after rounding = the same pair).  hi >= +0 orders by its bit pattern; lo
// (either sign) by the usual sign-flip map.  One unsigned max per element and tap replaces two conversions, an add, a compare and three selects.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pair_key(half_t h, half_t l) {
  const unsigned hb = __builtin_bit_cast(unsigned short, h), lb = __builtin_bit_cast(unsigned short, l);
  return (hb << 16) | ((lb & 0x8000u) ? (~lb & 0xffffu) : (lb | 0x8000u));
}
__device__ __forceinline__ void key_pair(unsigned k, half_t& h, half_t& l) {
  const unsigned lk = k & 0xffffu;
  h = __builtin_bit_cast(half_t, (unsigned short)(k >> 16));
  l = __builtin_bit_cast(half_t, (unsigned short)((lk & 0x8000u) ? (lk & 0x7fffu) : (~lk & 0xffffu)));
}
constexpr int S_IN = 2 * IN_BYTES;                // both input planes
constexpr int S_OP = 80;                          // conv-output pixel pitch: 16 keys of 4 B + 16
constexpr int S_OUT = NPX * S_OP;
constexpr int S_LDS = S_IN + S_OUT;               // 48 080 B

__global__ __launch_bounds__(NT, 3) void stem_pool_split_kernel(const StemPoolSplitParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_s[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a0 = blockIdx.y * PR, b0 = blockIdx.x * PC, n = blockIdx.z;
  char* il = smem_s;
  char* ol = smem_s + S_IN;

  // ---- input patch -> LDS, plane after plane (see stem_pool_kernel)
  {
    const int by0 = 4 * a0 - 2, bx0 = 4 * b0 - 2;
    const half_t* img = p.in + (long long)n * p.in_Hp * p.in_Wp * 4;
    for (int t = tid; t < 2 * IR * (IP / 2); t += NT) {
      const int pl = t >= IR * (IP / 2) ? 1 : 0, u = t - pl * IR * (IP / 2);
      const int r = u / (IP / 2), c = (u - r * (IP / 2)) * 2;
      const int by = by0 + r, bx = bx0 + c;
      half8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)0.f;
      if (by >= 0 && by < p.in_Hp && bx >= 0 && bx + 1 < p.in_Wp) v = *(const half8*)(img + pl * p.in_lo + ((long long)by * p.in_Wp + bx) * 4);
      *(half8*)(il + pl * IN_BYTES + (r * IP + c) * 8) = v;
    }
  }
  const int fi = lane & 15, fq = lane >> 4;
  int boff[BPW];
#pragma unroll
  for (int b = 0; b < BPW; ++b) {
    const int blk = wave + b * NW;
    int px = blk * 16 + fi;
    if (px >= NPX) px = NPX - 1;
    const int i = px / CC, j = px - i * CC;
    boff[b] = ((2 * i) * IP + 2 * j + 2 * fq) * 8;
  }
  __syncthreads();                                // the input patch is complete

#pragma unroll 1
  for (int hf = 0; hf < 2; ++hf) {
    f32x4 acc[BPW][2];
#pragma unroll
    for (int b = 0; b < BPW; ++b) { acc[b][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[b][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {        // W_hi.X_hi, W_hi.X_lo, W_lo.X_hi
      const half_t* wpl = p.wf + (long long)((pass == 2 ? 7 : 0) * 4 + 2 * hf) * 512 + lane * 8;      // fragment (plane, kh, block) = 512 halfs
      const char* xpl = il + (pass == 1 ? IN_BYTES : 0);
#pragma unroll
      for (int kh = 0; kh < 7; ++kh) {
        half8 wf[2], xf[BPW];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) wf[mb] = *(const half8*)(wpl + (kh * 4 + mb) * 512);
#pragma unroll
        for (int b = 0; b < BPW; ++b) xf[b] = *(const half8*)(xpl + boff[b] + kh * IP * 8);
#pragma unroll
        for (int b = 0; b < BPW; ++b)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) acc[b][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mb], xf[b], acc[b][mb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int c16 = (2 * hf + mb) * 16;         // the quarter's first channel
      __syncthreads();                            // the pooling of the previous quarter is done with the patch
      // ---- scale + bias + ReLU, split into (hi, lo) exactly as the stand-alone conv stores it; pixels outside the map are the pooling's zero padding
      const f32x4 sv = *(const f32x4*)(p.wscale + c16 + fq * 4), bv = *(const f32x4*)(p.bias + c16 + fq * 4);
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        const int blk = wave + b * NW;
        const int px = blk * 16 + fi;
        if (blk >= NBLK || px >= NPX) continue;
        const int i = px / CC, j = px - i * CC;
        const int cy = 2 * a0 - 1 + i, cx = 2 * b0 - 1 + j;
        const bool inside = cy >= 0 && cy < p.Hc && cx >= 0 && cx < p.Wc;
        u32x4 key;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float f = acc[b][mb][r] * sv[r] + bv[r];
          f = f > 0.f ? f : 0.f;
          f = f > 65504.f ? 65504.f : f;
          const half_t h = (half_t)f, l = (half_t)(f - (float)h);
          key[r] = inside ? pair_key(h, l) : pair_key((half_t)0.f, (half_t)0.f);
        }
        *(u32x4*)(ol + px * S_OP + fq * 16) = key;
      }
      __syncthreads();
      // ---- 3x3 stride-2 max of the VALUES hi + lo, the winning pair stored (maxpool3x3s2_split_kernel) -- as an unsigned max of the pairs' keys:
      // thread = (pooled pixel, 4 channels)
      {
        const int pp = tid >> 2, cg = tid & 3;
        const int a = pp / PC, b = pp - a * PC;
        const int py = a0 + a, pxo = b0 + b;
        if (py < p.Hq && pxo < p.Wq) {
          u32x4 m = *(const u32x4*)(ol + ((2 * a) * CC + 2 * b) * S_OP + cg * 16);
#pragma unroll
          for (int t = 1; t < 9; ++t) {
            const u32x4 v = *(const u32x4*)(ol + ((2 * a + t / 3) * CC + 2 * b + t % 3) * S_OP + cg * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
          }
          half4 mh, ml;
#pragma unroll
          for (int e = 0; e < 4; ++e) { half_t h, l; key_pair(m[e], h, l); mh[e] = h; ml[e] = l; }
          half_t* op = p.out + (((long long)n * (p.Hq + 2) + py + 1) * (p.Wq + 2) + pxo + 1) * 64 + c16 + cg * 4;
          *(half4*)op = mh;
          *(half4*)(op + p.out_lo) = ml;
        }
      }
    }
  }
}

}  // namespace

int launch_stem_pool_split(const StemPoolSplitParams& p, hipStream_t stream) {
  RS_CHECK(p.in && p.wf && p.wscale && p.bias && p.out && p.N >= 1, RS_ERR_ARG, "stem_pool_split: null argument");
  RS_CHECK((p.in_Wp & 1) == 0, RS_ERR_ARG, "stem_pool_split: odd row pitch %d", p.in_Wp);
  RS_CHECK(p.Hc == (p.in_Hp - 6 - 1) / 2 + 1 && p.Wc == (p.in_Wp - 6 - 1) / 2 + 1 && p.Hq == (p.Hc - 1) / 2 + 1 && p.Wq == (p.Wc - 1) / 2 + 1,
           RS_ERR_ARG, "stem_pool_split: geometry (input %d x %d with halo 3, conv %d x %d, pooled %d x %d)", p.in_Hp, p.in_Wp, p.Hc, p.Wc, p.Hq, p.Wq);
  static bool attr = false;
  if (!attr) {
    RS_HIP(hipFuncSetAttribute((const void*)stem_pool_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS));
    attr = true;
  }
  const dim3 grid(cdiv(p.Wq, PC), cdiv(p.Hq, PR), p.N);
  hipLaunchKernelGGL(stem_pool_split_kernel, grid, dim3(NT), S_LDS, stream, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_stem_pool(const StemPoolParams& p, hipStream_t stream) {
  RS_CHECK(p.in && p.wf && p.bias && p.out && p.N >= 1, RS_ERR_ARG, "stem_pool: null argument");
  RS_CHECK((p.in_Wp & 1) == 0, RS_ERR_ARG, "stem_pool: odd row pitch %d", p.in_Wp);
  RS_CHECK(p.Hc == (p.in_Hp - 6 - 1) / 2 + 1 && p.Wc == (p.in_Wp - 6 - 1) / 2 + 1 && p.Hq == (p.Hc - 1) / 2 + 1 && p.Wq == (p.Wc - 1) / 2 + 1,
           RS_ERR_ARG, "stem_pool: geometry (input %d x %d with halo 3, conv %d x %d, pooled %d x %d)", p.in_Hp, p.in_Wp, p.Hc, p.Wc, p.Hq, p.Wq);
  const dim3 grid(cdiv(p.Wq, PC), cdiv(p.Hq, PR), p.N);
#ifdef RS_STEM_DIAG      // ablation builds (results wrong by construction): RS_DEEP_DBG bit 0 no weight staging, 1 no input staging, 2 one K step, 3 no LDS stores of the conv outputs, 4 no pooling
  switch (rs_debug().deep_dbg) {
    case 1: hipLaunchKernelGGL(stem_pool_kernel<1>, grid, dim3(NT), 0, stream, p); return RS_OK;
    case 2: hipLaunchKernelGGL(stem_pool_kernel<2>, grid, dim3(NT), 0, stream, p); return RS_OK;
    case 3: hipLaunchKernelGGL(stem_pool_kernel<3>, grid, dim3(NT), 0, stream, p); return RS_OK;
    case 4: hipLaunchKernelGGL(stem_pool_kernel<4>, grid, dim3(NT), 0, stream, p); return RS_OK;
    case 8: hipLaunchKernelGGL(stem_pool_kernel<8>, grid, dim3(NT), 0, stream, p); return RS_OK;
    case 16: hipLaunchKernelGGL(stem_pool_kernel<16>, grid, dim3(NT), 0, stream, p); return RS_OK;
    case 31: hipLaunchKernelGGL(stem_pool_kernel<31>, grid, dim3(NT), 0, stream, p); return RS_OK;
    default: break;
  }
#endif
  hipLaunchKernelGGL(stem_pool_kernel<0>, grid, dim3(NT), 0, stream, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
