// Proposal / detection "glue" of the Mask R-CNN forward on gfx950: everything between the conv
// GEMMs.  All box and score arithmetic is fp32 in the same operation order as detectron2 0.6 /
// torchvision 0.11.3 (this file is compiled with -ffp-contract=off so no mul+add is fused), all
// index work is exact; capacities are fixed and counts stay on the device (no host sync).
//
//   rpn_select_kernel   RPN.predict_proposals / find_top_rpn_proposals, per (image, level):
//                       radix-select top-k logits (ties: lower anchor index first), bitonic sort,
//                       anchor-free decode (anchors recomputed from the index), clip, non-empty.
//                       [EXT d2: modeling/proposal_generator/{rpn,proposal_utils}.py,
//                        modeling/anchor_generator.py, modeling/box_regression.py]  R:40-56,222-251
//   nms_kernel          torchvision nms (IoU > thr suppresses), one workgroup per (image, level) or
//                       (image, class): 64-bit suppression bitmask in LDS + single-wave scan.
//                       [EXT tv: csrc/ops/cuda/nms_kernel.cu]
//   rpn_merge_kernel    batched_nms result order (score desc) + POST_NMS_TOPK.   R:247
//   roi_align_kernel    ROIPooler level assignment + ROIAlign(aligned=True, sampling_ratio=0)
//                       [EXT d2: modeling/poolers.py; EXT tv: csrc/ops/cuda/roi_align_kernel.cu] R:172-174,219-221
//   box_candidates_kernel / det_merge_kernel   fast_rcnn_inference_single_image + the box part of
//                       detector_postprocess  [EXT d2: modeling/roi_heads/fast_rcnn.py,
//                       modeling/postprocessing.py]  R:160-165,190,194,321
//   mask_predict_kernel mask predictor 1x1 on the predicted class + sigmoid (mask_rcnn_inference)
//   paste_masks_kernel  paste_masks_in_image (grid_sample bilinear, zeros, align_corners=False) >= thr,
//                       bit-packed output  [EXT d2: layers/mask_ops.py]
#include "detect.h"

namespace {

__device__ __forceinline__ uint32_t fkey(float f) {
  f = f + 0.0f;   // -0 -> +0 so that equal floats have equal keys
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// descending bitonic sort of n (power of two) 64-bit keys in LDS
template <int T>
__device__ void bitonic_sort_desc(unsigned long long* s, int n, int tid) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n; i += T) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = s[i], b = s[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) { s[i] = b; s[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
}

// smallest power of two >= c inside [lo, hi]: the bitonic sorts of the merge kernels run over the entries that exist (the rest of the list is zero
// keys, which a descending sort leaves where they are), not over the capacity -- 66 passes over 2 048 entries instead of 91 over 8 192 when a tile
// keeps 2 000 candidates; at batch 1 these one-workgroup kernels are a third of the forward's latency
__device__ __forceinline__ int sort_size(unsigned c, int lo, int hi) {
  int n = lo;
  while (n < hi && (unsigned)n < c) n <<= 1;
  return n;
}

// Box2BoxTransform.apply_deltas for one box, one delta quadruple (fp32, detectron2 op order).
__device__ __forceinline__ void apply_deltas(const float b[4], const float d[4], float wx, float wy, float ww, float wh,
                                             float scale_clamp, float out[4]) {
  const float widths = b[2] - b[0];
  const float heights = b[3] - b[1];
  const float ctr_x = b[0] + 0.5f * widths;
  const float ctr_y = b[1] + 0.5f * heights;
  const float dx = rs_fdiv(d[0], wx), dy = rs_fdiv(d[1], wy);
  float dw = rs_fdiv(d[2], ww), dh = rs_fdiv(d[3], wh);
  dw = dw > scale_clamp ? scale_clamp : dw;
  dh = dh > scale_clamp ? scale_clamp : dh;
  const float pcx = dx * widths + ctr_x;
  const float pcy = dy * heights + ctr_y;
  const float pw = expf(dw) * widths;
  const float ph = expf(dh) * heights;
  out[0] = pcx - 0.5f * pw;
  out[1] = pcy - 0.5f * ph;
  out[2] = pcx + 0.5f * pw;
  out[3] = pcy + 0.5f * ph;
}
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// ---------------------------------------------------------------------------------------------
// RPN: select + decode
// ---------------------------------------------------------------------------------------------
// CAP = candidate capacity per (image, level): 1024 (inference, PRE_NMS_TOPK_TEST 1000) or 2048 (training, PRE_NMS_TOPK_TRAIN 2000)
template <int CAP>
__global__ __launch_bounds__(1024) void rpn_select_kernel(const RpnParams p) {
  __shared__ unsigned long long list[CAP];
  __shared__ int hist[256];
  __shared__ int hist16[16 * 257];
  __shared__ unsigned int sh_prefix, sh_need, sh_cnt, sh_idx_thr, sh_ccount, sh_tie;
  const int l = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int H = p.H[l], W = p.W[l], A = p.A;
  const int HW = H * W;
  const int n_el = HW * A;
  const int k = n_el < p.topk ? n_el : p.topk;
  const float* head = p.head[l] + (long long)n * HW * p.cs;
  uint32_t* keys = p.keys[l] + (long long)n * 2 * n_el;     // [n_el] ordered keys, then [n_el] candidate indices
  uint32_t* cidx = keys + n_el;

  // Scan A: ordered keys to scratch + histogram of the top 8 bits.
  if (tid < 256) hist[tid] = 0;
  if (tid == 0) { sh_prefix = 0; sh_need = (unsigned)k; sh_cnt = 0; sh_ccount = 0; sh_idx_thr = 0xFFFFFFFFu; }
  for (int i = tid; i < CAP; i += 1024) list[i] = 0ull;
  __syncthreads();
  // Logits share sign and exponent, so the top digit hits a handful of bins: 16 privatised, bank-
  // staggered sub-histograms cut the same-address serialisation of the LDS atomics 16-fold.
  // (wave-aggregated ballot atomics were measured slower: 0.29 vs 0.24 ms)
  for (int i = tid; i < 16 * 257; i += 1024) hist16[i] = 0;
  __syncthreads();
  // 8 independent loads in flight per lane: with one workgroup per (image, level) the scan is bound by
  // load latency, not bandwidth.
  for (int base = 0; base < n_el; base += 8192) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = base + u * 1024 + tid;
      const int pix = e / A, a = e - pix * A;
      v[u] = e < n_el ? head[(long long)pix * p.cs + a] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = base + u * 1024 + tid;
      if (e < n_el) {
        const uint32_t key = fkey(v[u]);
        keys[e] = key;
        atomicAdd(&hist16[(tid & 15) * 257 + (key >> 24)], 1);
      }
    }
  }
  __syncthreads();
  if (tid < 256) {
    int sum = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) sum += hist16[q * 257 + tid];
    hist[tid] = sum;
  }
  __syncthreads();
  auto pick_desc = [&](int shift) {     // thread 0: bin holding the need-th largest, walking from the top
    unsigned need = sh_need, acc = 0;
    int b = 255;
    for (; b > 0; --b) {
      if (acc + (unsigned)hist[b] >= need) break;
      acc += (unsigned)hist[b];
    }
    sh_need = need - acc;              // still to take from bin b
    sh_prefix |= ((unsigned)b) << shift;
    sh_tie = (unsigned)hist[b];        // elements in bin b under the current prefix
  };
  if (p.debug == 1) return;
  if (tid == 0) pick_desc(24);
  __syncthreads();
  if (p.debug == 2) return;
  // Scan B: everything above the selected top-digit bin is in; the bin itself (typically ~10 % of the
  // anchors) is compacted to a candidate list, so the remaining digit passes touch only candidates.
  {
    const uint32_t b0 = sh_prefix >> 24;
    for (int base = 0; base < n_el; base += 8192) {
      uint32_t kv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + u * 1024 + tid;
        kv[u] = e < n_el ? keys[e] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + u * 1024 + tid;
        if (e >= n_el) continue;
        const uint32_t key = kv[u];
        const uint32_t top = key >> 24;
        if (top > b0) {
          const unsigned pos = atomicAdd(&sh_cnt, 1u);
          if (pos < (unsigned)CAP) list[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)e);
        } else if (top == b0) {
          const unsigned c = atomicAdd(&sh_ccount, 1u);
          cidx[c] = (uint32_t)e;
        }
      }
    }
  }
  __syncthreads();
  if (p.debug == 3) return;
  const int nc = (int)sh_ccount;
  for (int d = 1; d < 4; ++d) {
    const int shift = 24 - 8 * d;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = sh_prefix;
    for (int ci = tid; ci < nc; ci += 1024) {
      const uint32_t key = keys[cidx[ci]];
      if ((key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255], 1);
    }
    __syncthreads();
    if (tid == 0) pick_desc(shift);
    __syncthreads();
  }
  if (p.debug == 4) return;
  const uint32_t T = sh_prefix;       // k-th largest key
  // ties on the threshold key: take the lowest anchor indices (radix select on the index)
  if (sh_tie > sh_need) {              // uniform
    __syncthreads();
    if (tid == 0) sh_prefix = 0;
    __syncthreads();
    for (int d = 0; d < 3; ++d) {
      const int shift = 16 - 8 * d;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const uint32_t prefix = sh_prefix;
      for (int ci = tid; ci < nc; ci += 1024) {
        const uint32_t e = cidx[ci];
        if (keys[e] == T && (d == 0 || (e >> (shift + 8)) == (prefix >> (shift + 8))))
          atomicAdd(&hist[(e >> shift) & 255], 1);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned need = sh_need, acc = 0;
        int b = 0;
        for (; b < 255; ++b) {
          if (acc + (unsigned)hist[b] >= need) break;
          acc += (unsigned)hist[b];
        }
        sh_need = need - acc;
        sh_prefix |= ((unsigned)b) << shift;
      }
      __syncthreads();
    }
    if (tid == 0) sh_idx_thr = sh_prefix;   // largest index taken among the ties
    __syncthreads();
  }
  const uint32_t idx_thr = sh_idx_thr;
  for (int ci = tid; ci < nc; ci += 1024) {
    const uint32_t e = cidx[ci];
    const uint32_t key = keys[e];
    if (key > T || (key == T && e <= idx_thr)) {
      const unsigned pos = atomicAdd(&sh_cnt, 1u);
      if (pos < (unsigned)CAP) list[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - e);
    }
  }
  __syncthreads();
  if (p.debug == 5) return;
  bitonic_sort_desc<1024>(list, CAP, tid);
  if (p.debug == 6) return;

  const long long ob = ((long long)n * p.L + l) * CAP;
  if (tid == 0) p.cand_count[n * p.L + l] = k;
  for (int ti = tid; ti < k; ti += 1024) {
    const unsigned long long c = list[ti];
    const uint32_t e = 0xFFFFFFFFu - (uint32_t)(c & 0xFFFFFFFFull);
    const float score = fkey_inv((uint32_t)(c >> 32));
    const int pix = e / A, a = e - pix * A;
    const int y = pix / W, x = pix - y * W;
    const float sx = (float)(x * p.stride[l]) + p.offset * (float)p.stride[l];
    const float sy = (float)(y * p.stride[l]) + p.offset * (float)p.stride[l];
    float anc[4] = {sx + p.base[l][a][0], sy + p.base[l][a][1], sx + p.base[l][a][2], sy + p.base[l][a][3]};
    const float* dp = head + (long long)pix * p.cs + A + a * 4;
    float d[4] = {dp[0], dp[1], dp[2], dp[3]};
    float b[4];
    apply_deltas(anc, d, p.wx, p.wy, p.ww, p.wh, p.scale_clamp, b);
    const float ih = p.img_hw ? p.img_hw[2 * n] : p.img_h, iw = p.img_hw ? p.img_hw[2 * n + 1] : p.img_w;
    b[0] = clampf(b[0], 0.f, iw);
    b[1] = clampf(b[1], 0.f, ih);
    b[2] = clampf(b[2], 0.f, iw);
    b[3] = clampf(b[3], 0.f, ih);
    float* ob4 = p.cand_boxes + (ob + ti) * 4;
    ob4[0] = b[0]; ob4[1] = b[1]; ob4[2] = b[2]; ob4[3] = b[3];
    p.cand_scores[ob + ti] = score;
    p.cand_valid[ob + ti] = ((b[2] - b[0]) > p.min_size && (b[3] - b[1]) > p.min_size) ? 1 : 0;
    if (p.cand_index) p.cand_index[ob + ti] = (int)e;
  }
}

// ---------------------------------------------------------------------------------------------
// NMS over a segment of <= CAP boxes already in priority order.  CAP 1024: the suppression mask (128 KB) lives in LDS;
// CAP 2048 (training, PRE_NMS_TOPK_TRAIN 2000): 512 KB per segment in a global scratch buffer (L2-resident).
// ---------------------------------------------------------------------------------------------
// GM: the suppression mask lives in global memory (NmsParams::scratch) instead of LDS -- needed for CAP 2048, and used for CAP 1024 when few segments
// are launched (batch 1-3): the quadratic mask build is then shared by gridDim.y workgroups (mode 1) and the scan runs as a second launch (mode 2)
template <int CAP, bool GM>
__global__ __launch_bounds__(1024) void nms_kernel(const NmsParams p) {
  constexpr int WPR = CAP / 64;                                       // mask words per row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* sbox = (float4*)smem;                                       // CAP * 16
  float* sarea = (float*)(smem + CAP * 16);                           // CAP * 4
  unsigned long long* sremoved = (unsigned long long*)(smem + CAP * 20);   // WPR * 8
  const int s = blockIdx.x, tid = threadIdx.x;
  unsigned long long* mask = !GM ? (unsigned long long*)(smem + CAP * 20 + 256)
                                 : p.scratch + (long long)s * CAP * WPR;
  int n = p.count[s];
  if (n > CAP) n = CAP;
  const float* boxes = p.boxes + (long long)s * p.cap * 4;
  const uint8_t* valid = p.valid ? p.valid + (long long)s * p.cap : nullptr;
  uint8_t* keep = p.keep + (long long)s * p.cap;
  if (!(GM && p.mode == 1)) for (int i = tid; i < p.cap; i += 1024) keep[i] = 0;
  if (tid < WPR) sremoved[tid] = 0ull;
  __syncthreads();
  for (int ti = tid; ti < n; ti += 1024) {
    const float4 b = *(const float4*)(boxes + ti * 4);
    sbox[ti] = b;
    sarea[ti] = (b.z - b.x) * (b.w - b.y);
    if (valid && !valid[ti]) atomicOr(&sremoved[ti >> 6], 1ull << (ti & 63));
  }
  __syncthreads();
  const int nw = (n + 63) >> 6;
  // Row i needs words (i>>6) .. nw-1 only, a triangular amount of work.  Rows are paired (i, n-1-i)
  // so that every task slot (row pair, word slot) carries the same load and all 16 waves stay busy.
  const int nh = (n + 1) >> 1;
  const int nhp = (nh + 63) & ~63;
  const bool build = p.debug != 2 && !(GM && p.mode == 2);
  for (int idx = tid + blockIdx.y * 1024; idx < (nw + 2) * nhp && build; idx += 1024 * gridDim.y) {
    const int wq = idx / nhp, ip = idx - wq * nhp;
    if (ip >= nh) continue;
    const int first = nw - (ip >> 6);
    int i, w;
    if (wq < first) { i = ip; w = (ip >> 6) + wq; }
    else {
      i = n - 1 - ip;
      w = (i >> 6) + (wq - first);
      if (i == ip || w >= nw) continue;
    }
    unsigned long long bits = 0ull;
    const int j0 = w * 64;
    if (j0 + 63 > i) {
      const float4 a = sbox[i];
      const float sa = sarea[i];
      const float thr = p.thresh;
      // torchvision: suppress when fl(inter / (sa + sb - inter)) > thr.  The division is only needed when
      // inter is within 1e-5 (relative) of thr * union; outside that band the comparison of the products
      // decides identically (fp32 rounding is 6e-8), so the result stays bit-exact with the reference.
      auto test = [&](int j) -> bool {
        const float4 q = sbox[j];
        const float iw = fmaxf(fminf(a.z, q.z) - fmaxf(a.x, q.x), 0.f);
        const float ih = fmaxf(fminf(a.w, q.w) - fmaxf(a.y, q.y), 0.f);
        const float inter = iw * ih;
        if (!(inter > 0.f)) return false;           // 0/x = 0, 0/0 = NaN: never > thr
        const float uni = sa + sarea[j] - inter;
        const float tu = thr * uni;
        if (inter > tu * 1.00001f) return true;
        if (!(inter > tu * 0.99999f)) return false;
        return rs_fdiv(inter, uni) > thr;
      };
      unsigned lo = 0u, hi = 0u;
      if (j0 > i && j0 + 64 <= n) {                 // word entirely right of the diagonal and inside n
#pragma unroll 4
        for (int b = 0; b < 32; ++b) if (test(j0 + b)) lo |= 1u << b;
#pragma unroll 4
        for (int b = 0; b < 32; ++b) if (test(j0 + 32 + b)) hi |= 1u << b;
      } else {
        for (int b = 0; b < 32; ++b) { const int j = j0 + b; if (j > i && j < n && test(j)) lo |= 1u << b; }
        for (int b = 0; b < 32; ++b) { const int j = j0 + 32 + b; if (j > i && j < n && test(j)) hi |= 1u << b; }
      }
      bits = ((unsigned long long)hi << 32) | lo;
    }
    mask[(long long)i * WPR + w] = bits;
  }
  if (GM && p.mode == 1) return;                // mask build only: the scan is the next launch (mode 2)
  if (GM) __threadfence();       // mask rows in global memory: visible to the scanning wave after the barrier
  __syncthreads();
  // Greedy scan, one wave, 64 boxes (one mask word) per step: the intra-chunk part is resolved on
  // the scalar unit from the diagonal words (lane b holds row b), then the rows of the survivors
  // are OR-ed into the later words by all 64 lanes in parallel.
  if (tid < 64 && p.debug != 1) {
    const int lane = tid;
    unsigned long long removed = lane < WPR ? sremoved[lane] : 0ull;   // lane w holds word w
    for (int c = 0; c < nw; ++c) {
      const int base = c * 64;
      const int cnt = (n - base) < 64 ? (n - base) : 64;
      const unsigned long long rem_c = __shfl(removed, c);
      const unsigned long long diag = lane < cnt ? mask[(long long)(base + lane) * WPR + c] : 0ull;
      const int dlo = (int)(unsigned)(diag & 0xFFFFFFFFull), dhi = (int)(unsigned)(diag >> 32);
      unsigned long long alive = ~rem_c;
      if (cnt < 64) alive &= (1ull << cnt) - 1ull;
      unsigned alo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(alive & 0xFFFFFFFFull));
      unsigned ahi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(alive >> 32));
      unsigned long long al = ((unsigned long long)ahi << 32) | alo;
      unsigned long long kept = 0ull;
      while (al) {
        const int b = __builtin_ctzll(al);
        kept |= 1ull << b;
        const unsigned rlo = (unsigned)__builtin_amdgcn_readlane(dlo, b);
        const unsigned rhi = (unsigned)__builtin_amdgcn_readlane(dhi, b);
        al &= ~(((unsigned long long)rhi << 32) | rlo);
        al &= ~(1ull << b);
      }
      if (lane < cnt) keep[base + lane] = (uint8_t)((kept >> lane) & 1ull);
      const int w = lane & (WPR - 1);
      unsigned long long part = 0ull;
      if (w > c && w < nw) {
        for (int b = lane / WPR; b < cnt; b += 64 / WPR)
          if ((kept >> b) & 1ull) part |= mask[(long long)(base + b) * WPR + w];
      }
      if (WPR == 16) part |= __shfl_xor(part, 16);
      part |= __shfl_xor(part, 32);
      if (lane < WPR) removed |= part;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// RPN: merge levels (score desc, ties: lower (level, rank) first), keep post_topk
// ---------------------------------------------------------------------------------------------
template <int CAP>
__global__ __launch_bounds__(1024) void rpn_merge_kernel(const RpnMergeParams p) {
  constexpr int SORTN = CAP * 8;                          // up to 8 levels' candidates (power of two for the bitonic sort)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* list = (unsigned long long*)smem;   // SORTN
  __shared__ unsigned int cnt;
  const int n = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) cnt = 0;
  for (int i = tid; i < SORTN; i += 1024) list[i] = 0ull;
  __syncthreads();
  for (int l = 0; l < p.L; ++l) {
    const int c = p.cand_count[n * p.L + l];
    const long long ob = ((long long)n * p.L + l) * CAP;
    for (int i = tid; i < c; i += 1024) {
      if (p.keep[ob + i]) {
        const unsigned pos = atomicAdd(&cnt, 1u);
        list[pos] = ((unsigned long long)fkey(p.cand_scores[ob + i]) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)(l * CAP + i));
      }
    }
  }
  __syncthreads();
  bitonic_sort_desc<1024>(list, sort_size(cnt, 1024, SORTN), tid);
  const int total = (int)cnt < p.post_topk ? (int)cnt : p.post_topk;
  if (tid == 0) p.prop_count[n] = total;
  for (int i = tid; i < p.cap; i += 1024) {
    float* o = p.prop_boxes + ((long long)n * p.cap + i) * 4;
    if (i < total) {
      const uint32_t pos = 0xFFFFFFFFu - (uint32_t)(list[i] & 0xFFFFFFFFull);
      const long long src = (long long)n * p.L * CAP + pos;
      const float* b = p.cand_boxes + src * 4;
      o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3];
      p.prop_scores[(long long)n * p.cap + i] = p.cand_scores[src];
      if (p.prop_level) p.prop_level[(long long)n * p.cap + i] = (int)(pos / CAP);
    } else {
      o[0] = o[1] = o[2] = o[3] = 0.f;
      p.prop_scores[(long long)n * p.cap + i] = 0.f;
    }
  }
  if (!p.prop_order) return;                  // uniform
  // Visiting order of the RoI pooler: the proposals leave this kernel in score order, i.e. scattered over the image; sorted by
  // (pooler level, top row, left column) consecutive workgroups of box.roi_align read neighbouring rows of ONE feature map and hit
  // in L2 (measured HBM bytes of that kernel: 1.77x its algorithmic traffic in score order).  Pure scheduling: every RoI is
  // pooled into its own slot exactly as before.
  __syncthreads();                            // list[] (the score order) has been consumed
  for (int i = tid; i < 1024; i += 1024) {
    unsigned long long key = 0ull;            // descending sort: invalid slots (key 0) come last
    if (i < total && i < p.cap) {
      const float* b = p.prop_boxes + ((long long)n * p.cap + i) * 4;
      const float area = (b[2] - b[0]) * (b[3] - b[1]);
      const float v = rs_fdiv(sqrtf(area), 224.0f) + 1e-8f;
      const unsigned lvl = v >= 2.0f ? 3u : (v >= 1.0f ? 2u : (v >= 0.5f ? 1u : 0u));
      unsigned yq = (unsigned)fmaxf(b[1], 0.f), xq = (unsigned)fmaxf(b[0], 0.f);
      yq = yq > 8191u ? 8191u : yq; xq = xq > 8191u ? 8191u : xq;
      const unsigned k32 = (lvl << 26) | (yq << 13) | xq;                 // ascending in (level, y, x) ...
      key = ((unsigned long long)(0xFFFFFFFFu - k32) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i) ;   // ... under a descending sort
      key |= 1ull << 63;                      // valid slots before the zero keys (k32 < 2^28, so bit 63 of ~k32 is set anyway)
    }
    list[i] = key;
  }
  __syncthreads();
  bitonic_sort_desc<1024>(list, 1024, tid);
  for (int i = tid; i < p.cap && i < 1024; i += 1024) {
    const unsigned long long c = list[i];
    // invalid slots: any permutation of the remaining indices -- hand out total, total+1, ... in order
    p.prop_order[(long long)n * p.cap + i] = n * p.cap + (c ? (int)(0xFFFFFFFFu - (uint32_t)(c & 0xFFFFFFFFull)) : i);
  }
}

// ---------------------------------------------------------------------------------------------
// ROIAlign (aligned = true, adaptive sampling), C == 256, one workgroup per RoI.
// Phase 1: the P*gh row samples and P*gw column samples of the RoI are computed ONCE (address
// offset of the low/high neighbour + the two interpolation weights, zero weights for samples the
// reference skips) into LDS.  Phase 2: each half-wave owns one bin at a time; a lane carries 8
// consecutive channels (16-byte loads, 512 contiguous bytes per half-wave per neighbour), so the
// per-sample VALU work is 4 weight products + 32 multiply/adds instead of the full coordinate
// arithmetic.  Operation order of the accumulation follows torchvision's kernel exactly.
// ---------------------------------------------------------------------------------------------
#define RS_ROI_MAXS 512
__global__ __launch_bounds__(256) void roi_align_kernel(const RoiAlignParams p) {
  __shared__ int s_lo[2][RS_ROI_MAXS], s_hi[2][RS_ROI_MAXS];     // [0] = y (row offsets), [1] = x (column offsets), in elements
  __shared__ float s_l[2][RS_ROI_MAXS], s_h[2][RS_ROI_MAXS];
  const int entry = blockIdx.x;
  const int tid = threadIdx.x;
  int n_entries = p.S;
  if (p.n_entries) { const int c = *p.n_entries; n_entries = c < n_entries ? c : n_entries; }
  if (entry >= n_entries) return;
  const int slot = p.slot_list ? p.slot_list[entry] : entry;
  const int n = slot / p.slots_per_image;
  const int P = p.P, PP = P + 2 * p.out_pad;
  const int es = p.f32 == 1 ? 4 : 2;            // element size of features / output (p.f32 == 2: split-operand mode, two fp16 planes)
  char* out = (char*)p.out + (long long)entry * PP * PP * 256 * es;
  const int hw = tid >> 5, l32 = tid & 31;       // half-wave id, lane inside it (8 channels each)
  bool valid = true;
  if (p.per_image_count) valid = (slot - n * p.slots_per_image) < p.per_image_count[n];
  if (!valid) {
    for (int b = hw; b < P * P; b += 8) {
      const int ph = b / P, pw = b - ph * P;
      char* o = out + (((long long)(ph + p.out_pad) * PP + pw + p.out_pad) * 256 + l32 * 8) * es;
      if (p.f32 == 1) { ((f32x4*)o)[0] = f32x4{0.f, 0.f, 0.f, 0.f}; ((f32x4*)o)[1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      else { half8 z; for (int i = 0; i < 8; ++i) z[i] = (half_t)0.f; *(half8*)o = z; if (p.f32 == 2) *(half8*)(o + p.out_lo * 2) = z; }
    }
    return;
  }
  const float* r = p.rois + (long long)slot * 4;
  const float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
  // assign_boxes_to_levels: floor(4 + log2(sqrt(area)/224 + 1e-8)) clamped to [2,5]; evaluated with
  // exact power-of-two thresholds instead of log2 (identical wherever log2 is exact at 2^k).
  const float area = (x2 - x1) * (y2 - y1);
  const float v = rs_fdiv(sqrtf(area), 224.0f) + 1e-8f;
  int lvl = v >= 2.0f ? 3 : (v >= 1.0f ? 2 : (v >= 0.5f ? 1 : 0));
  if (lvl > p.nlevels - 1) lvl = p.nlevels - 1;
  if (p.out_level) { if (tid == 0) p.out_level[entry] = lvl; }
  const int H = p.H[lvl], W = p.W[lvl];
  const float sc = p.scale[lvl];
  const char* feat = (const char*)p.feat[lvl] + ((long long)n * (H + 2) * (W + 2) * 256 + l32 * 8) * es;
  const float roi_start_w = x1 * sc - 0.5f;
  const float roi_start_h = y1 * sc - 0.5f;
  const float roi_end_w = x2 * sc - 0.5f;
  const float roi_end_h = y2 * sc - 0.5f;
  const float roi_w = roi_end_w - roi_start_w;
  const float roi_h = roi_end_h - roi_start_h;
  const float bin_h = rs_fdiv(roi_h, (float)P);
  const float bin_w = rs_fdiv(roi_w, (float)P);
  int gh = (int)ceilf(rs_fdiv(roi_h, (float)P));
  int gw = (int)ceilf(rs_fdiv(roi_w, (float)P));
  if (gh < 0) gh = 0;
  if (gw < 0) gw = 0;
  const float count = (float)((gh * gw) > 1 ? (gh * gw) : 1);
  const bool fast = (P * gh <= RS_ROI_MAXS) && (P * gw <= RS_ROI_MAXS);

  // one sample coordinate -> (low offset, high offset, l, h); out-of-range samples get zero weights
  auto prep = [&](float c, int size, int pitch, int& lo, int& hi, float& l, float& h) {
    const bool oob = (c < -1.0f || c > (float)size);
    if (c <= 0.f) c = 0.f;
    int c_low = (int)c, c_high;
    if (c_low >= size - 1) { c_high = c_low = size - 1; c = (float)c_low; } else { c_high = c_low + 1; }
    l = c - (float)c_low;
    h = 1.f - l;
    if (oob) { l = 0.f; h = 0.f; c_low = 0; c_high = 0; }
    lo = (c_low + 1) * pitch;
    hi = (c_high + 1) * pitch;
  };
  if (fast) {
    for (int t = tid; t < P * gh; t += 256) {
      const int ph = t / gh, iy = t - ph * gh;
      const float y = roi_start_h + (float)ph * bin_h + rs_fdiv(((float)iy + 0.5f) * bin_h, (float)gh);
      prep(y, H, (W + 2) * 256, s_lo[0][t], s_hi[0][t], s_l[0][t], s_h[0][t]);
    }
    for (int t = tid; t < P * gw; t += 256) {
      const int pw = t / gw, ix = t - pw * gw;
      const float x = roi_start_w + (float)pw * bin_w + rs_fdiv(((float)ix + 0.5f) * bin_w, (float)gw);
      prep(x, W, 256, s_lo[1][t], s_hi[1][t], s_l[1][t], s_h[1][t]);
    }
  }
  __syncthreads();
  for (int b0 = 0; b0 < P * P; b0 += 8) {
    const int b = b0 + hw;
    const bool live = b < P * P;
    const int bb = live ? b : P * P - 1;
    const int ph = bb / P, pw = bb - ph * P;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (int iy = 0; iy < gh; ++iy) {
      int ylo, yhi; float ly, hy;
      if (fast) { const int t = ph * gh + iy; ylo = s_lo[0][t]; yhi = s_hi[0][t]; ly = s_l[0][t]; hy = s_h[0][t]; }
      else {
        const float y = roi_start_h + (float)ph * bin_h + rs_fdiv(((float)iy + 0.5f) * bin_h, (float)gh);
        prep(y, H, (W + 2) * 256, ylo, yhi, ly, hy);
      }
      for (int ix = 0; ix < gw; ++ix) {
        int xlo, xhi; float lx, hx;
        if (fast) { const int t = pw * gw + ix; xlo = s_lo[1][t]; xhi = s_hi[1][t]; lx = s_l[1][t]; hx = s_h[1][t]; }
        else {
          const float x = roi_start_w + (float)pw * bin_w + rs_fdiv(((float)ix + 0.5f) * bin_w, (float)gw);
          prep(x, W, 256, xlo, xhi, lx, hx);
        }
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        float f1[8], f2[8], f3[8], f4[8];
        if (p.f32 == 2) {
          const long long lo2 = p.feat_lo[lvl] * 2;
          const half8 v1 = *(const half8*)(feat + (long long)(ylo + xlo) * 2), u1 = *(const half8*)(feat + lo2 + (long long)(ylo + xlo) * 2);
          const half8 v2 = *(const half8*)(feat + (long long)(ylo + xhi) * 2), u2 = *(const half8*)(feat + lo2 + (long long)(ylo + xhi) * 2);
          const half8 v3 = *(const half8*)(feat + (long long)(yhi + xlo) * 2), u3 = *(const half8*)(feat + lo2 + (long long)(yhi + xlo) * 2);
          const half8 v4 = *(const half8*)(feat + (long long)(yhi + xhi) * 2), u4 = *(const half8*)(feat + lo2 + (long long)(yhi + xhi) * 2);
#pragma unroll
          for (int c = 0; c < 8; ++c) {      // fp32(hi) + fp32(lo) is exact
            f1[c] = (float)v1[c] + (float)u1[c]; f2[c] = (float)v2[c] + (float)u2[c];
            f3[c] = (float)v3[c] + (float)u3[c]; f4[c] = (float)v4[c] + (float)u4[c];
          }
        } else if (p.f32) {
          const f32x4* q1 = (const f32x4*)(feat + (long long)(ylo + xlo) * 4);
          const f32x4* q2 = (const f32x4*)(feat + (long long)(ylo + xhi) * 4);
          const f32x4* q3 = (const f32x4*)(feat + (long long)(yhi + xlo) * 4);
          const f32x4* q4 = (const f32x4*)(feat + (long long)(yhi + xhi) * 4);
#pragma unroll
          for (int c = 0; c < 8; ++c) { f1[c] = q1[c >> 2][c & 3]; f2[c] = q2[c >> 2][c & 3]; f3[c] = q3[c >> 2][c & 3]; f4[c] = q4[c >> 2][c & 3]; }
        } else {
          const half8 v1 = *(const half8*)(feat + (long long)(ylo + xlo) * 2);
          const half8 v2 = *(const half8*)(feat + (long long)(ylo + xhi) * 2);
          const half8 v3 = *(const half8*)(feat + (long long)(yhi + xlo) * 2);
          const half8 v4 = *(const half8*)(feat + (long long)(yhi + xhi) * 2);
#pragma unroll
          for (int c = 0; c < 8; ++c) { f1[c] = (float)v1[c]; f2[c] = (float)v2[c]; f3[c] = (float)v3[c]; f4[c] = (float)v4[c]; }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float val = w1 * f1[c] + w2 * f2[c] + w3 * f3[c] + w4 * f4[c];
          acc[c] += val;
        }
      }
    }
    if (live) {
      char* op = out + (((long long)(ph + p.out_pad) * PP + pw + p.out_pad) * 256 + l32 * 8) * es;
      if (p.f32 == 1) {
        ((f32x4*)op)[0] = f32x4{rs_fdiv(acc[0], count), rs_fdiv(acc[1], count), rs_fdiv(acc[2], count), rs_fdiv(acc[3], count)};
        ((f32x4*)op)[1] = f32x4{rs_fdiv(acc[4], count), rs_fdiv(acc[5], count), rs_fdiv(acc[6], count), rs_fdiv(acc[7], count)};
      } else {
        half8 o, ol;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float f = rs_fdiv(acc[c], count);
          o[c] = (half_t)f;
          ol[c] = (half_t)(f - (float)o[c]);
        }
        *(half8*)op = o;
        if (p.f32 == 2) *(half8*)(op + p.out_lo * 2) = ol;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ROIAlign, production (fp16) form.  The average over the gh x gw bilinear samples of a bin is separable:
//   out[ph][pw][c] = 1/count * sum_y sum_x  wy[ph][y] * wx[pw][x] * F[y][x][c],
// where wy[ph][y] is the sum over the bin's gh row samples of the sample's weight on feature row y (h on
// y_low, l on y_high; nothing for the samples torchvision skips) and wx likewise.  So each bin reads every cell
// of its (<= gh+2) x (<= gw+2) window ONCE instead of 4 corner cells per sample (16 -> 9 reads at g = 2,
// 64 -> 25 at g = 4): the kernel is bound by L1/L2 request rate, not HBM.  Same half-wave-per-bin, 8 channels
// per lane layout as roi_align_kernel; fp32 accumulation; the summation order differs from torchvision's
// (weights are pre-summed), which the fp32 validation mode avoids by using roi_align_kernel.
// RoIs whose window exceeds the LDS table fall back to per-sample evaluation inside this kernel.
// ---------------------------------------------------------------------------------------------
#define RS_ROI_WMAX 24   // window rows/cols per bin held in LDS (g <= 22)
#define RS_ROI_PMAX 14
// SPLIT: the split-operand precision mode (p.f32 == 2): features and output are hi / lo fp16 planes; a cell is fp32(hi) + fp32(lo) (exact).
template <bool SPLIT>
__global__ __launch_bounds__(256, SPLIT ? 4 : 6) void roi_align_win_kernel(const RoiAlignParams p) {
  __shared__ float s_w[2][RS_ROI_PMAX][RS_ROI_WMAX];   // [0] = wy[ph][j], [1] = wx[pw][i]
  __shared__ int s_base[2][RS_ROI_PMAX], s_len[2][RS_ROI_PMAX];
  int entry = blockIdx.x;
  if (p.order) {            // XCD k (workgroups k, k+8, ...) walks the k-th eighth of the visiting order
    const int q8 = p.S >> 3, r8 = p.S & 7, x8 = entry & 7;
    entry = p.order[(x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (entry >> 3)];
  }
  const int tid = threadIdx.x;
  int n_entries = p.S;
  if (p.n_entries) { const int c = *p.n_entries; n_entries = c < n_entries ? c : n_entries; }
  if (entry >= n_entries) return;
  const int slot = p.slot_list ? p.slot_list[entry] : entry;
  const int n = slot / p.slots_per_image;
  const int P = p.P, PP = P + 2 * p.out_pad;
  half_t* out = p.out + (long long)entry * PP * PP * 256;
  const int hw = tid >> 5, l32 = tid & 31;
  bool valid = true;
  if (p.per_image_count) valid = (slot - n * p.slots_per_image) < p.per_image_count[n];
  if (!valid) {
    half8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (half_t)0.f;
    for (int b = hw; b < P * P; b += 8) {
      const int ph = b / P, pw = b - ph * P;
      *(half8*)(out + ((long long)(ph + p.out_pad) * PP + pw + p.out_pad) * 256 + l32 * 8) = z;
      if constexpr (SPLIT) *(half8*)(out + p.out_lo + ((long long)(ph + p.out_pad) * PP + pw + p.out_pad) * 256 + l32 * 8) = z;
    }
    return;
  }
  const float* r = p.rois + (long long)slot * 4;
  const float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
  const float area = (x2 - x1) * (y2 - y1);
  const float v = rs_fdiv(sqrtf(area), 224.0f) + 1e-8f;
  int lvl = v >= 2.0f ? 3 : (v >= 1.0f ? 2 : (v >= 0.5f ? 1 : 0));
  if (lvl > p.nlevels - 1) lvl = p.nlevels - 1;
  if (p.out_level) { if (tid == 0) p.out_level[entry] = lvl; }
  const int H = p.H[lvl], W = p.W[lvl];
  const float sc = p.scale[lvl];
  const half_t* feat = p.feat[lvl] + ((long long)n * (H + 2) * (W + 2) + (W + 2) + 1) * 256 + l32 * 8;   // cell (0,0)
  const long long flo = SPLIT ? p.feat_lo[lvl] : 0;
  // a cell's 8 channels as floats
  auto cell = [&](const half_t* q, float (&f)[8]) {
    const half8 v = *(const half8*)q;
    if constexpr (SPLIT) {
      const half8 u = *(const half8*)(q + flo);
#pragma unroll
      for (int c = 0; c < 8; ++c) f[c] = (float)v[c] + (float)u[c];
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) f[c] = (float)v[c];
    }
  };
  const float roi_start_w = x1 * sc - 0.5f;
  const float roi_start_h = y1 * sc - 0.5f;
  const float roi_w = (x2 * sc - 0.5f) - roi_start_w;
  const float roi_h = (y2 * sc - 0.5f) - roi_start_h;
  const float bin_h = rs_fdiv(roi_h, (float)P);
  const float bin_w = rs_fdiv(roi_w, (float)P);
  int gh = (int)ceilf(rs_fdiv(roi_h, (float)P));
  int gw = (int)ceilf(rs_fdiv(roi_w, (float)P));
  if (gh < 0) gh = 0;
  if (gw < 0) gw = 0;
  const float count = (float)((gh * gw) > 1 ? (gh * gw) : 1);

  // threads 0..P-1 build the row tables, 32..32+P-1 the column tables (serial over the g samples of the bin, in
  // sample order, so the pre-summed weights are deterministic)
  if ((tid < P) || (tid >= 32 && tid < 32 + P)) {
    const int ax = tid >= 32 ? 1 : 0;
    const int b = ax ? tid - 32 : tid;
    const int g = ax ? gw : gh;
    const int size = ax ? W : H;
    const float start = ax ? roi_start_w : roi_start_h;
    const float bin = ax ? bin_w : bin_h;
    float* w = s_w[ax][b];
    for (int j = 0; j < RS_ROI_WMAX; ++j) w[j] = 0.f;
    int base = 0, len = 0;
    bool have = false, overflow = false;
    for (int i = 0; i < g; ++i) {
      float c = start + (float)b * bin + rs_fdiv(((float)i + 0.5f) * bin, (float)g);
      if (c < -1.0f || c > (float)size) continue;          // torchvision: sample contributes 0
      if (c <= 0.f) c = 0.f;
      int lo = (int)c, hi;
      if (lo >= size - 1) { hi = lo = size - 1; c = (float)lo; } else { hi = lo + 1; }
      const float l = c - (float)lo, h = 1.f - l;
      if (!have) { base = lo; have = true; }
      if (hi - base >= RS_ROI_WMAX) { overflow = true; break; }
      w[lo - base] += h;
      w[hi - base] += l;
      len = hi - base + 1;
    }
    s_base[ax][b] = base;
    s_len[ax][b] = overflow ? -1 : len;
  }
  __syncthreads();
  for (int b0 = 0; b0 < P * P; b0 += 8) {
    const int b = b0 + hw;
    if (b >= P * P) break;                                  // uniform per half-wave; no barrier below
    const int ph = b / P, pw = b - ph * P;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    const int ny = s_len[0][ph], nx = s_len[1][pw];
    if (ny >= 0 && nx >= 0) {
      const half_t* f0 = feat + ((long long)s_base[0][ph] * (W + 2) + s_base[1][pw]) * 256;
      const float* wy = s_w[0][ph];
      const float* wx = s_w[1][pw];
      for (int j = 0; j < ny; ++j) {
        const float wj = wy[j];
        const half_t* fr = f0 + (long long)j * (W + 2) * 256;
        int i = 0;
        for (; i + 2 <= nx; i += 2) {
          float v0[8], v1[8];
          cell(fr + i * 256, v0);
          cell(fr + (i + 1) * 256, v1);
          const float w0 = wj * wx[i], w1 = wj * wx[i + 1];
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[c] += w0 * v0[c];
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[c] += w1 * v1[c];
        }
        if (i < nx) {
          float v0[8];
          cell(fr + i * 256, v0);
          const float w0 = wj * wx[i];
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[c] += w0 * v0[c];
        }
      }
    } else {
      // window larger than the table (very elongated RoI): per-sample evaluation, torchvision's order
      for (int iy = 0; iy < gh; ++iy) {
        float y = roi_start_h + (float)ph * bin_h + rs_fdiv(((float)iy + 0.5f) * bin_h, (float)gh);
        if (y < -1.0f || y > (float)H) continue;
        if (y <= 0.f) y = 0.f;
        int ylo = (int)y, yhi;
        if (ylo >= H - 1) { yhi = ylo = H - 1; y = (float)ylo; } else { yhi = ylo + 1; }
        const float ly = y - (float)ylo, hy = 1.f - ly;
        for (int ix = 0; ix < gw; ++ix) {
          float x = roi_start_w + (float)pw * bin_w + rs_fdiv(((float)ix + 0.5f) * bin_w, (float)gw);
          if (x < -1.0f || x > (float)W) continue;
          if (x <= 0.f) x = 0.f;
          int xlo = (int)x, xhi;
          if (xlo >= W - 1) { xhi = xlo = W - 1; x = (float)xlo; } else { xhi = xlo + 1; }
          const float lx = x - (float)xlo, hx = 1.f - lx;
          float v1[8], v2[8], v3[8], v4[8];
          cell(feat + ((long long)ylo * (W + 2) + xlo) * 256, v1);
          cell(feat + ((long long)ylo * (W + 2) + xhi) * 256, v2);
          cell(feat + ((long long)yhi * (W + 2) + xlo) * 256, v3);
          cell(feat + ((long long)yhi * (W + 2) + xhi) * 256, v4);
          const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[c] += w1 * v1[c] + w2 * v2[c] + w3 * v3[c] + w4 * v4[c];
        }
      }
    }
    half8 o, ol;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float f = rs_fdiv(acc[c], count);
      o[c] = (half_t)f;
      if constexpr (SPLIT) ol[c] = (half_t)(f - (float)o[c]);
    }
    *(half8*)(out + ((long long)(ph + p.out_pad) * PP + pw + p.out_pad) * 256 + l32 * 8) = o;
    if constexpr (SPLIT) *(half8*)(out + p.out_lo + ((long long)(ph + p.out_pad) * PP + pw + p.out_pad) * 256 + l32 * 8) = ol;
  }
}

// ---------------------------------------------------------------------------------------------
// ROIAlign backward (training path): the adjoint of roi_align_win_kernel.  For every bin the
// incoming gradient g[ph][pw][c]/count is spread over the bin's cell window with the same separable weights,
//   dF[y][x][c] += wy[ph][y] * wx[pw][x] * g[ph][pw][c] / count,
// with float atomics into the fp32 gradient maps (as torchvision's roi_align_backward_kernel does with atomicAdd:
// [EXT tv: csrc/ops/cuda/roi_align_kernel.cu]; summation order, hence the last bits, vary from run to run there too).
// Same level assignment and per-bin tables as the forward kernel; the work is laid out per CELL of the RoI's window
// (gather over the bins that reach the cell, then one atomic per channel), see below.
// ---------------------------------------------------------------------------------------------
#define RS_ROI_CELLS 320   // rows/cols of a whole RoI window the gather form handles (14 bins x 22 samples + 2 at most)
template <typename G>     // G: storage type of the incoming gradient (half_t, or float in the reference-precision trainer)
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const RoiAlignParams p) {
  __shared__ float s_w[2][RS_ROI_PMAX][RS_ROI_WMAX];
  __shared__ int s_base[2][RS_ROI_PMAX], s_len[2][RS_ROI_PMAX];
  __shared__ short s_lo[2][RS_ROI_CELLS], s_hi[2][RS_ROI_CELLS];
  __shared__ int s_org[2], s_ext[2];
  const int entry = blockIdx.x;
  const int tid = threadIdx.x;
  if (p.bwd_overflow && *p.bwd_overflow == 0) return;      // every entry was handled by roi_bwd_gather_kernel (the usual case)
  int n_entries = p.S;
  if (p.n_entries) { const int c = *p.n_entries; n_entries = c < n_entries ? c : n_entries; }
  if (entry >= n_entries) return;
  const int slot = p.slot_list ? p.slot_list[entry] : entry;
  const int n = slot / p.slots_per_image;
  const int P = p.P, PP = P + 2 * p.out_pad;
  const G* gout = (const G*)p.out + (long long)entry * PP * PP * 256;
  const int l32 = tid & 31;
  if (p.per_image_count && (slot - n * p.slots_per_image) >= p.per_image_count[n]) return;
  const float* r = p.rois + (long long)slot * 4;
  const float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
  const float area = (x2 - x1) * (y2 - y1);
  const float v = rs_fdiv(sqrtf(area), 224.0f) + 1e-8f;
  int lvl = v >= 2.0f ? 3 : (v >= 1.0f ? 2 : (v >= 0.5f ? 1 : 0));
  if (lvl > p.nlevels - 1) lvl = p.nlevels - 1;
  const int H = p.H[lvl], W = p.W[lvl];
  const float sc = p.scale[lvl];
  float* dfeat = p.dfeat[lvl] + ((long long)n * (H + 2) * (W + 2) + (W + 2) + 1) * 256 + l32 * 8;   // cell (0,0)
  const float roi_start_w = x1 * sc - 0.5f;
  const float roi_start_h = y1 * sc - 0.5f;
  const float roi_w = (x2 * sc - 0.5f) - roi_start_w;
  const float roi_h = (y2 * sc - 0.5f) - roi_start_h;
  const float bin_h = rs_fdiv(roi_h, (float)P);
  const float bin_w = rs_fdiv(roi_w, (float)P);
  int gh = (int)ceilf(rs_fdiv(roi_h, (float)P));
  int gw = (int)ceilf(rs_fdiv(roi_w, (float)P));
  if (gh < 0) gh = 0;
  if (gw < 0) gw = 0;
  const float count = (float)((gh * gw) > 1 ? (gh * gw) : 1);
  if ((tid < P) || (tid >= 32 && tid < 32 + P)) {
    const int ax = tid >= 32 ? 1 : 0;
    const int b = ax ? tid - 32 : tid;
    const int g = ax ? gw : gh;
    const int size = ax ? W : H;
    const float start = ax ? roi_start_w : roi_start_h;
    const float bin = ax ? bin_w : bin_h;
    float* w = s_w[ax][b];
    for (int j = 0; j < RS_ROI_WMAX; ++j) w[j] = 0.f;
    int base = 0, len = 0;
    bool have = false, overflow = false;
    for (int i = 0; i < g; ++i) {
      float c = start + (float)b * bin + rs_fdiv(((float)i + 0.5f) * bin, (float)g);
      if (c < -1.0f || c > (float)size) continue;
      if (c <= 0.f) c = 0.f;
      int lo = (int)c, hi;
      if (lo >= size - 1) { hi = lo = size - 1; c = (float)lo; } else { hi = lo + 1; }
      const float l = c - (float)lo, h = 1.f - l;
      if (!have) { base = lo; have = true; }
      if (hi - base >= RS_ROI_WMAX) { overflow = true; break; }
      w[lo - base] += h;
      w[hi - base] += l;
      len = hi - base + 1;
    }
    s_base[ax][b] = base;
    s_len[ax][b] = overflow ? -1 : len;
  }
  __syncthreads();
  // One WAVE per bin, lane l owns channels l, l+64, l+128, l+192: every atomic instruction of the wave then covers 64
  // CONSECUTIVE floats (four full 64-byte lines) of one cell, which the L2 atomic units take as four line updates --
  // with 8 consecutive channels per lane (the forward layout) the same instruction scatters 64 words over 2 KB and the
  // kernel is ~6x slower (measured: 25.7 -> see DESIGN.md ms for 8 x 1024 RoIs).
  const int wv = tid >> 6, ln = tid & 63;
  float* dfl = dfeat - l32 * 8 + ln;              // undo the forward-style channel offset baked into dfeat

  // ---- gather form: neighbouring bins' windows overlap (by 1-2 cells on each side), so per-bin scattering issues
  // 1.5x (P=7, g=3) to 3.3x (P=14, g=2) more atomics than the RoI has cells.  Instead one wave per CELL of the RoI's whole
  // window sums the few bins that reach it (gradient tile read through L1/L2) and issues ONE atomic per channel.
  // s_lo/s_hi: per window row / column the range of bins that may cover it.
  if (tid == 0 || tid == 32) {
    const int ax = tid >> 5;
    int org = 0x7fffffff, end = -1, bad = 0;
    for (int b = 0; b < P; ++b) {
      const int len = s_len[ax][b];
      if (len < 0) { bad = 1; break; }
      if (len == 0) continue;
      org = min(org, s_base[ax][b]);
      end = max(end, s_base[ax][b] + len);
    }
    if (end < 0) { org = 0; end = 0; }
    if (end - org > RS_ROI_CELLS) bad = 1;
    s_org[ax] = org;
    s_ext[ax] = bad ? -1 : end - org;
  }
  __syncthreads();
  const int eh = s_ext[0], ew = s_ext[1];
  if (eh >= 0 && ew >= 0) {
    if (p.bwd_overflow) return;                              // ... as was this one
    for (int t = tid; t < eh + ew; t += 256) {
      const int ax = t >= eh ? 1 : 0;
      const int rel = ax ? t - eh : t;
      const int y = s_org[ax] + rel;
      int lo = P, hi = 0;
      for (int b = 0; b < P; ++b) {
        const int j = y - s_base[ax][b];
        if (j >= 0 && j < s_len[ax][b] && s_w[ax][b][j] != 0.f) { lo = min(lo, b); hi = max(hi, b + 1); }
      }
      s_lo[ax][rel] = (short)lo;
      s_hi[ax][rel] = (short)hi;
    }
    __syncthreads();
    const int oy = s_org[0], ox = s_org[1];
    for (int cell = wv; cell < eh * ew; cell += 4) {           // uniform per wave; no barrier below
      const int ty = cell / ew, tx = cell - ty * ew;
      const int plo = s_lo[0][ty], phi = s_hi[0][ty], qlo = s_lo[1][tx], qhi = s_hi[1][tx];
      if (plo >= phi || qlo >= qhi) continue;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int ph = plo; ph < phi; ++ph) {
        const int jy = oy + ty - s_base[0][ph];
        if (jy < 0 || jy >= s_len[0][ph]) continue;
        const float wy = s_w[0][ph][jy];
        if (wy == 0.f) continue;
        for (int pw = qlo; pw < qhi; ++pw) {
          const int jx = ox + tx - s_base[1][pw];
          if (jx < 0 || jx >= s_len[1][pw]) continue;
          const float wgt = wy * s_w[1][pw][jx];
          if (wgt == 0.f) continue;
          const G* gp = gout + ((long long)(ph + p.out_pad) * PP + pw + p.out_pad) * 256 + ln;
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] += wgt * rs_fdiv((float)gp[c * 64], count);
        }
      }
      float* d = dfl + ((long long)(oy + ty) * (W + 2) + ox + tx) * 256;
#pragma unroll
      for (int c = 0; c < 4; ++c) atomicAdd(d + c * 64, acc[c]);
    }
    return;
  }

  // ---- window larger than the tables (very elongated RoI): per-bin scatter
  for (int b0 = 0; b0 < P * P; b0 += 4) {
    const int b = b0 + wv;
    if (b >= P * P) break;                        // uniform per wave; no barrier below
    const int ph = b / P, pw = b - ph * P;
    const G* gp = gout + ((long long)(ph + p.out_pad) * PP + pw + p.out_pad) * 256 + ln;
    float gsc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) gsc[c] = rs_fdiv((float)gp[c * 64], count);
    const int ny = s_len[0][ph], nx = s_len[1][pw];
    if (ny >= 0 && nx >= 0) {
      float* f0 = dfl + ((long long)s_base[0][ph] * (W + 2) + s_base[1][pw]) * 256;
      for (int j = 0; j < ny; ++j) {
        const float wj = s_w[0][ph][j];
        if (wj == 0.f) continue;
        for (int i = 0; i < nx; ++i) {
          const float wgt = wj * s_w[1][pw][i];
          if (wgt == 0.f) continue;
          float* d = f0 + ((long long)j * (W + 2) + i) * 256;
#pragma unroll
          for (int c = 0; c < 4; ++c) atomicAdd(d + c * 64, wgt * gsc[c]);
        }
      }
    } else {
      for (int iy = 0; iy < gh; ++iy) {
        float y = roi_start_h + (float)ph * bin_h + rs_fdiv(((float)iy + 0.5f) * bin_h, (float)gh);
        if (y < -1.0f || y > (float)H) continue;
        if (y <= 0.f) y = 0.f;
        int ylo = (int)y, yhi;
        if (ylo >= H - 1) { yhi = ylo = H - 1; y = (float)ylo; } else { yhi = ylo + 1; }
        const float ly = y - (float)ylo, hy = 1.f - ly;
        for (int ix = 0; ix < gw; ++ix) {
          float x = roi_start_w + (float)pw * bin_w + rs_fdiv(((float)ix + 0.5f) * bin_w, (float)gw);
          if (x < -1.0f || x > (float)W) continue;
          if (x <= 0.f) x = 0.f;
          int xlo = (int)x, xhi;
          if (xlo >= W - 1) { xhi = xlo = W - 1; x = (float)xlo; } else { xhi = xlo + 1; }
          const float lx = x - (float)xlo, hx = 1.f - lx;
          float* d1 = dfl + ((long long)ylo * (W + 2) + xlo) * 256;
          float* d2 = dfl + ((long long)ylo * (W + 2) + xhi) * 256;
          float* d3 = dfl + ((long long)yhi * (W + 2) + xlo) * 256;
          float* d4 = dfl + ((long long)yhi * (W + 2) + xhi) * 256;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            atomicAdd(d1 + c * 64, hy * hx * gsc[c]);
            atomicAdd(d2 + c * 64, hy * lx * gsc[c]);
            atomicAdd(d3 + c * 64, ly * hx * gsc[c]);
            atomicAdd(d4 + c * 64, ly * lx * gsc[c]);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ROIAlign backward, owner-computes form (round 3).  The atomic kernel above adds every RoI's window into the fp32 maps with one float
// atomic per cell and channel: 4 096 RoIs of ~20 x 20 cells x 256 channels are 0.4 G atomics -- 1.4 ms at batch 8, at the chip's atomic
// rate, in an order that differs from run to run (so did the weights a training run produced).  Here every 8 x 8-cell REGION of a map is
// owned by one workgroup, which walks the RoIs that reach it in ENTRY ORDER and adds their contributions into registers (64 floats per
// thread: 8 cells x 8 channels), then adds the total into the map with plain loads and stores -- no atomics, a fixed summation order,
// bit-reproducible gradients.  The per-RoI tables (the same separable weights as the forward kernel) are built once per entry by a
// pre-pass; a RoI's contribution to a cell is computed exactly as the atomic kernel computes it.  RoIs whose window exceeds the
// tables (very elongated ones) are counted and left to the atomic kernel, which otherwise returns at once.
// ---------------------------------------------------------------------------------------------
struct RoiBwdTable {
  int img, lvl;               // lvl < 0: contributes nothing (invalid slot, empty window); lvl >= 4: left to the atomic kernel
  int y0, y1, x0, x1;         // cell window [y0, y1) x [x0, x1) of the whole RoI at its level
  float count;
  int pad_;
  int base[2][RS_ROI_PMAX], len[2][RS_ROI_PMAX];
  float w[2][RS_ROI_PMAX][RS_ROI_WMAX];
};
static_assert(sizeof(RoiBwdTable) == RS_ROI_BWD_TABLE_BYTES, "RS_ROI_BWD_TABLE_BYTES");

__global__ __launch_bounds__(64) void roi_bwd_prep_kernel(const RoiAlignParams p, RoiBwdTable* tabs, int* n_overflow) {
  __shared__ float s_w[2][RS_ROI_PMAX][RS_ROI_WMAX];
  __shared__ int s_base[2][RS_ROI_PMAX], s_len[2][RS_ROI_PMAX];
  const int entry = blockIdx.x, tid = threadIdx.x;
  RoiBwdTable* T = tabs + entry;
  int n_entries = p.S;
  if (p.n_entries) { const int c = *p.n_entries; n_entries = c < n_entries ? c : n_entries; }
  bool valid = entry < n_entries;
  int slot = 0, n = 0;
  if (valid) {
    slot = p.slot_list ? p.slot_list[entry] : entry;
    n = slot / p.slots_per_image;
    if (p.per_image_count && (slot - n * p.slots_per_image) >= p.per_image_count[n]) valid = false;
  }
  if (!valid) { if (tid == 0) { T->img = -1; T->lvl = -1; } return; }
  const int P = p.P;
  const float* r = p.rois + (long long)slot * 4;
  const float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
  const float area = (x2 - x1) * (y2 - y1);
  const float v = rs_fdiv(sqrtf(area), 224.0f) + 1e-8f;
  int lvl = v >= 2.0f ? 3 : (v >= 1.0f ? 2 : (v >= 0.5f ? 1 : 0));
  if (lvl > p.nlevels - 1) lvl = p.nlevels - 1;
  const int H = p.H[lvl], W = p.W[lvl];
  const float sc = p.scale[lvl];
  const float roi_start_w = x1 * sc - 0.5f;
  const float roi_start_h = y1 * sc - 0.5f;
  const float roi_w = (x2 * sc - 0.5f) - roi_start_w;
  const float roi_h = (y2 * sc - 0.5f) - roi_start_h;
  const float bin_h = rs_fdiv(roi_h, (float)P);
  const float bin_w = rs_fdiv(roi_w, (float)P);
  int gh = (int)ceilf(rs_fdiv(roi_h, (float)P));
  int gw = (int)ceilf(rs_fdiv(roi_w, (float)P));
  if (gh < 0) gh = 0;
  if (gw < 0) gw = 0;
  const float count = (float)((gh * gw) > 1 ? (gh * gw) : 1);
  if ((tid < P) || (tid >= 32 && tid < 32 + P)) {               // the forward kernel's tables, built the same way
    const int ax = tid >= 32 ? 1 : 0;
    const int b = ax ? tid - 32 : tid;
    const int g = ax ? gw : gh;
    const int size = ax ? W : H;
    const float start = ax ? roi_start_w : roi_start_h;
    const float bin = ax ? bin_w : bin_h;
    float* w = s_w[ax][b];
    for (int j = 0; j < RS_ROI_WMAX; ++j) w[j] = 0.f;
    int base = 0, len = 0;
    bool have = false, overflow = false;
    for (int i = 0; i < g; ++i) {
      float c = start + (float)b * bin + rs_fdiv(((float)i + 0.5f) * bin, (float)g);
      if (c < -1.0f || c > (float)size) continue;
      if (c <= 0.f) c = 0.f;
      int lo = (int)c, hi;
      if (lo >= size - 1) { hi = lo = size - 1; c = (float)lo; } else { hi = lo + 1; }
      const float l = c - (float)lo, h = 1.f - l;
      if (!have) { base = lo; have = true; }
      if (hi - base >= RS_ROI_WMAX) { overflow = true; break; }
      w[lo - base] += h;
      w[hi - base] += l;
      len = hi - base + 1;
    }
    s_base[ax][b] = base;
    s_len[ax][b] = overflow ? -1 : len;
  }
  __syncthreads();
  if (tid == 0) {
    int org[2], end[2], bad = 0;
    for (int ax = 0; ax < 2; ++ax) {
      org[ax] = 0x7fffffff; end[ax] = -1;
      for (int b = 0; b < P; ++b) {
        const int len = s_len[ax][b];
        if (len < 0) { bad = 1; break; }
        if (len == 0) continue;
        org[ax] = min(org[ax], s_base[ax][b]);
        end[ax] = max(end[ax], s_base[ax][b] + len);
      }
      if (end[ax] < 0) { org[ax] = 0; end[ax] = 0; }
      if (end[ax] - org[ax] > RS_ROI_CELLS) bad = 1;             // the atomic kernel's own criterion for its gather form
    }
    T->img = n;
    T->lvl = bad ? 4 + lvl : ((end[0] > org[0] && end[1] > org[1]) ? lvl : -1);
    T->y0 = org[0]; T->y1 = end[0]; T->x0 = org[1]; T->x1 = end[1];
    T->count = count;
    if (bad) atomicAdd(n_overflow, 1);
  }
  for (int i = tid; i < 2 * RS_ROI_PMAX; i += 64) { (&T->base[0][0])[i] = (&s_base[0][0])[i]; (&T->len[0][0])[i] = (&s_len[0][0])[i]; }
  for (int i = tid; i < 2 * RS_ROI_PMAX * RS_ROI_WMAX; i += 64) (&T->w[0][0][0])[i] = (&s_w[0][0][0])[i];
}

struct RoiBwdGeom { int rh[4], rw[4], off[5]; };   // regions per level (rows, columns) and their running sum per image

template <typename G>
__device__ __forceinline__ void roi_grad8(const G* gp, float g[8]);
template <>
__device__ __forceinline__ void roi_grad8<half_t>(const half_t* gp, float g[8]) {
  const half8 v = *(const half8*)gp;
#pragma unroll
  for (int c = 0; c < 8; ++c) g[c] = (float)v[c];
}
template <>
__device__ __forceinline__ void roi_grad8<float>(const float* gp, float g[8]) {
  const f32x4 a = *(const f32x4*)gp, b = *(const f32x4*)(gp + 4);
  g[0] = a[0]; g[1] = a[1]; g[2] = a[2]; g[3] = a[3]; g[4] = b[0]; g[5] = b[1]; g[6] = b[2]; g[7] = b[3];
}

#define RS_ROI_BWD_LIST 2048     // RoIs of one image a region can meet (box head: 512 per image, mask head: 256)
#define RS_ROI_BWD_TABW (2 * RS_ROI_PMAX * 2 + 2 * RS_ROI_PMAX * RS_ROI_WMAX)   // words of base + len + w in a RoiBwdTable
template <typename G>
__global__ __launch_bounds__(256) void roi_bwd_gather_kernel(const RoiAlignParams p, const RoiBwdTable* tabs, const RoiBwdGeom geo) {
  __shared__ unsigned short s_list[RS_ROI_BWD_LIST];
  __shared__ int s_cnt, s_wave[4];
  __shared__ int s_tab[2][RS_ROI_BWD_TABW];    // two RoIs' base / len / w (the RoiBwdTable layout from `base` on): one in use, one being filled
  __shared__ float s_inv[2];
  const int tid = threadIdx.x, hw = tid >> 5, l32 = tid & 31, wv = tid >> 6, ln = tid & 63;
  const int per_image = geo.off[4];
  const int n = blockIdx.x / per_image;
  int rr = blockIdx.x - n * per_image;
  int lvl = 0;
  while (lvl < 3 && rr >= geo.off[lvl + 1]) ++lvl;
  rr -= geo.off[lvl];
  const int ry = rr / geo.rw[lvl], rx = rr - ry * geo.rw[lvl];
  const int cy0 = ry * 8, cx0 = rx * 8;
  const int H = p.H[lvl], W = p.W[lvl];
  // ---- the entries of this image that reach the region, in entry order
  int e0, e1;
  if (p.slot_list) { e0 = 0; e1 = p.S; if (p.n_entries) { const int c = *p.n_entries; e1 = c < e1 ? c : e1; } }
  else { e0 = n * p.slots_per_image; e1 = e0 + p.slots_per_image; if (e1 > p.S) e1 = p.S; }
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  for (int c0 = e0; c0 < e1; c0 += 256) {
    const int e = c0 + tid;
    bool hit = false;
    if (e < e1) {
      const RoiBwdTable* T = tabs + e;
      hit = T->lvl == lvl && T->img == n && T->y0 < cy0 + 8 && T->y1 > cy0 && T->x0 < cx0 + 8 && T->x1 > cx0;
    }
    const unsigned long long m = __ballot(hit);
    if (ln == 0) s_wave[wv] = __popcll(m);
    __syncthreads();
    int pos = s_cnt + __popcll(m & ((1ull << ln) - 1ull));
    for (int w = 0; w < wv; ++w) pos += s_wave[w];
    if (hit && pos < RS_ROI_BWD_LIST) s_list[pos] = (unsigned short)(e - e0);
    __syncthreads();
    if (tid == 0) { int c = s_cnt + s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3]; s_cnt = c < RS_ROI_BWD_LIST ? c : RS_ROI_BWD_LIST; }
    __syncthreads();
  }
  const int cnt = s_cnt;
  if (cnt == 0) return;                         // nothing reaches the region: the map keeps what it holds
  const int P = p.P, PP = P + 2 * p.out_pad;
  const int y = cy0 + hw;                       // this half-wave's row of the region; the thread owns 8 channels of its 8 cells
  float acc[8][8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[x][c] = 0.f;
  // table of list entry k -> registers (3 words per thread), then -> s_tab[k & 1]; the loads of k + 1 are in flight under the work on k
  constexpr int TW = (RS_ROI_BWD_TABW + 255) / 256;
  int treg[TW];
  float tinv = 0.f;
  auto fetch = [&](int k) {
    const RoiBwdTable* T = tabs + e0 + s_list[k];
    const int* src = &T->base[0][0];
#pragma unroll
    for (int i = 0; i < TW; ++i) { const int o = tid + i * 256; treg[i] = o < RS_ROI_BWD_TABW ? src[o] : 0; }
    tinv = rs_fdiv(1.0f, T->count);
  };
  auto commit = [&](int k) {
#pragma unroll
    for (int i = 0; i < TW; ++i) { const int o = tid + i * 256; if (o < RS_ROI_BWD_TABW) s_tab[k & 1][o] = treg[i]; }
    if (tid == 0) s_inv[k & 1] = tinv;
  };
  fetch(0);
  commit(0);
  __syncthreads();
  for (int k = 0; k < cnt; ++k) {
    if (k + 1 < cnt) fetch(k + 1);
    const int* tb = s_tab[k & 1];
    const int* base0 = tb;                                   // [2][PMAX] base, [2][PMAX] len, [2][PMAX][WMAX] w
    const int* base1 = tb + RS_ROI_PMAX;
    const int* len0 = tb + 2 * RS_ROI_PMAX;
    const int* len1 = tb + 3 * RS_ROI_PMAX;
    const float* w0 = (const float*)(tb + 4 * RS_ROI_PMAX);
    const float* w1 = w0 + RS_ROI_PMAX * RS_ROI_WMAX;
    const float inv = s_inv[k & 1];
    // bins whose column window meets the region's 8 columns (uniform over the workgroup)
    int qlo = P, qhi = 0;
    for (int pw = 0; pw < P; ++pw)
      if (len1[pw] > 0 && base1[pw] < cx0 + 8 && base1[pw] + len1[pw] > cx0) { qlo = min(qlo, pw); qhi = max(qhi, pw + 1); }
    const G* gout = (const G*)p.out + (long long)(e0 + s_list[k]) * PP * PP * 256 + l32 * 8;
    for (int ph = 0; ph < P; ++ph) {
      const int jy = y - base0[ph];
      if (jy < 0 || jy >= len0[ph]) continue;
      const float wy = w0[ph * RS_ROI_WMAX + jy];
      if (wy == 0.f) continue;
      const float wyc = wy * inv;                            // weight of the row and 1 / (samples per bin), once per bin row
      const G* grow = gout + (long long)(ph + p.out_pad) * PP * 256;
      for (int q0 = qlo; q0 < qhi; q0 += 6) {
        float g[6][8];
#pragma unroll
        for (int u = 0; u < 6; ++u) {                        // six independent loads in flight (slots past the range re-load its first bin)
          const int pw = q0 + u < qhi ? q0 + u : qlo;
          roi_grad8<G>(grow + (long long)(pw + p.out_pad) * 256, g[u]);
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int pw = q0 + u;
          if (pw >= qhi) break;
          const int bx = base1[pw], lx = len1[pw];
          const float* wx = w1 + pw * RS_ROI_WMAX;
#pragma unroll
          for (int x = 0; x < 8; ++x) {
            const int jx = cx0 + x - bx;
            if (jx < 0 || jx >= lx) continue;
            const float wgt = wyc * wx[jx];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[x][c] += wgt * g[u][c];
          }
        }
      }
    }
    if (k + 1 < cnt) commit(k + 1);             // the other buffer was last read before the barrier that ended iteration k - 1
    __syncthreads();
  }
  if (y >= H) return;
  float* drow = p.dfeat[lvl] + (((long long)n * (H + 2) + y + 1) * (W + 2) + cx0 + 1) * 256 + l32 * 8;
#pragma unroll
  for (int x = 0; x < 8; ++x) {
    if (cx0 + x >= W) break;
    f32x4* d = (f32x4*)(drow + (long long)x * 256);
    f32x4 a = d[0], b = d[1];
    a[0] += acc[x][0]; a[1] += acc[x][1]; a[2] += acc[x][2]; a[3] += acc[x][3];
    b[0] += acc[x][4]; b[1] += acc[x][5]; b[2] += acc[x][6]; b[3] += acc[x][7];
    d[0] = a; d[1] = b;
  }
}

// ---------------------------------------------------------------------------------------------
// Box head: softmax + per-class decode + threshold + per-(image,class) sort
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void box_candidates_kernel(const BoxCandParams p) {
  __shared__ unsigned long long list[1024];
  __shared__ unsigned int cnt;
  const int k = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int K = p.K;
  const int np = p.prop_count[n];
  if (tid == 0) cnt = 0;
  __syncthreads();
  unsigned long long comp = 0ull;
  float box[4] = {0.f, 0.f, 0.f, 0.f};
  float score = 0.f;
  if (tid < np && tid < p.cap) {
    const float* pr = p.pred + ((long long)n * p.cap + tid) * p.cs;
    float mx = pr[0];
    for (int c = 1; c <= K; ++c) mx = fmaxf(mx, pr[c]);
    float sum = 0.f;
    for (int c = 0; c <= K; ++c) sum += expf(pr[c] - mx);
    score = rs_fdiv(expf(pr[k] - mx), sum);
    const float* pb = p.prop_boxes + ((long long)n * p.cap + tid) * 4;
    float b[4] = {pb[0], pb[1], pb[2], pb[3]};
    const float* dp = pr + (K + 1) + k * 4;
    float d[4] = {dp[0], dp[1], dp[2], dp[3]};
    apply_deltas(b, d, p.wx, p.wy, p.ww, p.wh, p.scale_clamp, box);
    box[0] = clampf(box[0], 0.f, p.img_w);
    box[1] = clampf(box[1], 0.f, p.img_h);
    box[2] = clampf(box[2], 0.f, p.img_w);
    box[3] = clampf(box[3], 0.f, p.img_h);
    float* db = p.dec_boxes + (((long long)n * p.cap + tid) * K + k) * 4;
    db[0] = box[0]; db[1] = box[1]; db[2] = box[2]; db[3] = box[3];
    p.dec_scores[((long long)n * p.cap + tid) * K + k] = score;
    if (score > p.score_thresh) {
      comp = ((unsigned long long)fkey(score) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)tid);
      atomicAdd(&cnt, 1u);
    }
  }
  list[tid] = comp;
  __syncthreads();
  bitonic_sort_desc<1024>(list, 1024, tid);
  const int c = (int)cnt;
  const long long sb = ((long long)n * K + k) * 1024;
  if (tid == 0) p.seg_count[n * K + k] = c;
  if (tid < c) {
    const int r = (int)(0xFFFFFFFFu - (uint32_t)(list[tid] & 0xFFFFFFFFull));
    const float* db = p.dec_boxes + (((long long)n * p.cap + r) * K + k) * 4;
    float* o = p.seg_boxes + (sb + tid) * 4;
    o[0] = db[0]; o[1] = db[1]; o[2] = db[2]; o[3] = db[3];
    p.seg_roi[sb + tid] = r;
  }
}

// Gather NMS survivors of all classes, order by score (ties: lower roi*K+class first), keep the
// first dets_per_image, then detector_postprocess's box part (scale to tile, clip, drop empty).
__global__ __launch_bounds__(1024) void det_merge_kernel(const DetMergeParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* list = (unsigned long long*)smem;   // 8192
  __shared__ unsigned int cnt;
  __shared__ unsigned char flag[1024];
  __shared__ int dst[1024];
  const int n = blockIdx.x, tid = threadIdx.x, K = p.K;
  if (tid == 0) cnt = 0;
  for (int i = tid; i < 8192; i += 1024) list[i] = 0ull;
  __syncthreads();
  for (int k = 0; k < K; ++k) {
    const int c = p.seg_count[n * K + k];
    const long long sb = ((long long)n * K + k) * 1024;
    for (int i = tid; i < c; i += 1024) {
      if (p.keep[sb + i]) {
        const int r = p.seg_roi[sb + i];
        const float sc = p.dec_scores[((long long)n * p.cap + r) * K + k];
        const unsigned pos = atomicAdd(&cnt, 1u);
        list[pos] = ((unsigned long long)fkey(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)(r * K + k));
      }
    }
  }
  __syncthreads();
  bitonic_sort_desc<1024>(list, sort_size(cnt, 1024, 8192), tid);
  const int D = p.dets_per_image;
  const int nd = (int)cnt < D ? (int)cnt : D;
  float bn[4] = {0, 0, 0, 0}, bo[4] = {0, 0, 0, 0};
  float sc = 0.f;
  int cls = 0, roi = 0;
  bool ok = false;
  if (tid < nd) {
    const uint32_t flat = 0xFFFFFFFFu - (uint32_t)(list[tid] & 0xFFFFFFFFull);
    roi = (int)(flat / (uint32_t)K);
    cls = (int)(flat - (uint32_t)roi * (uint32_t)K);
    const float* db = p.dec_boxes + (((long long)n * p.cap + roi) * K + cls) * 4;
    bn[0] = db[0]; bn[1] = db[1]; bn[2] = db[2]; bn[3] = db[3];
    sc = fkey_inv((uint32_t)(list[tid] >> 32));
    bo[0] = clampf(bn[0] * p.scale_x, 0.f, p.out_w);
    bo[1] = clampf(bn[1] * p.scale_y, 0.f, p.out_h);
    bo[2] = clampf(bn[2] * p.scale_x, 0.f, p.out_w);
    bo[3] = clampf(bn[3] * p.scale_y, 0.f, p.out_h);
    ok = ((bo[2] - bo[0]) > 0.f) && ((bo[3] - bo[1]) > 0.f);
  }
  flag[tid] = ok ? 1 : 0;
  __syncthreads();
  if (tid == 0) {
    int c = 0;
    for (int i = 0; i < nd; ++i) { dst[i] = c; c += flag[i]; }
    p.det_count[n] = c;
  }
  __syncthreads();
  if (ok) {
    const long long s = (long long)n * D + dst[tid];
    float* o1 = p.det_boxes_net + s * 4;
    float* o2 = p.det_boxes + s * 4;
    o1[0] = bn[0]; o1[1] = bn[1]; o1[2] = bn[2]; o1[3] = bn[3];
    o2[0] = bo[0]; o2[1] = bo[1]; o2[2] = bo[2]; o2[3] = bo[3];
    p.det_scores[s] = sc;
    p.det_classes[s] = cls;
    if (p.det_roi) p.det_roi[s] = roi;
  }
}

// exclusive scan of per-image detection counts -> compact entry list for the mask head
__global__ __launch_bounds__(1024) void det_compact_kernel(const int* det_count, int N, int D, int* slot_list, int* total) {
  __shared__ int off[1025];
  const int tid = threadIdx.x;
  if (tid == 0) {
    int c = 0;
    for (int i = 0; i < N; ++i) { off[i] = c; c += det_count[i]; }
    off[N] = c;
    *total = c;
  }
  __syncthreads();
  for (int n = 0; n < N; ++n) {
    const int c = det_count[n];
    for (int i = tid; i < c; i += 1024) slot_list[off[n] + i] = n * D + i;
  }
}

// ---------------------------------------------------------------------------------------------
// Mask predictor (1x1 conv, predicted class only) + sigmoid.  in: [R][S][S][256] fp16
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_predict_kernel(const MaskPredictParams p) {
  const int total = *p.n_entries;
  const int SS = p.S * p.S;
  const int tid = threadIdx.x;
  const int sub = tid & 15;         // 16 lanes per pixel, 16 channels each
  const long long pix = (long long)blockIdx.x * 16 + (tid >> 4);
  const long long npix = (long long)total * SS;
  if (pix >= npix) return;          // whole 16-lane group exits together
  const int entry = (int)(pix / SS);
  const int slot = p.slot_list[entry];
  const int cls = p.det_classes[slot];
  const float* w = p.w + (long long)cls * 256 + sub * 16;
  float xv[16];
  if (p.f32 == 2) {         // split-operand mode: hi + lo planes
    const half_t* x = p.in + pix * 256 + sub * 16;
    const half8 a = *(const half8*)x, b = *(const half8*)(x + 8), al = *(const half8*)(x + p.in_lo), bl = *(const half8*)(x + p.in_lo + 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) { xv[i] = (float)a[i] + (float)al[i]; xv[8 + i] = (float)b[i] + (float)bl[i]; }
  } else if (p.f32) {
    const float* x = (const float*)p.in + pix * 256 + sub * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) xv[i] = x[i];
  } else {
    const half_t* x = p.in + pix * 256 + sub * 16;
    const half8 a = *(const half8*)x, b = *(const half8*)(x + 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) { xv[i] = (float)a[i]; xv[8 + i] = (float)b[i]; }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += xv[i] * w[i];
  s += __shfl_xor(s, 8, 16);
  s += __shfl_xor(s, 4, 16);
  s += __shfl_xor(s, 2, 16);
  s += __shfl_xor(s, 1, 16);
  if (sub == 0) {
    const float logit = s + p.b[cls];
    const float prob = rs_fdiv(1.f, 1.f + expf(-logit));
    p.out[(long long)slot * SS + (pix - (long long)entry * SS)] = prob;
  }
}

// logits accumulated by the fused deconv+predictor GEMM -> + class bias -> sigmoid, in place ([slots][S][S])
__global__ __launch_bounds__(256) void mask_sigmoid_kernel(const MaskPredictParams p) {
  const int total = *p.n_entries;
  const int SS = p.S * p.S;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)total * SS) return;
  const int entry = (int)(gid / SS);
  const int slot = p.slot_list[entry];
  float* o = p.out + (long long)slot * SS + (gid - (long long)entry * SS);
  const float logit = *o + p.b[p.det_classes[slot]];
  *o = rs_fdiv(1.f, 1.f + expf(-logit));
}

// ---------------------------------------------------------------------------------------------
// paste_masks_in_image: one thread = 8 horizontally adjacent output pixels = one output byte
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void paste_masks_kernel(const PasteParams p) {
  // one thread = one 32-bit word = 32 horizontally adjacent output pixels (4 output bytes, one dword store)
  const int total = *p.n_entries;
  const int Wb = (p.out_w + 7) >> 3;          // bytes per output row
  const int Ww = (Wb + 3) >> 2;               // words per output row
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)p.out_h * Ww;
  if (gid >= (long long)total * per) return;
  const int entry = (int)(gid / per);
  const int rem = (int)(gid - (long long)entry * per);
  const int y = rem / Ww, xw = rem - y * Ww;
  const int slot = p.slot_list[entry];
  const float* bx = p.det_boxes + (long long)slot * 4;
  const float x0 = bx[0], y0 = bx[1], x1 = bx[2], y1 = bx[3];
  const float* m = p.probs + (long long)slot * p.S * p.S;
  const int S = p.S;
  unsigned int word = 0;
  // img_y = (y + 0.5 - y0) / (y1 - y0) * 2 - 1 ; iy = ((img_y + 1) * S - 1) / 2
  const float gy = rs_fdiv((float)y + 0.5f - y0, y1 - y0) * 2.f - 1.f;
  const float iy = ((gy + 1.f) * (float)S - 1.f) * 0.5f;
  const float fy = floorf(iy);
  const int iy0 = (int)fy, iy1 = iy0 + 1;
  const float wy1 = iy - fy, wy0 = 1.f - wy1;
  // conservative column range that can sample inside the SxS map (one mask texel of margin on both sides)
  const float margin = rs_fdiv(fabsf(x1 - x0), (float)S) + 1.f;
  const int xa = xw * 32, xz = xa + 31;
  if (iy1 >= 0 && iy0 < S && (float)xz + 0.5f >= fminf(x0, x1) - margin && (float)xa + 0.5f <= fmaxf(x0, x1) + margin) {
    for (int b = 0; b < 32; ++b) {
      const int x = xa + b;
      if (x >= p.out_w) break;
      const float gx = rs_fdiv((float)x + 0.5f - x0, x1 - x0) * 2.f - 1.f;
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
      if (v >= p.threshold) word |= 1u << b;
    }
  }
  uint8_t* o = p.out + (long long)slot * p.out_h * Wb + (long long)y * Wb + xw * 4;
  if ((Wb & 3) == 0) {
    *(unsigned int*)o = word;
  } else {
    for (int k = 0; k < 4 && xw * 4 + k < Wb; ++k) o[k] = (uint8_t)(word >> (8 * k));
  }
}

// ---------------------------------------------------------------------------------------------
// mask crops for the host: plan (sizes + exclusive scan, one workgroup) and copy (one workgroup per slot)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void mask_crop_plan_kernel(const CropParams p) {
  __shared__ unsigned int part[1024];
  __shared__ unsigned long long base_sh;
  const int tid = threadIdx.x;
  const int E = p.n * p.D;
  if (tid == 0) base_sh = 0ull;
  __syncthreads();
  for (int e0 = 0; e0 < E; e0 += 1024) {
    const int e = e0 + tid;
    unsigned int size = 0;
    int x0b = 0, y0 = 0, wb = 0, rows = 0;
    if (e < E) {
      const int i = e / p.D, d = e - i * p.D;
      if (d < p.det_count[i]) {
        const float* b = p.det_boxes + (long long)e * 4;
        // the pasted mask is zero at every pixel whose centre lies outside the box (grid_sample, zeros padding, >= 0.5 of a
        // probability < 1); one pixel of margin covers the rounding of the sample coordinates
        int xa = (int)floorf(fminf(b[0], b[2])) - 1, xz = (int)ceilf(fmaxf(b[0], b[2])) + 1;
        int ya = (int)floorf(fminf(b[1], b[3])) - 1, yz = (int)ceilf(fmaxf(b[1], b[3])) + 1;
        xa = xa < 0 ? 0 : xa; ya = ya < 0 ? 0 : ya;
        xz = xz > p.w - 1 ? p.w - 1 : xz; yz = yz > p.h - 1 ? p.h - 1 : yz;
        if (xz >= xa && yz >= ya) {
          x0b = xa >> 3; wb = (xz >> 3) - x0b + 1; y0 = ya; rows = yz - ya + 1;
          size = (unsigned int)(wb * rows);
        }
      }
    }
    part[tid] = size;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // inclusive scan (Hillis-Steele)
      const unsigned int v = tid >= off ? part[tid - off] : 0u;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    const unsigned long long base = base_sh;
    if (e < E) {
      int* r = p.rects + (long long)e * 4;
      r[0] = x0b; r[1] = y0; r[2] = wb; r[3] = rows;
      p.offsets[e] = (unsigned int)(base + part[tid] - size);
    }
    __syncthreads();
    if (tid == 1023) base_sh = base + part[1023];
    __syncthreads();
  }
  if (tid == 0) *p.total = base_sh;
}

__global__ __launch_bounds__(256) void mask_crop_copy_kernel(const CropParams p) {
  const int e = blockIdx.x;
  const int* r = p.rects + (long long)e * 4;
  const int x0b = r[0], y0 = r[1], wb = r[2], rows = r[3];
  const int total = wb * rows;
  if (total == 0) return;
  const uint8_t* src = p.masks + ((long long)e * p.h + y0) * p.Wb + x0b;
  uint8_t* dst = p.data + p.offsets[e];
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int rr = idx / wb, c = idx - rr * wb;
    dst[idx] = src[(long long)rr * p.Wb + c];
  }
}

}  // namespace

// ----------------------------------------------------------------------------------------------- launchers
int launch_mask_crops(const CropParams& p, hipStream_t s) {
  RS_CHECK(p.n > 0 && p.D > 0 && p.masks && p.rects && p.offsets && p.total && p.data, RS_ERR_ARG, "mask crops: bad argument");
  hipLaunchKernelGGL(mask_crop_plan_kernel, dim3(1), dim3(1024), 0, s, p);
  hipLaunchKernelGGL(mask_crop_copy_kernel, dim3(p.n * p.D), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_rpn_select(const RpnParams& p, hipStream_t s) {
  const int cap = p.cand_cap ? p.cand_cap : 1024;
  RS_CHECK((cap == 1024 || cap == 2048) && p.topk <= cap && p.A <= RS_MAX_ANCHORS && p.L <= RS_MAX_LEVELS, RS_ERR_UNSUPPORTED,
           "rpn: topk %d / capacity %d / A %d / L %d out of range", p.topk, cap, p.A, p.L);
  RpnParams q = p;
  q.debug = rs_debug().select_debug;
  if (cap == 1024) hipLaunchKernelGGL(rpn_select_kernel<1024>, dim3(p.L, p.N), dim3(1024), 0, s, q);
  else hipLaunchKernelGGL(rpn_select_kernel<2048>, dim3(p.L, p.N), dim3(1024), 0, s, q);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_nms(const NmsParams& p, int segments, hipStream_t s) {
  const int lds = 1024 * 20 + 256 + 1024 * 128;
  const int lds_big = 2048 * 20 + 256;
  static bool done = false;
  if (!done) {
    RS_HIP(hipFuncSetAttribute((const void*)nms_kernel<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    RS_HIP(hipFuncSetAttribute((const void*)nms_kernel<2048, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big));
    done = true;
  }
  RS_CHECK(p.cap > 0 && p.cap <= 2048 && (p.cap <= 1024 || p.scratch), RS_ERR_ARG, "nms: capacity %d (more than 1024 boxes need NmsParams::scratch)", p.cap);
  NmsParams q = p;
  q.debug = rs_debug().nms_debug;
  // few segments (images x levels / classes) and a quadratic mask build: with one workgroup per segment most of the chip idles, so the rows of a
  // segment are shared by `parts` workgroups (mask in global memory) and the (serial, cheap) scan runs as a second launch.  Same tests, same
  // greedy scan: identical keep flags.
  int parts = segments >= 256 ? 1 : (256 + segments - 1) / segments;
  if (parts > 16) parts = 16;
  if (p.cap <= 1024) {
    if (!p.scratch || segments > 32 || parts == 1) hipLaunchKernelGGL((nms_kernel<1024, false>), dim3(segments), dim3(1024), lds, s, q);
    else {
      if (parts > 8) parts = 8;
      q.mode = 1;
      hipLaunchKernelGGL((nms_kernel<1024, true>), dim3(segments, parts), dim3(1024), 1024 * 20 + 256, s, q);
      q.mode = 2;
      hipLaunchKernelGGL((nms_kernel<1024, true>), dim3(segments), dim3(1024), 1024 * 20 + 256, s, q);
    }
  } else if (parts == 1) {
    q.mode = 0;
    hipLaunchKernelGGL((nms_kernel<2048, true>), dim3(segments), dim3(1024), lds_big, s, q);
  } else {
    q.mode = 1;
    hipLaunchKernelGGL((nms_kernel<2048, true>), dim3(segments, parts), dim3(1024), lds_big, s, q);
    q.mode = 2;
    hipLaunchKernelGGL((nms_kernel<2048, true>), dim3(segments), dim3(1024), lds_big, s, q);
  }
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_rpn_merge(const RpnMergeParams& p, int N, hipStream_t s) {
  const int cap = p.cand_cap ? p.cand_cap : 1024;
  RS_CHECK(p.L <= 8 && (cap == 1024 || cap == 2048), RS_ERR_UNSUPPORTED, "rpn merge: %d levels / capacity %d", p.L, cap);
  static bool done = false;
  if (!done) {
    RS_HIP(hipFuncSetAttribute((const void*)rpn_merge_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    RS_HIP(hipFuncSetAttribute((const void*)rpn_merge_kernel<2048>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    done = true;
  }
  if (cap == 1024) hipLaunchKernelGGL(rpn_merge_kernel<1024>, dim3(N), dim3(1024), 65536, s, p);
  else hipLaunchKernelGGL(rpn_merge_kernel<2048>, dim3(N), dim3(1024), 131072, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_roi_align(const RoiAlignParams& p, hipStream_t s) {
  RS_CHECK(p.C == 256, RS_ERR_UNSUPPORTED, "roi_align: C must be 256 (got %d)", p.C);
  RS_CHECK(p.S > 0, RS_ERR_ARG, "roi_align: S");
  const int use_win = rs_debug().roi_window;
  // the windowed kernel pre-sums the sample weights per feature row / column (fp32 rounding of the weights: ~1e-7 relative); the fp32 validation
  // mode keeps torchvision's per-sample order, the split-operand mode takes the window (RS_ROI_WINDOW=2: per-sample there too)
  if (!p.f32 && use_win && p.P <= RS_ROI_PMAX) hipLaunchKernelGGL(roi_align_win_kernel<false>, dim3(p.S), dim3(256), 0, s, p);
  else if (p.f32 == 2 && use_win == 1 && p.P <= RS_ROI_PMAX) hipLaunchKernelGGL(roi_align_win_kernel<true>, dim3(p.S), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(roi_align_kernel, dim3(p.S), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_roi_align_bwd(const RoiAlignParams& p_in, hipStream_t s) {
  RoiAlignParams p = p_in;
  RS_CHECK(p.C == 256 && p.P <= RS_ROI_PMAX, RS_ERR_UNSUPPORTED, "roi_align backward: C must be 256, P <= %d", RS_ROI_PMAX);
  RS_CHECK(p.S > 0, RS_ERR_ARG, "roi_align backward: S");
  for (int l = 0; l < p.nlevels; ++l) RS_CHECK(p.dfeat[l] != nullptr, RS_ERR_ARG, "roi_align backward: null gradient map");
  const bool gather = p.bwd_tables && p.bwd_overflow && p.n_images > 0 && p.nlevels == 4 && p.slots_per_image <= RS_ROI_BWD_LIST &&
                      (!p.slot_list || p.S <= RS_ROI_BWD_LIST * 8) && rs_debug().roi_bwd_atomic == 0;
  if (!gather) p.bwd_overflow = nullptr;                       // the atomic kernel serves every entry
  else {
    RoiBwdGeom geo;
    geo.off[0] = 0;
    for (int l = 0; l < 4; ++l) { geo.rh[l] = cdiv(p.H[l], 8); geo.rw[l] = cdiv(p.W[l], 8); geo.off[l + 1] = geo.off[l] + geo.rh[l] * geo.rw[l]; }
    RS_HIP(hipMemsetAsync(p.bwd_overflow, 0, sizeof(int), s));
    hipLaunchKernelGGL(roi_bwd_prep_kernel, dim3(p.S), dim3(64), 0, s, p, (RoiBwdTable*)p.bwd_tables, p.bwd_overflow);
    const dim3 grid((unsigned)(p.n_images * geo.off[4]));
    if (p.f32) hipLaunchKernelGGL(roi_bwd_gather_kernel<float>, grid, dim3(256), 0, s, p, (const RoiBwdTable*)p.bwd_tables, geo);
    else hipLaunchKernelGGL(roi_bwd_gather_kernel<half_t>, grid, dim3(256), 0, s, p, (const RoiBwdTable*)p.bwd_tables, geo);
  }
  if (p.f32) hipLaunchKernelGGL(roi_align_bwd_kernel<float>, dim3(p.S), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(roi_align_bwd_kernel<half_t>, dim3(p.S), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_box_candidates(const BoxCandParams& p, int N, hipStream_t s) {
  RS_CHECK(p.cap <= 1024, RS_ERR_UNSUPPORTED, "box head: more than 1024 proposals per image");
  hipLaunchKernelGGL(box_candidates_kernel, dim3(p.K, N), dim3(1024), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_det_merge(const DetMergeParams& p, int N, hipStream_t s) {
  RS_CHECK(p.K * 1024 <= 8192, RS_ERR_UNSUPPORTED, "det merge: NUM_CLASSES %d > 8 not supported yet", p.K);
  RS_CHECK(p.dets_per_image <= 1024, RS_ERR_UNSUPPORTED, "det merge: DETECTIONS_PER_IMAGE > 1024");
  static bool done = false;
  if (!done) {
    RS_HIP(hipFuncSetAttribute((const void*)det_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    done = true;
  }
  hipLaunchKernelGGL(det_merge_kernel, dim3(N), dim3(1024), 65536, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_det_compact(const int* det_count, int N, int D, int* slot_list, int* total, hipStream_t s) {
  RS_CHECK(N <= 1024, RS_ERR_UNSUPPORTED, "batch > 1024");
  hipLaunchKernelGGL(det_compact_kernel, dim3(1), dim3(1024), 0, s, det_count, N, D, slot_list, total);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_mask_predict(const MaskPredictParams& p, int capacity_entries, hipStream_t s) {
  const long long npix = (long long)capacity_entries * p.S * p.S;
  hipLaunchKernelGGL(mask_predict_kernel, dim3(cdiv(npix, 16)), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_mask_sigmoid(const MaskPredictParams& p, int capacity_entries, hipStream_t s) {
  const long long npix = (long long)capacity_entries * p.S * p.S;
  hipLaunchKernelGGL(mask_sigmoid_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_paste_masks(const PasteParams& p, int capacity_entries, hipStream_t s) {
  const long long total = (long long)capacity_entries * p.out_h * ((((p.out_w + 7) >> 3) + 3) >> 2);
  hipLaunchKernelGGL(paste_masks_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
