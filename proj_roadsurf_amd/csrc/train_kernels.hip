// Training-path element-wise / reduction kernels.  fp32 math in detectron2's
// operation order, compiled with -ffp-contract=off like the detection glue.
//
//   rpn_loss_kernel        RPN.losses: BCE-with-logits on the sampled anchors + L1 (SMOOTH_L1_BETA 0.0, R:251) on the
//                          positive anchors' deltas vs Box2BoxTransform.get_deltas(anchor, matched gt), both divided by
//                          BATCH_SIZE_PER_IMAGE * N (R:223)  [EXT d2: modeling/proposal_generator/rpn.py losses,
//                          modeling/box_regression.py get_deltas / _dense_box_regression_loss]
//   box_loss_kernel        FastRCNNOutputLayers.losses: softmax cross-entropy (mean over all sampled RoIs) + L1 on the
//                          gt class' deltas of foreground RoIs / number of sampled RoIs (weights (10,10,5,5), R:160-164,175)
//                          [EXT d2: modeling/roi_heads/fast_rcnn.py]
//   mask_loss_kernel       mask_rcnn_loss: BCE-with-logits of the gt class' 28x28 logits vs the rasterised gt mask, mean over
//                          all foreground masks and pixels  [EXT d2: modeling/roi_heads/mask_head.py]
//   sgd_momentum_kernel    torch.optim.SGD step: g += wd*w; buf = mu*buf + g; w -= lr*buf (MOMENTUM 0.9, WEIGHT_DECAY 1e-4,
//                          no Nesterov, R:281-282,303)  [EXT d2: solver/build.py; torch/optim/sgd.py]
//   fold_weights_kernel    fp32 master weight -> the two fp16 GEMM operands of the next step: the forward weight
//                          [Cout][(kh,kw,ci)] with the FrozenBN scale folded in, and its transposed, tap-flipped copy
//                          [Cin][(kh',kw',co)] for the input-gradient convolution
// Each loss kernel writes the GRADIENT of the (weighted) total loss w.r.t. its logits/deltas, scaled by `loss_scale`
// (fp16 loss scaling), and accumulates the loss value itself into an fp32 scalar for logging.
#include "common.h"
#include "train.h"

namespace {

__device__ __forceinline__ float bce_with_logits(float x, float t) {
  // torch: max(x,0) - x*t + log1p(exp(-|x|))
  return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) { return rs_fdiv(1.f, 1.f + expf(-x)); }
__device__ __forceinline__ float sgnf(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }

__device__ __forceinline__ void get_deltas(const float s[4], const float t[4], float wx, float wy, float ww, float wh, float out[4]) {
  const float sw = s[2] - s[0], sh = s[3] - s[1];
  const float scx = s[0] + 0.5f * sw, scy = s[1] + 0.5f * sh;
  const float tw = t[2] - t[0], th = t[3] - t[1];
  const float tcx = t[0] + 0.5f * tw, tcy = t[1] + 0.5f * th;
  out[0] = rs_fdiv(wx * (tcx - scx), sw);
  out[1] = rs_fdiv(wy * (tcy - scy), sh);
  out[2] = ww * logf(rs_fdiv(tw, sw));
  out[3] = wh * logf(rs_fdiv(th, sh));
}

__device__ __forceinline__ void block_sum_to(float v, float* dst) {
  // wave reduction, then one atomic per wave (loss logging only; gradients never go through atomics here)
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0 && v != 0.f) atomicAdd(dst, v);
}

// one thread per anchor of one level; head output rows are [pixel][cs] fp32: columns [0,A) logits, [A,5A) deltas (a*4+d)
template <typename G>
__global__ __launch_bounds__(256) void rpn_loss_kernel(const RpnLossParams p) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;          // anchor index inside the level, order (y, x, a)
  float lc = 0.f, ll = 0.f;
  if (i < p.n_anchors) {
    const int px = i / p.A, a = i - px * p.A;
    const long long row = ((long long)n * p.HW + px) * p.cs;
    const int label = p.labels[(long long)n * p.total_anchors + p.level_off + i];
    const int dcs = p.dcs ? p.dcs : p.cs;
    G* g = (G*)p.dhead + ((long long)n * p.HW + px) * dcs;
    float gl = 0.f, gd[4] = {0.f, 0.f, 0.f, 0.f};
    if (label >= 0) {
      const float x = p.head[row + a], t = (float)label;
      lc = rs_fdiv(bce_with_logits(x, t), p.normalizer);
      gl = rs_fdiv(sigmoidf(x) - t, p.normalizer);
      if (label == 1) {
        const float* an = p.anchors + ((long long)p.level_off + i) * 4;
        const long long ai = (long long)n * p.total_anchors + p.level_off + i;
        const float* gt = p.matched_gt ? p.matched_gt + ai * 4 : p.gt + ((long long)n * p.gt_cap + p.matched[ai]) * 4;
        const float s[4] = {an[0], an[1], an[2], an[3]}, t4[4] = {gt[0], gt[1], gt[2], gt[3]};
        float tgt[4];
        get_deltas(s, t4, 1.f, 1.f, 1.f, 1.f, tgt);
        for (int d = 0; d < 4; ++d) {
          const float diff = p.head[row + p.A + a * 4 + d] - tgt[d];
          ll += rs_fdiv(fabsf(diff), p.normalizer);
          gd[d] = rs_fdiv(sgnf(diff), p.normalizer);
        }
      }
    }
    g[a] = (G)(gl * p.loss_scale);
    for (int d = 0; d < 4; ++d) g[p.A + a * 4 + d] = (G)(gd[d] * p.loss_scale);
    if (a == 0) for (int c = 5 * p.A; c < dcs; ++c) g[c] = (G)0.f;       // padding columns of the fused head
  }
  block_sum_to(lc, p.loss_out);
  block_sum_to(ll, p.loss_out + 1);
}

// one thread per sampled RoI; pred rows [roi][cs] fp32: [0,K] class logits (background = K), then 4K deltas
template <typename G>
__global__ __launch_bounds__(256) void box_loss_kernel(const BoxLossParams p) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  float lc = 0.f, ll = 0.f;
  float n_valid = p.n_valid;
  if (p.n_valid_counts) {
    int t = 0;
    for (int i = 0; i < p.n_valid_n * 2; ++i) t += p.n_valid_counts[i];
    n_valid = (float)(t > 0 ? t : 1);
  }
  if (r < p.n_rois) {
    const float* pr = p.pred + (long long)r * p.cs;
    const int dcs = p.dcs ? p.dcs : p.cs;
    G* g = (G*)p.dpred + (long long)r * dcs;
    const int K = p.K;
    const int cls = p.gt_classes[r];                       // 0..K-1 foreground, K background, -1 ignored slot
    for (int c = 0; c < dcs; ++c) g[c] = (G)0.f;
    if (cls >= 0 && cls <= K) {                            // anything else (empty slot, corrupt label) contributes nothing
      float mx = pr[0];
      for (int c = 1; c <= K; ++c) mx = fmaxf(mx, pr[c]);
      float sum = 0.f;
      for (int c = 0; c <= K; ++c) sum += expf(pr[c] - mx);
      const float lse = mx + logf(sum);
      lc = rs_fdiv(lse - pr[cls], n_valid);
      for (int c = 0; c <= K; ++c) {
        const float sm = expf(pr[c] - lse);
        g[c] = (G)(rs_fdiv(sm - (c == cls ? 1.f : 0.f), n_valid) * p.loss_scale);
      }
      if (cls < K) {
        const float* pb = p.proposals + (long long)r * 4;
        const float* gb = p.gt_boxes + (long long)r * 4;
        const float s[4] = {pb[0], pb[1], pb[2], pb[3]}, t4[4] = {gb[0], gb[1], gb[2], gb[3]};
        float tgt[4];
        get_deltas(s, t4, p.wx, p.wy, p.ww, p.wh, tgt);
        for (int d = 0; d < 4; ++d) {
          const float diff = pr[K + 1 + cls * 4 + d] - tgt[d];
          ll += rs_fdiv(fabsf(diff), n_valid);
          g[K + 1 + cls * 4 + d] = (G)(rs_fdiv(sgnf(diff), n_valid) * p.loss_scale);
        }
      }
    }
  }
  block_sum_to(lc, p.loss_out);
  block_sum_to(ll, p.loss_out + 1);
}

// one thread per (mask, pixel); logits [mask][S*S][cs] fp32 (channel = class), targets [mask][S*S] uint8
template <typename G>
__global__ __launch_bounds__(256) void mask_loss_kernel(const MaskLossParams p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  float l = 0.f;
  const long long per = (long long)p.S * p.S;
  int n_masks = p.n_masks;
  if (p.n_masks_ptr) { const int c = *p.n_masks_ptr; n_masks = c < n_masks ? c : n_masks; }
  if (i < (long long)n_masks * per) {
    const int m = (int)(i / per);
    const int cls = p.gt_classes[m];
    const int dcs = p.dcs ? p.dcs : p.cs;
    G* g = (G*)p.dlogits + i * dcs;
    for (int c = 0; c < dcs; ++c) g[c] = (G)0.f;
    if (cls >= 0 && cls < p.cs) {
      const float x = p.logits[i * p.cs + cls], t = (float)p.targets[i];
      const float norm = (float)n_masks * (float)per;
      l = rs_fdiv(bce_with_logits(x, t), norm);
      g[cls] = (G)(rs_fdiv(sigmoidf(x) - t, norm) * p.loss_scale);
    }
  }
  block_sum_to(l, p.loss_out);
}

// skip: optional device flag (gradient overflow found by grad_nonfinite_kernel): when set the step is not applied -- neither the
// weights nor the momentum buffer change (what torch.cuda.amp.GradScaler.step does with an inf/nan gradient)
__global__ __launch_bounds__(256) void sgd_momentum_kernel(float* w, float* buf, const float* grad, long long n, float lr, float momentum,
                                                          float weight_decay, float inv_loss_scale, int first_step, const int* skip) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (skip && *skip) return;
  float g = grad[i] * inv_loss_scale;
  g = g + weight_decay * w[i];
  const float b = first_step ? g : momentum * buf[i] + g;       // torch: the buffer starts as a clone of the first gradient
  buf[i] = b;
  w[i] = w[i] - lr * b;
}

// fp16 loss scaling: the activation gradients are fp16, so a too large scale shows up as inf / nan in the fp32 weight gradients
// (an inf activation gradient poisons every weight gradient downstream of it).  One flag for the whole flat buffer.
__global__ __launch_bounds__(256) void grad_nonfinite_kernel(const float* grad, long long n, int* flag) {
  const long long stride = (long long)gridDim.x * 256 * 4;
  bool bad = false;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const f32x4 v = *(const f32x4*)(grad + i);
      bad = bad || !(fabsf(v[0]) <= 3.0e38f) || !(fabsf(v[1]) <= 3.0e38f) || !(fabsf(v[2]) <= 3.0e38f) || !(fabsf(v[3]) <= 3.0e38f);
    } else {
      for (long long j = i; j < n; ++j) bad = bad || !(fabsf(grad[j]) <= 3.0e38f);
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// w32: [Cout][Kpad] fp32 master, K = (kh,kw,ci); fwd: same layout fp16 with scale[co] folded; bwd: [Cin][KpadT] fp16,
// column ((KH-1-kh)*KW + (KW-1-kw))*kc + co (kc = Cout rounded up to the 64-channel K step; the padding columns stay zero)
// One block = one 32 (co) x 32 (ci) tile of one tap: the master rows are read along ci, the forward operand written in the
// same layout, and the tile goes through LDS so that the transposed copy is written along co (64-byte runs instead of one
// 2-byte element per row: the scattered form cost 0.31 ms per step for the 44 M weights, this one is bandwidth-bound).
template <typename T>
__device__ __forceinline__ void fold_tile(const FoldDesc& d, unsigned blk, T (*tile)[34]) {
  const int cit = (d.Cin + 31) >> 5, taps = d.KH * d.KW;
  const int ct = (int)(blk % cit);
  const unsigned r = blk / cit;
  const int tap = (int)(r % taps), cot = (int)(r / taps);
  const int kh = tap / d.KW, kw = tap - kh * d.KW;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = cot * 32 + ty + j * 8, ci = ct * 32 + tx;
    T h = (T)0.f;
    if (co < d.Cout && ci < d.Cin) {
      const long long idx = (long long)co * d.Kpad + tap * d.Cin + ci;
      float v = d.w32[idx] * (d.scale ? d.scale[co] : 1.f);
      // keep the fp32 product as its own rounding step: hipcc otherwise selects v_fma_mixlo_f16 (product rounded ONCE, to
      // fp16), which differs from the host fold (numpy: fp32 multiply, then astype(float16)) on fp32-rounding ties -- measured
      // 5 of 73,728 weights one fp16 ulp apart
      asm volatile("" : "+v"(v));
      h = (T)v;
      ((T*)d.fwd)[idx] = h;
    }
    tile[ty + j * 8][tx] = h;
  }
  if (!d.bwd) return;                       // block-uniform
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ci = ct * 32 + ty + j * 8, co = cot * 32 + tx;
    if (co < d.Cout && ci < d.Cin)          // kc >= Cout: channel stride per tap
      ((T*)d.bwd)[(long long)ci * d.KpadT + ((d.KH - 1 - kh) * d.KW + (d.KW - 1 - kw)) * d.kc + co] = tile[tx][ty + j * 8];
  }
}
__global__ __launch_bounds__(256) void fold_weights_kernel(const FoldDesc d) {
  __shared__ float tile32[32][34];
  if (d.f32) fold_tile<float>(d, blockIdx.x, tile32);
  else fold_tile<half_t>(d, blockIdx.x, (half_t(*)[34])tile32);
}
// every layer of a trainer in ONE launch: the block looks its layer up in the table (block_start ascending, n_desc <= a few
// hundred: binary search), then folds one tile of it. A descriptor with Cout == 0 is a bias copy: fwd32[g*n + j] =
// w32[j] for g < tile (the 2x2 deconv's forward bias is the master bias repeated per GEMM), 256 elements per block.
__global__ __launch_bounds__(256) void fold_table_kernel(const FoldDesc* __restrict__ table, int n_desc) {
  __shared__ float tile32[32][34];
  int lo = 0, hi = n_desc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block_start <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const FoldDesc d = table[lo];
  if (d.Cout == 0) {
    const long long i = (long long)(blockIdx.x - d.block_start) * 256 + threadIdx.x;
    const long long n = (long long)d.Cin * d.KH;           // Cin = length, KH = tile count
    if (i < n) d.fwd32[i] = d.w32[i % d.Cin];
    return;
  }
  if (d.f32) fold_tile<float>(d, blockIdx.x - d.block_start, tile32);     // block-uniform
  else fold_tile<half_t>(d, blockIdx.x - d.block_start, (half_t(*)[34])tile32);
}

// ---------------------------------------------------------------------------------------------
// Matcher ([EXT d2: modeling/matcher.py]) on pairwise_iou ([EXT d2: structures/boxes.py]): per box the best gt
// (lowest index on ties) and the label of its IoU band; optionally "low-quality matches": every box whose IoU with
// some gt EQUALS that gt's highest IoU over all boxes becomes positive (second kernel, same IoU arithmetic bit for bit).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float iou_d2(const float a[4], float area_a, const float b[4], float area_b) {
  const float w = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
  const float h = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
  const float inter = fmaxf(w, 0.f) * fmaxf(h, 0.f);
  return inter > 0.f ? rs_fdiv(inter, area_a + area_b - inter) : 0.f;
}

__global__ __launch_bounds__(256) void match_kernel(const MatchParams p) {
  __shared__ float s_gt[256 * 4];
  __shared__ float s_area[256];
  const int n = blockIdx.y;
  const int G = p.gt_count[n] < p.gt_cap ? p.gt_count[n] : p.gt_cap;
  for (int i = threadIdx.x; i < G; i += 256) {
    const float* g = p.gt + ((long long)n * p.gt_cap + i) * 4;
    s_gt[i * 4] = g[0]; s_gt[i * 4 + 1] = g[1]; s_gt[i * 4 + 2] = g[2]; s_gt[i * 4 + 3] = g[3];
    s_area[i] = (g[2] - g[0]) * (g[3] - g[1]);
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n_boxes) return;
  const long long o = (long long)n * p.n_boxes + i;
  if (p.box_count && i >= p.box_count[n]) { p.matched[o] = 0; p.labels[o] = -1; if (p.best_iou) p.best_iou[o] = 0.f; return; }
  const float* bp = p.boxes + (p.per_image_boxes ? o : (long long)i) * 4;
  const float b[4] = {bp[0], bp[1], bp[2], bp[3]};
  const float area = (b[2] - b[0]) * (b[3] - b[1]);
  float best = -1.f;
  int arg = 0;
  for (int g = 0; g < G; ++g) {
    const float v = iou_d2(&s_gt[g * 4], s_area[g], b, area);
    if (v > best) { best = v; arg = g; }
    if (p.gt_best) atomicMax(p.gt_best + (long long)n * p.gt_cap + g, __float_as_uint(v));   // v >= 0: bit order == value order
  }
  int label = p.lbl_lo;
  if (G > 0) label = best >= p.t_hi ? p.lbl_hi : (best >= p.t_lo ? p.lbl_mid : p.lbl_lo);
  p.matched[o] = arg;
  p.labels[o] = label;
  if (p.best_iou) p.best_iou[o] = G > 0 ? best : 0.f;
}

__global__ __launch_bounds__(256) void match_lowq_kernel(const MatchParams p) {
  __shared__ float s_gt[256 * 4];
  __shared__ float s_area[256];
  __shared__ float s_best[256];
  const int n = blockIdx.y;
  const int G = p.gt_count[n] < p.gt_cap ? p.gt_count[n] : p.gt_cap;
  for (int i = threadIdx.x; i < G; i += 256) {
    const float* g = p.gt + ((long long)n * p.gt_cap + i) * 4;
    s_gt[i * 4] = g[0]; s_gt[i * 4 + 1] = g[1]; s_gt[i * 4 + 2] = g[2]; s_gt[i * 4 + 3] = g[3];
    s_area[i] = (g[2] - g[0]) * (g[3] - g[1]);
    s_best[i] = __uint_as_float(p.gt_best[(long long)n * p.gt_cap + i]);
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n_boxes) return;
  const long long o = (long long)n * p.n_boxes + i;
  if (p.box_count && i >= p.box_count[n]) return;
  const float* bp = p.boxes + (p.per_image_boxes ? o : (long long)i) * 4;
  const float b[4] = {bp[0], bp[1], bp[2], bp[3]};
  const float area = (b[2] - b[0]) * (b[3] - b[1]);
  bool hit = false;
  for (int g = 0; g < G; ++g) hit = hit || (iou_d2(&s_gt[g * 4], s_area[g], b, area) == s_best[g]);
  if (hit) p.labels[o] = 1;
}

// ---------------------------------------------------------------------------------------------
// subsample_labels ([EXT d2: modeling/sampling.py]): up to floor(num_samples * positive_fraction) positives and the rest
// negatives, uniformly at random.  Every candidate gets a 64-bit key (hash(seed, image, index) << 32 | index); the k
// candidates with the SMALLEST keys are the sample (a random permutation's first k) -- found with an 8-pass radix
// select, deterministic for a given seed.  One workgroup per image.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int mix32(unsigned int x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned long long sample_key(unsigned int seed, int n, int i) {
  const unsigned int h = mix32(mix32(seed ^ (0x9e3779b9u * (unsigned)(n + 1))) + (unsigned)i * 0x85ebca6bu);
  return ((unsigned long long)h << 32) | (unsigned int)i;
}

// smallest key such that exactly k candidates of `group` (positives or negatives) have key <= it; k >= 1
__device__ unsigned long long kth_key(const int* labels, int n_el, int bg, bool want_pos, int k, unsigned int seed, int img, int* hist) {
  unsigned long long prefix = 0ull;
  int need = k;
  for (int pass = 7; pass >= 0; --pass) {
    for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const int shift = pass * 8;
    const unsigned long long hi_mask = pass == 7 ? 0ull : (~0ull << (shift + 8));
    for (int i = threadIdx.x; i < n_el; i += blockDim.x) {
      const int l = labels[i];
      const bool cand = want_pos ? (l != -1 && l != bg) : (l == bg);
      if (!cand) continue;
      const unsigned long long key = sample_key(seed, img, i);
      if ((key & hi_mask) != (prefix & hi_mask)) continue;
      atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
    }
    __syncthreads();
    int acc = 0, digit = 0;
    for (int b = 0; b < 256; ++b) {          // every thread walks the 256 bins: same result everywhere, no extra barrier
      if (acc + hist[b] >= need) { digit = b; break; }
      acc += hist[b];
    }
    need -= acc;
    prefix |= (unsigned long long)digit << shift;
    __syncthreads();
  }
  return prefix;
}

__global__ __launch_bounds__(1024) void subsample_kernel(const SubsampleParams p) {
  __shared__ int hist[256];
  __shared__ int s_cnt[2];
  __shared__ int s_out[2];
  const int n = blockIdx.x;
  int* labels = p.labels + (long long)n * p.n;
  if (threadIdx.x < 2) { s_cnt[threadIdx.x] = 0; s_out[threadIdx.x] = 0; }
  __syncthreads();
  int cp = 0, cn = 0;
  for (int i = threadIdx.x; i < p.n; i += blockDim.x) {
    const int l = labels[i];
    if (l == p.bg_label) ++cn; else if (l != -1) ++cp;
  }
  for (int o = 32; o > 0; o >>= 1) { cp += __shfl_xor(cp, o); cn += __shfl_xor(cn, o); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cnt[0], cp); atomicAdd(&s_cnt[1], cn); }
  __syncthreads();
  const int P = s_cnt[0], Q = s_cnt[1];
  int num_pos = (int)((float)p.num_samples * p.positive_fraction);
  if (num_pos > P) num_pos = P;
  int num_neg = p.num_samples - num_pos;
  if (num_neg > Q) num_neg = Q;
  unsigned long long thr_pos = 0ull, thr_neg = 0ull;
  if (num_pos > 0 && num_pos < P) thr_pos = kth_key(labels, p.n, p.bg_label, true, num_pos, p.seed, n, hist);
  if (num_neg > 0 && num_neg < Q) thr_neg = kth_key(labels, p.n, p.bg_label, false, num_neg, p.seed, n, hist);
  __syncthreads();
  if (threadIdx.x == 0 && p.sampled_count) { p.sampled_count[n * 2] = num_pos; p.sampled_count[n * 2 + 1] = num_neg; }
  if (p.sampled) for (int i = threadIdx.x; i < p.num_samples; i += blockDim.x) p.sampled[(long long)n * p.num_samples + i] = -1;
  __syncthreads();
  // selection pass; in roi mode the list positions come from an ordered scan (ascending index inside each group)
  for (int base = 0; base < p.n; base += blockDim.x) {
    const int i = base + threadIdx.x;
    int sel = 0;       // 1 = sampled positive, 2 = sampled negative
    if (i < p.n) {
      const int l = labels[i];
      if (l == p.bg_label) {
        if (num_neg > 0 && (num_neg >= Q || sample_key(p.seed, n, i) <= thr_neg)) sel = 2;
      } else if (l != -1) {
        if (num_pos > 0 && (num_pos >= P || sample_key(p.seed, n, i) <= thr_pos)) sel = 1;
      }
      if (p.rpn_mode) labels[i] = sel == 1 ? 1 : (sel == 2 ? 0 : -1);
    }
    if (p.sampled) {
      // wave-level ordered compaction, waves in order through s_out
      for (int grp = 1; grp <= 2; ++grp) {
        const unsigned long long bal = __ballot(sel == grp);
        const int lane = threadIdx.x & 63;
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        const int wcount = __popcll(bal);
        __shared__ int w_off[16];
        if (lane == 0) w_off[threadIdx.x >> 6] = wcount;
        __syncthreads();
        int off = s_out[grp - 1];
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += w_off[w];
        if (sel == grp) p.sampled[(long long)n * p.num_samples + (grp == 2 ? num_pos : 0) + off + before] = i;
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += w_off[w]; s_out[grp - 1] += t; }
        __syncthreads();
      }
    }
  }
}

// bias gradient: grad[c] = sum over rows of dy[row][c] (halo rows are zero, so the whole buffer can be summed).
// Stage 1: workgroup = one row slice over ALL channels: thread t owns the 8-channel group t % (C/8) of rows t / (C/8),
// + 256/(C/8), ... of its slice, so a wave reads whole contiguous rows (C = 256: two rows per load instruction); fixed
// order inside (per thread ascending rows, then the row groups in ascending order) -> scratch[z][c].  Stage 2 adds the
// slices in order (bitwise reproducible, no atomics).
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T* dy, long long rows, int C, int cout, float* scratch,
                                                       const int* m_count, int m_mul) {
  typedef T V8 __attribute__((ext_vector_type(8)));
  __shared__ float red[256][9];
  if (m_count) { const long long mc = (long long)(*m_count) * m_mul; if (mc < rows) rows = mc; }
  const int cgs = C >> 3;
  for (int cg0 = 0; cg0 < cgs; cg0 += 256) {          // C <= 2048: one pass
    const int ncg = min(256, cgs - cg0);
    const int rpi = 256 / ncg;                        // rows per load round
    const int cg = threadIdx.x % ncg, ro = threadIdx.x / ncg;
    const long long per = (rows + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * per;
    long long r1 = r0 + per;
    if (r1 > rows) r1 = rows;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ro < rpi) {
      const T* src = dy + (long long)(cg0 + cg) * 8;
      long long r = r0 + ro;
      for (; r + 3 * rpi < r1; r += 4 * rpi) {
        const V8 v0 = *(const V8*)(src + r * C), v1 = *(const V8*)(src + (r + rpi) * C);
        const V8 v2 = *(const V8*)(src + (r + 2 * rpi) * C), v3 = *(const V8*)(src + (r + 3 * rpi) * C);
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i] += (float)v0[i]; acc[i] += (float)v1[i]; acc[i] += (float)v2[i]; acc[i] += (float)v3[i]; }
      }
      for (; r < r1; r += rpi) {
        const V8 v = *(const V8*)(src + r * C);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc[i];
    __syncthreads();
    // thread (cg, i): sum over the row groups in ascending order
    for (int t = threadIdx.x; t < ncg * 8; t += 256) {
      const int g = t >> 3, i = t & 7;
      float sum = 0.f;
      for (int q = 0; q < rpi; ++q) sum += red[q * ncg + g][i];
      const int c = (cg0 + g) * 8 + i;
      if (c < cout) scratch[(long long)blockIdx.x * cout + c] = sum;
    }
  }
}
// block = 16 channels x 16 slice groups: group g adds the slices z = g, g+16, ... in ascending order, then the 16 group sums
// are added in ascending order (fixed order, hence reproducible)
__global__ __launch_bounds__(256) void bias_grad_reduce_kernel(const float* scratch, int slices, int cout, float* grad, int accumulate) {
  __shared__ float part[16][17];
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + e;
  float s = 0.f;
  if (c < cout) {
    int z = g;
    for (; z + 48 < slices; z += 64) {
      const float a = scratch[(long long)z * cout + c], b = scratch[(long long)(z + 16) * cout + c];
      const float d = scratch[(long long)(z + 32) * cout + c], f = scratch[(long long)(z + 48) * cout + c];
      s += a; s += b; s += d; s += f;
    }
    for (; z < slices; z += 16) s += scratch[(long long)z * cout + c];
  }
  part[g][e] = s;
  __syncthreads();
  if (g != 0 || c >= cout) return;
  s = part[0][e];
  for (int q = 1; q < 16; ++q) s += part[q][e];
  grad[c] = accumulate ? grad[c] + s : s;
}

// backward of LastLevelMaxPool (max_pool2d k=1 s=2): d_fine[2y][2x] += d_coarse[y][x]; both NHWC fp16 with halo 1
template <typename T>
__global__ __launch_bounds__(256) void subsample2_bwd_kernel(const T* dc, T* df, int N, int Hf, int Wf, int Hc, int Wc, int C) {
  typedef T V8 __attribute__((ext_vector_type(8)));
  const int cv = C >> 3;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * Hc * Wc * cv;
  if (gid >= total) return;
  const int c8 = (int)(gid % cv);
  long long t = gid / cv;
  const int x = (int)(t % Wc); t /= Wc;
  const int y = (int)(t % Hc);
  const int n = (int)(t / Hc);
  const V8 a = *(const V8*)(dc + (((long long)n * (Hc + 2) + y + 1) * (Wc + 2) + x + 1) * C + c8 * 8);
  T* o = df + (((long long)n * (Hf + 2) + 2 * y + 1) * (Wf + 2) + 2 * x + 1) * C + c8 * 8;
  V8 b = *(const V8*)o;
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = (T)((float)b[i] + (float)a[i]);
  *(V8*)o = b;
}

// ---------------------------------------------------------------------------------------------
// ROIHeads.label_and_sample_proposals glue ([EXT d2: modeling/roi_heads/roi_heads.py]): candidates = proposals + gt boxes,
// Matcher labels -> classes, sampled candidates -> the box head's fixed-capacity proposal buffer and its targets.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roi_candidates_kernel(const RoiSampleParams p) {
  const int n = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.cand_cap) return;
  const int np = p.prop_count[n] < p.prop_cap ? p.prop_count[n] : p.prop_cap;
  const int ng = p.gt_count[n] < p.gt_cap ? p.gt_count[n] : p.gt_cap;
  float b[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < np) { const float* s = p.prop_boxes + ((long long)n * p.prop_cap + i) * 4; b[0] = s[0]; b[1] = s[1]; b[2] = s[2]; b[3] = s[3]; }
  else if (i < np + ng) { const float* s = p.gt_boxes + ((long long)n * p.gt_cap + (i - np)) * 4; b[0] = s[0]; b[1] = s[1]; b[2] = s[2]; b[3] = s[3]; }
  float* o = p.cand_boxes + ((long long)n * p.cand_cap + i) * 4;
  o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3];
  if (i == 0) p.cand_count[n] = (np + ng) < p.cand_cap ? (np + ng) : p.cand_cap;
}

__global__ __launch_bounds__(256) void roi_classes_kernel(const RoiSampleParams p) {
  const int n = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.cand_cap) return;
  const long long o = (long long)n * p.cand_cap + i;
  int cls = -1;
  if (i < p.cand_count[n]) {
    const int lab = p.labels[o];
    if (p.gt_count[n] <= 0) cls = p.K;                        // no ground truth: everything is background
    else cls = lab == 1 ? p.gt_classes[(long long)n * p.gt_cap + p.matched[o]] : (lab == 0 ? p.K : -1);
  }
  p.labels[o] = cls;
}

__global__ __launch_bounds__(256) void roi_gather_kernel(const RoiSampleParams p) {
  const int n = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p.out_cap) return;
  const int total = p.sampled_count[n * 2] + p.sampled_count[n * 2 + 1];
  const long long o = (long long)n * p.out_cap + j;
  float b[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f};
  int cls = -1, gi = 0;
  if (j < total && j < p.num_samples) {
    const int c = p.sampled[(long long)n * p.num_samples + j];
    const long long ci = (long long)n * p.cand_cap + c;
    const float* s = p.cand_boxes + ci * 4;
    b[0] = s[0]; b[1] = s[1]; b[2] = s[2]; b[3] = s[3];
    cls = p.labels[ci];
    gi = p.matched[ci];
    if (p.gt_count[n] > 0) { const float* q = p.gt_boxes + ((long long)n * p.gt_cap + gi) * 4; g[0] = q[0]; g[1] = q[1]; g[2] = q[2]; g[3] = q[3]; }
  }
  float* ob = p.out_boxes + o * 4;
  ob[0] = b[0]; ob[1] = b[1]; ob[2] = b[2]; ob[3] = b[3];
  float* og = p.out_gt_boxes + o * 4;
  og[0] = g[0]; og[1] = g[1]; og[2] = g[2]; og[3] = g[3];
  p.out_classes[o] = cls;
  p.out_gt_index[o] = gi;
  if (j == 0) p.out_count[n] = total < p.out_cap ? total : p.out_cap;
}

// select_foreground_proposals ([EXT d2: modeling/roi_heads/roi_heads.py]): the sampled set is foreground-first, so image n's
// mask-head entries are its slots j < (number of sampled foreground)
__global__ __launch_bounds__(256) void mask_entries_kernel(const MaskEntriesParams p) {
  __shared__ int s_off[65];
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int n = 0; n < p.N && n < 64; ++n) {
      s_off[n] = acc;
      int nf = p.sampled_count[n * 2];
      if (nf > p.per_image_cap) nf = p.per_image_cap;
      if (acc + nf > p.cap) nf = p.cap - acc;
      acc += nf;
    }
    s_off[p.N < 64 ? p.N : 64] = acc;
    *p.total = acc;
  }
  __syncthreads();
  for (int n = 0; n < p.N && n < 64; ++n) {
    const int nf = s_off[n + 1] - s_off[n];
    for (int j = threadIdx.x; j < nf; j += 256) {
      const int slot = n * p.slots_per_image + j;
      p.slots[s_off[n] + j] = slot;
      p.classes[s_off[n] + j] = p.roi_classes[slot];
    }
  }
}

}  // namespace

int launch_rpn_loss(const RpnLossParams& p, int N, hipStream_t s) {
  RS_CHECK(p.head && p.dhead && p.labels && p.anchors && (p.matched_gt || (p.gt && p.matched)) && p.loss_out && p.n_anchors > 0 && p.cs >= 5 * p.A, RS_ERR_ARG, "rpn_loss: bad arguments");
  if (p.d32) hipLaunchKernelGGL(rpn_loss_kernel<float>, dim3(cdiv(p.n_anchors, 256), N), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(rpn_loss_kernel<half_t>, dim3(cdiv(p.n_anchors, 256), N), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
int launch_box_loss(const BoxLossParams& p, hipStream_t s) {
  RS_CHECK(p.pred && p.dpred && p.gt_classes && p.proposals && p.gt_boxes && p.loss_out && p.n_rois > 0 && p.cs >= 5 * p.K + 1 && (p.n_valid > 0 || p.n_valid_counts), RS_ERR_ARG, "box_loss: bad arguments");
  if (p.d32) hipLaunchKernelGGL(box_loss_kernel<float>, dim3(cdiv(p.n_rois, 256)), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(box_loss_kernel<half_t>, dim3(cdiv(p.n_rois, 256)), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
int launch_mask_loss(const MaskLossParams& p, hipStream_t s) {
  RS_CHECK(p.logits && p.dlogits && p.targets && p.gt_classes && p.loss_out && p.n_masks > 0 && p.S > 0, RS_ERR_ARG, "mask_loss: bad arguments");
  if (p.d32) hipLaunchKernelGGL(mask_loss_kernel<float>, dim3((unsigned)cdiv((long long)p.n_masks * p.S * p.S, 256)), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(mask_loss_kernel<half_t>, dim3((unsigned)cdiv((long long)p.n_masks * p.S * p.S, 256)), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
int launch_sgd_momentum(float* w, float* buf, const float* grad, long long n, float lr, float momentum, float weight_decay,
                        float inv_loss_scale, int first_step, hipStream_t s, const int* skip) {
  RS_CHECK(w && buf && grad && n > 0, RS_ERR_ARG, "sgd: bad arguments");
  hipLaunchKernelGGL(sgd_momentum_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, w, buf, grad, n, lr, momentum, weight_decay,
                     inv_loss_scale, first_step, skip);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
int launch_grad_nonfinite(const float* grad, long long n, int* flag, hipStream_t s) {
  RS_CHECK(grad && flag && n > 0, RS_ERR_ARG, "grad check: bad arguments");
  RS_HIP(hipMemsetAsync(flag, 0, 4, s));
  hipLaunchKernelGGL(grad_nonfinite_kernel, dim3(2048), dim3(256), 0, s, grad, n, flag);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
int launch_fold_weights(const float* w32, const float* scale, half_t* fwd, half_t* bwd, int Cout, int Cin, int KH, int KW, int Kpad,
                        int kc, int KpadT, hipStream_t s) {
  RS_CHECK(w32 && fwd && Cout > 0 && Cin > 0 && kc >= Cout && (!bwd || KH * KW * kc <= KpadT), RS_ERR_ARG, "fold: bad arguments");
  FoldDesc d = {};
  d.w32 = w32; d.scale = scale; d.fwd = fwd; d.bwd = bwd;
  d.Cout = Cout; d.Cin = Cin; d.KH = KH; d.KW = KW; d.Kpad = Kpad; d.kc = kc; d.KpadT = KpadT;
  hipLaunchKernelGGL(fold_weights_kernel, dim3(fold_desc_blocks(d)), dim3(256), 0, s, d);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
unsigned fold_desc_blocks(const FoldDesc& d) {
  if (d.Cout == 0) return (unsigned)cdiv((long long)d.Cin * d.KH, 256);
  return (unsigned)(cdiv(d.Cout, 32) * d.KH * d.KW * cdiv(d.Cin, 32));
}
int launch_fold_table(const FoldDesc* table_dev, int n_desc, unsigned total_blocks, hipStream_t s) {
  RS_CHECK(table_dev && n_desc > 0 && total_blocks > 0, RS_ERR_ARG, "fold table: bad arguments");
  hipLaunchKernelGGL(fold_table_kernel, dim3(total_blocks), dim3(256), 0, s, table_dev, n_desc);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_match(const MatchParams& p, int N, hipStream_t s) {
  RS_CHECK(p.boxes && p.gt && p.gt_count && p.matched && p.labels && p.n_boxes > 0 && p.gt_cap >= 1 && p.gt_cap <= 256, RS_ERR_ARG,
           "match: bad arguments (gt capacity %d, at most 256)", p.gt_cap);
  if (p.gt_best) RS_HIP(hipMemsetAsync(p.gt_best, 0, (size_t)N * p.gt_cap * 4, s));
  hipLaunchKernelGGL(match_kernel, dim3(cdiv(p.n_boxes, 256), N), dim3(256), 0, s, p);
  if (p.gt_best) hipLaunchKernelGGL(match_lowq_kernel, dim3(cdiv(p.n_boxes, 256), N), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
int launch_subsample(const SubsampleParams& p, int N, hipStream_t s) {
  RS_CHECK(p.labels && p.n > 0 && p.num_samples > 0 && (p.rpn_mode || p.sampled), RS_ERR_ARG, "subsample: bad arguments");
  hipLaunchKernelGGL(subsample_kernel, dim3(N), dim3(1024), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_bias_grad(const half_t* dy, long long rows, int C, int cout, float* grad, int accumulate, hipStream_t s, const int* m_count,
                     int m_mul, float* scratch, int f32) {
  RS_CHECK(dy && grad && scratch && rows > 0 && C % 8 == 0 && cout > 0 && cout <= C, RS_ERR_ARG, "bias_grad: bad arguments");
  int slices = (int)(rows * (C >> 3) / 4096);    // >= 16 rows per thread and slice
  if (slices < 1) slices = 1;
  if (slices > RS_BIAS_GRAD_SLICES) slices = RS_BIAS_GRAD_SLICES;
  if (f32) hipLaunchKernelGGL(bias_grad_kernel<float>, dim3(slices), dim3(256), 0, s, (const float*)dy, rows, C, cout, scratch, m_count, m_mul);
  else hipLaunchKernelGGL(bias_grad_kernel<half_t>, dim3(slices), dim3(256), 0, s, dy, rows, C, cout, scratch, m_count, m_mul);
  hipLaunchKernelGGL(bias_grad_reduce_kernel, dim3(cdiv(cout, 16)), dim3(256), 0, s, scratch, slices, cout, grad, accumulate);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
int launch_subsample2_bwd(const half_t* d_coarse, half_t* d_fine, int N, int Hf, int Wf, int Hc, int Wc, int C, hipStream_t s, int f32) {
  RS_CHECK(d_coarse && d_fine && C % 8 == 0 && 2 * (Hc - 1) < Hf && 2 * (Wc - 1) < Wf, RS_ERR_ARG, "subsample2_bwd: bad arguments");
  const long long total = (long long)N * Hc * Wc * (C >> 3);
  if (f32) hipLaunchKernelGGL(subsample2_bwd_kernel<float>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, (const float*)d_coarse, (float*)d_fine, N, Hf, Wf, Hc, Wc, C);
  else hipLaunchKernelGGL(subsample2_bwd_kernel<half_t>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, d_coarse, d_fine, N, Hf, Wf, Hc, Wc, C);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_roi_candidates(const RoiSampleParams& p, int N, hipStream_t s) {
  RS_CHECK(p.prop_boxes && p.cand_boxes && p.cand_count && p.cand_cap > 0, RS_ERR_ARG, "roi_candidates: bad arguments");
  hipLaunchKernelGGL(roi_candidates_kernel, dim3(cdiv(p.cand_cap, 256), N), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
int launch_roi_classes(const RoiSampleParams& p, int N, hipStream_t s) {
  RS_CHECK(p.labels && p.matched && p.gt_classes && p.cand_count, RS_ERR_ARG, "roi_classes: bad arguments");
  hipLaunchKernelGGL(roi_classes_kernel, dim3(cdiv(p.cand_cap, 256), N), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
int launch_roi_gather(const RoiSampleParams& p, int N, hipStream_t s) {
  RS_CHECK(p.sampled && p.sampled_count && p.out_boxes && p.out_count && p.out_classes && p.out_gt_boxes && p.out_gt_index && p.num_samples <= p.out_cap,
           RS_ERR_ARG, "roi_gather: bad arguments");
  hipLaunchKernelGGL(roi_gather_kernel, dim3(cdiv(p.out_cap, 256), N), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int launch_mask_entries(const MaskEntriesParams& p, hipStream_t s) {
  RS_CHECK(p.sampled_count && p.roi_classes && p.slots && p.classes && p.total && p.N >= 1 && p.N <= 64, RS_ERR_ARG, "mask_entries: bad arguments");
  hipLaunchKernelGGL(mask_entries_kernel, dim3(1), dim3(256), 0, s, p);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
