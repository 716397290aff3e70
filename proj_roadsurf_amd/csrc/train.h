// Training-path kernels (train_kernels.hip): parameter blocks and launchers.
#pragma once
#include "common.h"

struct RpnLossParams {
  const float* head;        // one level's fused RPN head output [N][HW][cs] fp32: [0,A) logits, [A,5A) deltas
  half_t* dhead;            // gradient, same layout, fp16 (times loss_scale)
  const int* labels;        // [N][total_anchors]: 1 positive, 0 negative, -1 not sampled (after subsampling)
  const float* anchors;     // [total_anchors][4]
  const float* matched_gt;  // [N][total_anchors][4] matched gt box per anchor, or null: gathered from gt/matched below
  const float* gt;          // [N][gt_cap][4]
  const int* matched;       // [N][total_anchors] index of the matched gt
  float* loss_out;          // [2]: loss_rpn_cls, loss_rpn_loc (accumulated)
  int A, cs, dcs, HW, n_anchors, level_off, total_anchors, gt_cap;   // dcs: row stride of dhead (>= 5A; 0 = cs)
  float normalizer;         // BATCH_SIZE_PER_IMAGE * N
  float loss_scale;
  int d32;                  // 1: dhead points to fp32 (reference-precision trainer)
};
struct BoxLossParams {
  const float* pred;        // [n_rois][cs] fp32: [0,K] logits (K = background), then 4K deltas
  half_t* dpred;            // gradient fp16 (times loss_scale)
  const int* gt_classes;    // [n_rois]: class, K = background, -1 = empty slot (no loss)
  const float* proposals;   // [n_rois][4]
  const float* gt_boxes;    // [n_rois][4] matched gt box (used for foreground rows)
  float* loss_out;          // [2]: loss_cls, loss_box_reg
  int n_rois, K, cs, dcs;   // dcs: row stride of dpred (0 = cs)
  float n_valid;            // number of sampled RoIs (gt_classes >= 0): the mean's denominator; used when n_valid_counts is null
  const int* n_valid_counts;   // optional device array [n_valid_n][2] (sampled foreground, background per image): n_valid = their sum
  int n_valid_n;
  float wx, wy, ww, wh;
  float loss_scale;
  int d32;                  // 1: dpred points to fp32
};
struct MaskLossParams {
  const float* logits;      // [n_masks][S*S][cs] fp32, channel = class
  half_t* dlogits;          // gradient fp16 (times loss_scale)
  const uint8_t* targets;   // [n_masks][S*S] 0/1
  const int* gt_classes;    // [n_masks]
  float* loss_out;          // [1]
  const int* n_masks_ptr;   // optional device count (<= n_masks = capacity): the mean runs over it
  int n_masks, S, cs, dcs;  // dcs: row stride of dlogits (0 = cs)
  float loss_scale;
  int d32;                  // 1: dlogits points to fp32
};
int launch_rpn_loss(const RpnLossParams& p, int N, hipStream_t s);
int launch_box_loss(const BoxLossParams& p, hipStream_t s);
int launch_mask_loss(const MaskLossParams& p, hipStream_t s);
int launch_sgd_momentum(float* w, float* buf, const float* grad, long long n, float lr, float momentum, float weight_decay,
                        float inv_loss_scale, int first_step, hipStream_t s, const int* skip = nullptr);
int launch_grad_nonfinite(const float* grad, long long n, int* flag, hipStream_t s);   // flag = 1 if any inf / nan
// one layer of the fold (master fp32 -> fp16 GEMM operands); Cout == 0 marks a bias copy (w32 -> fwd32, Cin floats, KH times)
struct FoldDesc {
  const float* w32;
  const float* scale;
  half_t* fwd;
  half_t* bwd;
  float* fwd32;
  int Cout, Cin, KH, KW, Kpad, kc, KpadT;
  unsigned block_start;     // first block of this entry in the table launch
  int f32;                  // 1: fwd / bwd point to fp32 operands (reference-precision trainer: no rounding, same layouts)
};
unsigned fold_desc_blocks(const FoldDesc& d);
int launch_fold_table(const FoldDesc* table_dev, int n_desc, unsigned total_blocks, hipStream_t s);
int launch_fold_weights(const float* w32, const float* scale, half_t* fwd, half_t* bwd, int Cout, int Cin, int KH, int KW, int Kpad,
                        int kc, int KpadT, hipStream_t s);
#define RS_BIAS_GRAD_SLICES 1024  // scratch: RS_BIAS_GRAD_SLICES * cout floats
int launch_bias_grad(const half_t* dy, long long rows, int C, int cout, float* grad, int accumulate, hipStream_t s, const int* m_count,
                     int m_mul, float* scratch, int f32 = 0);
int launch_subsample2_bwd(const half_t* d_coarse, half_t* d_fine, int N, int Hf, int Wf, int Hc, int Wc, int C, hipStream_t s, int f32 = 0);

// ---- label assignment (Matcher + subsample_labels) ----
struct MatchParams {
  const float* boxes;       // [n_boxes][4] shared by all images (anchors), or [N][n_boxes][4] when per_image_boxes
  const float* gt;          // [N][gt_cap][4]
  const int* gt_count;      // [N]
  int* matched;             // [N][n_boxes] index of the best gt (0 when the image has none)
  int* labels;              // [N][n_boxes] label of the IoU band: lbl_lo (< t_lo), lbl_mid ([t_lo, t_hi)), lbl_hi (>= t_hi)
  float* best_iou;          // [N][n_boxes] (optional)
  unsigned int* gt_best;    // [N][gt_cap] scratch: bit pattern of every gt's highest IoU (zeroed by the launcher); null = no
                            // low-quality matches
  const int* box_count;     // optional [N]: boxes beyond it get label -1 (per-image proposal lists)
  int n_boxes, gt_cap, per_image_boxes;
  float t_lo, t_hi;
  int lbl_lo, lbl_mid, lbl_hi;
};
struct SubsampleParams {
  int* labels;              // [N][n] in: class / band label; out (rpn_mode): 1 sampled positive, 0 sampled negative, -1 rest
  int* sampled;             // [N][num_samples] (roi mode): sampled indices, positives first, each group ascending; -1 padded
  int* sampled_count;       // [N][2]: number of sampled positives, negatives
  int n, num_samples, bg_label, rpn_mode;
  float positive_fraction;
  unsigned int seed;        // changes every iteration
};
int launch_match(const MatchParams& p, int N, hipStream_t s);
int launch_subsample(const SubsampleParams& p, int N, hipStream_t s);

// ---- RoI-head proposal sampling glue (label_and_sample_proposals)
struct RoiSampleParams {
  const float* prop_boxes;  // [N][prop_cap][4] RPN proposals
  const int* prop_count;    // [N]
  const float* gt_boxes;    // [N][gt_cap][4]
  const int* gt_classes;    // [N][gt_cap]
  const int* gt_count;      // [N]
  float* cand_boxes;        // [N][cand_cap][4]: proposals then the image's gt boxes (PROPOSAL_APPEND_GT, R:193)
  int* cand_count;          // [N]
  int* matched;             // [N][cand_cap] (from the Matcher)
  int* labels;              // [N][cand_cap] in: Matcher label (1 / 0 / -1) -> out: class, K = background, -1 = no candidate
  const int* sampled;       // [N][num_samples] candidate indices (subsample), -1 padded
  const int* sampled_count; // [N][2]
  float* out_boxes;         // [N][out_cap][4] sampled boxes (the box head's proposal buffer)
  int* out_count;           // [N]
  int* out_classes;         // [N][out_cap] class, K = background, -1 = empty slot
  float* out_gt_boxes;      // [N][out_cap][4] matched gt box (foreground rows)
  int* out_gt_index;        // [N][out_cap] matched gt index
  int prop_cap, gt_cap, cand_cap, num_samples, out_cap, K;
};
int launch_roi_candidates(const RoiSampleParams& p, int N, hipStream_t s);
int launch_roi_classes(const RoiSampleParams& p, int N, hipStream_t s);
int launch_roi_gather(const RoiSampleParams& p, int N, hipStream_t s);

// foreground RoIs of the sampled set -> compact mask-head entry list
struct MaskEntriesParams {
  const int* sampled_count; // [N][2]
  const int* roi_classes;   // [N][slots_per_image]
  int* slots;               // [cap] out: slot = n * slots_per_image + j, image-major, j ascending
  int* classes;             // [cap] out
  int* total;               // [1] out
  int N, slots_per_image, per_image_cap, cap;
};
int launch_mask_entries(const MaskEntriesParams& p, hipStream_t s);
