// Parameter blocks of the proposal / detection kernels (detect_kernels.hip).
#pragma once
#include "common.h"

#define RS_MAX_LEVELS 5
#define RS_MAX_ANCHORS 8

struct RpnParams {
  const float* head[RS_MAX_LEVELS];   // [N][H][W][cs] fp32: channels [0,A) logits, [A,5A) deltas (a*4+d)
  uint32_t* keys[RS_MAX_LEVELS];      // scratch [N][2][H*W*A]: ordered keys, candidate indices
  int H[RS_MAX_LEVELS], W[RS_MAX_LEVELS], stride[RS_MAX_LEVELS];
  float base[RS_MAX_LEVELS][RS_MAX_ANCHORS][4];   // cell anchors (fp32 of the float64 formula)
  float offset;
  int L, A, N, cs;
  int topk;                 // PRE_NMS_TOPK_TEST (<= 1024)
  float img_h, img_w;       // clip size = resized image size
  const float* img_hw;      // optional [N][2] (h, w): clip size per image (training batches of mixed sizes); null = img_h / img_w
  float wx, wy, ww, wh, scale_clamp, min_size;
  float* cand_boxes;        // [N][L][1024][4]
  float* cand_scores;       // [N][L][1024]
  uint8_t* cand_valid;      // [N][L][1024]
  int* cand_count;          // [N][L]
  int* cand_index;          // optional [N][L][1024]: anchor index (y*W+x)*A+a
  int debug;                // timing experiments only: stop after phase <debug>
  int cand_cap;             // candidate slots per (image, level): 0/1024 (inference) or 2048 (training, PRE_NMS_TOPK_TRAIN 2000)
};

struct NmsParams {
  const float* boxes;       // [S][cap][4], priority order
  const int* count;         // [S]
  const uint8_t* valid;     // [S][cap] or nullptr
  uint8_t* keep;            // [S][cap]
  int cap;                  // 1024
  float thresh;
  int debug;                // timing experiments only: 1 = skip the scan, 2 = skip the mask build
  unsigned long long* scratch;   // cap > 1024: [segments][2048][32] suppression-mask words in global memory
  int mode;                 // cap > 1024 only: 0 = mask build + scan in one workgroup, 1 = mask build only (gridDim.y workgroups share
                            // a segment's rows), 2 = scan only (after a mode-1 launch)
};

struct RpnMergeParams {
  const float* cand_boxes;
  const float* cand_scores;
  const uint8_t* keep;
  const int* cand_count;
  int L, post_topk, cap;
  int cand_cap;             // 0/1024 or 2048: slots per (image, level) of the candidate arrays
  float* prop_boxes;        // [N][cap][4]
  float* prop_scores;       // [N][cap]
  int* prop_level;          // optional [N][cap]
  int* prop_count;          // [N]
  int* prop_order;          // optional [N][cap]: the image's slots sorted by (FPN level of the RoI pooler, top row, left column) -- the order
                            // box.roi_align visits them in (RoiAlignParams::order), so that neighbouring workgroups read neighbouring
                            // feature rows; slots beyond prop_count come last
};

struct RoiAlignParams {
  const half_t* feat[4];    // p2..p5, NHWC fp16, halo 1, 256 channels
  int H[4], W[4];
  float scale[4];
  int nlevels, C;
  const float* rois;        // [slots][4]
  const int* slot_list;     // optional entry -> slot
  const int* n_entries;     // optional device count of entries
  const int* per_image_count;   // optional: slot valid iff rank < count[image]
  int S;                    // entry capacity (grid size)
  int slots_per_image;
  half_t* out;              // [entry][P+2*out_pad][P+2*out_pad][256]
  int P, out_pad;
  int* out_level;           // optional [entry]
  const int* order;         // optional [S]: workgroup k processes entry order[remap(k)] (remap = the XCD-aware bijection of the conv
                            // kernels: each XCD's L2 sees one contiguous run of the order); results land in the entry's own slot
  int f32;                  // 1: fp32 validation mode, features and output are float; 2: split-operand mode, features and output are hi / lo
                            // fp16 planes (feat[l] / out = the hi plane, the lo plane feat_lo[l] / out_lo elements behind it)
  long long feat_lo[4], out_lo;
  // backward (roi_align_bwd_kernel): `out` holds the incoming gradient [entry][P+2*out_pad]^2[256] fp16 and the
  // gradient of the feature maps is accumulated (float atomics) into dfeat[level], fp32, same geometry as feat[level]
  float* dfeat[4];
  // owner-computes form of the backward (roi_bwd_prep_kernel + roi_bwd_gather_kernel): per-entry tables workspace (RS_ROI_BWD_TABLE_BYTES
  // each, S entries), a device counter of the entries left to the atomic kernel, and the number of images behind the entries.  Null
  // bwd_tables = the atomic kernel for every entry (round 1-2 behaviour).
  void* bwd_tables;
  int* bwd_overflow;
  int n_images;
};
#define RS_ROI_BWD_TABLE_BYTES 2944

struct BoxCandParams {
  const float* pred;        // [N][cap][cs]: [0,K] class logits, then 4K deltas
  const float* prop_boxes;  // [N][cap][4]
  const int* prop_count;    // [N]
  int K, cap, cs;
  float wx, wy, ww, wh, scale_clamp, img_h, img_w, score_thresh;
  float* dec_boxes;         // [N][cap][K][4]
  float* dec_scores;        // [N][cap][K]
  float* seg_boxes;         // [N][K][1024][4]
  int* seg_roi;             // [N][K][1024]
  int* seg_count;           // [N][K]
};

struct DetMergeParams {
  const float* dec_boxes;
  const float* dec_scores;
  const int* seg_roi;
  const int* seg_count;
  const uint8_t* keep;      // [N][K][1024]
  int K, cap, dets_per_image;
  float scale_x, scale_y, out_w, out_h;
  float* det_boxes_net;     // [N][D][4] network-input coordinates (mask RoIs)
  float* det_boxes;         // [N][D][4] tile coordinates
  float* det_scores;        // [N][D]
  int* det_classes;         // [N][D]
  int* det_roi;             // optional [N][D]
  int* det_count;           // [N]
};

struct MaskPredictParams {
  const half_t* in;         // [entry][S][S][256]
  const float* w;           // [K][256]
  const float* b;           // [K]
  const int* slot_list;
  const int* det_classes;   // [slots]
  const int* n_entries;
  float* out;               // [slots][S][S]
  int S;
  int f32;                  // 1: fp32 validation mode, `in` is float; 2: split-operand mode, `in` is the hi plane, the lo plane in_lo elements behind
  long long in_lo;
};

struct PasteParams {
  const float* probs;       // [slots][S][S]
  const float* det_boxes;   // [slots][4] tile coordinates
  const int* slot_list;
  const int* n_entries;
  uint8_t* out;             // [slots][out_h][ceil(out_w/8)], bit b of byte = pixel 8*byte+b
  int S, out_h, out_w;
  float threshold;
};

// Detection masks cropped to their boxes for the host (the pasted mask is zero outside its box): per (image, slot) a byte-aligned
// rectangle [x0b, x0b + wbytes) x [y0, y0 + rows) of the bit-packed canvas, stored back to back in (image, slot) order.
struct CropParams {
  const float* det_boxes;     // [n][D][4] tile pixels
  const int* det_count;       // [n]
  const uint8_t* masks;       // [n][D][h][Wb]
  int n, D, h, w, Wb;
  int* rects;                 // [n][D][4]: x0b, y0, wbytes, rows (zeros for empty slots)
  unsigned int* offsets;      // [n][D]
  unsigned long long* total;  // [1] bytes used
  uint8_t* data;
};
int launch_mask_crops(const CropParams& p, hipStream_t s);

int launch_rpn_select(const RpnParams& p, hipStream_t s);
int launch_nms(const NmsParams& p, int segments, hipStream_t s);
int launch_rpn_merge(const RpnMergeParams& p, int N, hipStream_t s);
int launch_roi_align(const RoiAlignParams& p, hipStream_t s);
int launch_roi_align_bwd(const RoiAlignParams& p, hipStream_t s);
int launch_box_candidates(const BoxCandParams& p, int N, hipStream_t s);
int launch_det_merge(const DetMergeParams& p, int N, hipStream_t s);
int launch_det_compact(const int* det_count, int N, int D, int* slot_list, int* total, hipStream_t s);
int launch_mask_predict(const MaskPredictParams& p, int capacity_entries, hipStream_t s);
int launch_mask_sigmoid(const MaskPredictParams& p, int capacity_entries, hipStream_t s);
int launch_paste_masks(const PasteParams& p, int capacity_entries, hipStream_t s);
