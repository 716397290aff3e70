/* rs_engine.h -- C ABI of librs_engine.so, the MI355X (gfx950) Mask R-CNN R50-FPN inference engine.
 *
 * Drop-in boundary.  The reference has no FFI for this path: proj-roadsurf calls
 * the STDL object-detector CLI (R:README.md:77-78), whose make_detections.py does, per tile,
 *     outputs = DefaultPredictor(cfg)(cv2.imread(file))        [EXT d2: engine/defaults.py]
 * configured by R:config/config_obj_detec.yaml:74-90 and R:config/detectron2_config_3bands.yaml.
 * rs_engine_infer() replaces exactly that call (batched): HWC uint8 BGR tiles in, per-tile
 * `Instances` fields out (pred_boxes XYXY in tile pixels, scores, pred_classes, pred_masks).
 * Everything behind it -- resize, normalisation, ResNet-50/FPN, RPN, RoIAlign, box head, NMS, mask
 * head, mask paste -- runs as hand-written HIP kernels; there is no CPU fallback.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns 0 (RS_OK) or
 * a negative error code and sets a message readable through rs_last_error().  An engine belongs
 * to one process and one GPU and is NOT thread-safe.  The caller owns all host buffers; the
 * engine owns its device memory; the weight blob may be freed after rs_engine_create().
 */
#ifndef RS_ENGINE_H
#define RS_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_ABI_VERSION 1
#define RS_SPEC_MAX_LEVELS 5
#define RS_SPEC_MAX_ANCHORS 8
#define RS_NUM_PHASES 3 /* rs_engine_infer_phase: 0 preprocess..RPN proposals, 1 box head..detections, 2 mask head + paste */
#define RS_MASK_SIDE 28 /* 2 * ROI_MASK_HEAD.POOLER_RESOLUTION (R:219) */

/* POD mirror of the detectron2 YAML fields the inference path reads
 * (R:config/detectron2_config_3bands.yaml; line numbers per field). */
typedef struct rs_spec {
  int32_t struct_size;            /* sizeof(rs_spec), ABI check */
  int32_t in_channels;            /* len(MODEL.PIXEL_MEAN) :81-84 (3; 4 = RGB+NIR tiles) */
  int32_t flip_channels;          /* INPUT.FORMAT == "RGB" :26 -> reverse the channel axis of the BGR input */
  float pixel_mean[4];            /* :81-84, applied in model-channel order (the BGR-listed quirk is preserved) */
  float pixel_std[4];             /* :85-88 */
  int32_t min_size_test;          /* :30 */
  int32_t max_size_test;          /* :28 */
  int32_t size_divisibility;      /* 32 (FPN) */
  int32_t res_blocks[4];          /* RESNETS.DEPTH 50 -> 3,4,6,3 :100 */
  int32_t stem_out_channels;      /* :110 */
  int32_t res2_out_channels;      /* :108 */
  int32_t stride_in_1x1;          /* :111 */
  int32_t fpn_out_channels;       /* :69 */
  int32_t num_levels;             /* len(RPN.IN_FEATURES) :231-236 */
  int32_t num_anchors;            /* len(ASPECT_RATIOS) x len(SIZES[l]) :45-56 */
  float cell_anchors[RS_SPEC_MAX_LEVELS][RS_SPEC_MAX_ANCHORS][4]; /* generate_cell_anchors, fp32 */
  float anchor_offset;            /* :50 */
  float rpn_bbox_reg_weights[4];  /* :224-228 */
  int32_t rpn_pre_nms_topk;       /* :249 */
  int32_t rpn_post_nms_topk;      /* :247 */
  float rpn_nms_thresh;           /* :245 */
  float rpn_min_size;             /* :90 */
  int32_t num_classes;            /* :191 (CLI overrides from the COCO categories) */
  float box_reg_weights[4];       /* :160-164 */
  float score_thresh_test;        /* :194 / R:config/config_obj_detec.yaml:90 */
  float nms_thresh_test;          /* :190 */
  int32_t detections_per_image;   /* :321 */
  int32_t box_fc_dim;             /* :167 */
  int32_t box_pooler_resolution;  /* :172 */
  int32_t mask_on;                /* :72 */
  int32_t mask_pooler_resolution; /* :219 */
  int32_t mask_num_conv;          /* :218 */
  int32_t mask_conv_dim;          /* :215 */
  float mask_threshold;           /* detector_postprocess default 0.5 */
  float scale_clamp;              /* Box2BoxTransform: log(1000/16) */
  int32_t precision;              /* 0 = production (fp16 operands, fp32 accumulate on MFMA); 1 = fp32 validation mode:
                                     every conv/linear layer in fp32 on the fp32 matrix cores, needs "<layer>.w32" blob entries;
                                     2 = SPLIT OPERANDS (inference engines): reference-equivalent arithmetic on the fp16 matrix
                                     cores -- every GEMM operand as hi + lo fp16 planes (22 significand bits), three products
                                     (hi.hi, hi.lo, lo.hi) into one fp32 accumulator, activations stored as two planes; needs
                                     "<layer>.ws" (fp16 [2][rows][Kpad]: hi rows then lo rows of the weight scaled by a power
                                     of two per row) and "<layer>.wsi" (fp32 [rows]: the inverse scales) blob entries.  The
                                     reference computes in fp32 (no SOLVER.AMP key, R:config/detectron2_config_3bands.yaml:268-305) */
} rs_spec;

/* Caller-allocated result block for n tiles, D = detections_per_image slots per tile.
 * Entries [0, count[i]) of tile i are valid and sorted by score (descending). Any pointer except
 * `count` may be NULL to skip that field. */
typedef struct rs_dets {
  int32_t* count;     /* [n] */
  float* boxes;       /* [n][D][4]  x1,y1,x2,y2 in tile pixels   (Instances.pred_boxes) */
  float* scores;      /* [n][D]                                  (Instances.scores) */
  int32_t* classes;   /* [n][D]     0-based contiguous class id  (Instances.pred_classes) */
  uint8_t* masks;     /* [n][D][h][ceil(w/8)] bit-packed rows, bit b of a byte = pixel 8*byte+b
                         (Instances.pred_masks, >= mask_threshold) */
  float* mask_probs;  /* [n][D][28][28] sigmoid probabilities before pasting */
} rs_dets;

/* Detection masks cropped to their boxes (a pasted mask is zero outside its box): what the streaming host interface copies
 * instead of the full canvases -- 100 x h x ceil(w/8) bytes per tile (3.3 MB at 512x512, 13 MB at 1024x1024) shrink to the
 * boxes' area.  Crop (i, d) covers byte columns [rects[0], rects[0] + rects[2]) and rows [rects[1], rects[1] + rects[3]) of
 * the bit-packed canvas rs_dets.masks describes, rows stored back to back at data + offsets[i][d]; crops follow each other
 * in (tile, slot) order.  Caller-allocated (pinned memory recommended); capacity n*D*h*ceil(w/8) always suffices. */
typedef struct rs_mask_crops {
  int32_t* rects;      /* [n][D][4]: first byte column, first row, bytes per row, rows; zeros for empty slots */
  uint32_t* offsets;   /* [n][D] */
  uint8_t* data;
  uint64_t capacity;   /* bytes available at data */
  uint64_t used;       /* out (rs_engine_fetch_crops_wait): bytes written */
} rs_mask_crops;

typedef struct rs_engine rs_engine;

/* Build an engine for tiles of one fixed shape (h, w, c) and batches of up to max_batch tiles.
 * `weights` is the blob written by proj_roadsurf_amd.weights.pack_weights() (see "weight blob").
 * `stream` is a hipStream_t to run on, or NULL to let the engine create its own. */
int rs_engine_create(const rs_spec* spec, const void* weights, size_t nbytes, int device_ordinal,
                     int max_batch, int tile_h, int tile_w, int tile_c, void* stream, rs_engine** out);
void rs_engine_destroy(rs_engine* e);

/* DefaultPredictor.__call__ for n tiles: tiles = [n][h][w][c] uint8, channel order as cv2.imread
 * returns it (BGR).  Host buffers; does H2D, the forward, D2H, and waits. */
int rs_engine_infer(rs_engine* e, const uint8_t* tiles_host, int n, rs_dets* out_host);

/* Same forward with tiles already resident in device memory; enqueues on the engine's stream and
 * returns without waiting.  Results stay on the device until rs_engine_fetch(). */
int rs_engine_infer_device(rs_engine* e, const uint8_t* tiles_dev, int n);
int rs_engine_sync(rs_engine* e);
int rs_engine_fetch(rs_engine* e, int n, rs_dets* out_host);

/* The same forward, enqueued one phase at a time (phase 0, 1, 2 in order, same tiles/n for all three) so that a
 * caller driving TWO engines created on one shared `stream` can interleave them: the latency-bound detection glue
 * at the end of phases 0 and 1 runs on an engine-private side stream and is hidden behind the other engine's
 * convolutions, which stay serialised on the shared stream.  Enqueue order per batch k (engine k mod 2):
 *   phase0(k), phase2(k-1), phase1(k)          -- see proj_roadsurf_amd/engine.py:LanePipeline
 * rs_engine_infer_device(e, t, n) == phases 0, 1, 2 back to back.  There is no reference counterpart
 * (DefaultPredictor is synchronous, one image per call: [EXT d2: engine/defaults.py]). */
int rs_engine_infer_phase(rs_engine* e, const uint8_t* tiles_dev, int n, int phase);
int rs_engine_phase_count(void);
/* Diagnostic (tools/parity/lanes_stress3.py): enqueue only the stages whose name contains `substr`, on the engine's stream, on the data of the last forward. */
int rs_debug_run_stages_matching(rs_engine* e, const char* substr, int n);

/* Asynchronous host interface (what a streaming caller -- make_detections.py over a tile list -- uses instead of rs_engine_infer):
 * pinned host memory, the upload enqueued on the engine's stream in front of the forward, and the results copied back on a
 * separate copy stream behind an event, so batch k's device-to-host copy (3.3 MB of packed masks per 512x512 tile) overlaps batch
 * k+1's forward.  With two engines on one shared stream (rs_engine_infer_phase) call rs_engine_fetch_async right after the
 * engine's phase 2 has been enqueued.  Buffers given to upload/fetch_async must come from rs_host_alloc and stay untouched until
 * rs_engine_fetch_wait returns. */
void* rs_host_alloc(size_t nbytes);
void rs_host_free(void* p);
/* Pin / unpin an existing host allocation (e.g. a shared-memory slab that decoder processes fill) so that rs_engine_upload_async
 * copies straight out of it; the caller keeps the bytes unchanged until the forward that consumes them has delivered its results. */
int rs_host_register(void* p, size_t bytes);
int rs_host_unregister(void* p);
int rs_engine_upload_async(rs_engine* e, const uint8_t* tiles_host, int n);
int rs_engine_fetch_async(rs_engine* e, int n, rs_dets* out_host);
int rs_engine_fetch_wait(rs_engine* e);
/* The same two calls with the masks as crops: rs_engine_fetch_crops_async enqueues, on the copy stream behind the forward, the
 * crop planning + compaction kernels and the copies of count / boxes / scores / classes (dets->masks and ->mask_probs are
 * ignored) and of the crop table; rs_engine_fetch_crops_wait waits for them, then copies exactly `used` bytes of crop data
 * and waits for that copy.  The engine's result buffers are free for the next forward after the first step. */
int rs_engine_fetch_crops_async(rs_engine* e, int n, rs_dets* dets_host, rs_mask_crops* crops_host);
int rs_engine_fetch_crops_wait(rs_engine* e, rs_mask_crops* crops_host);

/* Stream the engine launches on (hipStream_t), for event timing by the caller. */
void* rs_engine_stream(rs_engine* e);

/* Per-stage timing with HIP events on the engine's stream.  mode 0 = off; 1 = record + wait per
 * stage (serialises the host, debugging); 2 = record only, events are read back when
 * rs_engine_stage_info() is next called (up to 32 forwards' worth; extra forwards are not timed);
 * 3 = like 2 but only every 4th forward is bracketed (the others replay the captured hipGraph). */
int rs_engine_set_profiling(rs_engine* e, int mode);
int rs_engine_stage_count(rs_engine* e);
/* name_out: >= 96 bytes.  ms_total / calls accumulate since the last rs_engine_set_profiling().
 * flops = algorithmic FLOPs of one call at the last batch size (0 for non-GEMM stages),
 * bytes = algorithmic HBM bytes of one call (inputs read once + outputs written once). */
int rs_engine_stage_info(rs_engine* e, int i, char* name_out, double* ms_total, int* calls, double* flops, double* bytes);
/* Kernel symbol (tile variant) the stage's last call launched, "" for non-GEMM stages. name_out: >= 96 bytes. */
int rs_engine_stage_kernel(rs_engine* e, int i, char* name_out);
/* Tile-variant number of the stage's last call (-2 = not a GEMM stage, -1 = fp32 kernel; numbering: rs_op_conv_variant). */
int rs_engine_stage_variant(rs_engine* e, int i);

/* Intermediate tensors by name (parity tests): device pointer, dtype (1 f16, 2 f32, 3 i32, 4 u8, 5 = two fp16 planes of the
 * given shape back to back, value = plane 0 + plane 1: activations of precision 2),
 * up to 5 dims (dims[ndim..] = 1) and the spatial halo of NHWC activations. */
int rs_engine_tensor(rs_engine* e, const char* name, void** dev_ptr, int* dtype, int* ndim, int64_t dims[5], int* halo);
int rs_engine_tensor_count(rs_engine* e);
int rs_engine_tensor_name(rs_engine* e, int i, char* name_out /* >= 96 bytes */);

/* Network input geometry chosen by ResizeShortestEdge for this tile shape. */
int rs_engine_net_shape(rs_engine* e, int* resized_h, int* resized_w, int* padded_h, int* padded_w);

/* -------- stand-alone operators on caller-owned device memory (unit parity tests) -------- */

/* NHWC fp16 convolution / GEMM with fused epilogue.  in: [n][hi+2*in_halo][wi+2*in_halo][cin],
 * w: [cout_rows][kpad] fp16 (k = (kh,kw,cin), cin fastest, kpad multiple of 64), bias fp32,
 * out: [n][ho+2*out_halo][wo+2*out_halo][cout] fp16, or fp32 when out_f32.
 * residual (optional) has the geometry of out; upsample_add (optional) is
 * [n][ho/2+2*out_halo][wo/2+2*out_halo][cout] and is added at (y/2, x/2).
 * variant: -1 auto, 0 = 128x128 tile, 1 = 256x64, 2 = 256x16 (fp32 out), 3 = 256x128, 4 = 256x256, 7 = 64x128,
 * 8 = 128x64, 9 = 32x128, 10 = 64x256, 12 = conv_deep 256x256 (pixels x channels).  use_glds: 1 = direct
 * global->LDS staging (production), 0 = register staging (cross-check). */
int rs_op_conv2d(const void* in, const void* w, const float* bias, void* out, const void* residual,
                 const void* upsample_add, int n, int hi, int wi, int cin, int in_halo, int kh, int kw,
                 int stride, int pad, int cout, int kpad, int out_halo, int relu, int out_f32, int deconv2x,
                 int variant, int use_glds, void* stream);

/* Fused tail of an identity-shortcut bottleneck block of the 64-wide stage (csrc/bneck_fused.hip): t2 = relu(conv3x3(t1, w2) + b2),
 * out = relu(w3 . t2 + b3 + x), and -- when w1p is given -- t1n = relu(w1 . out + b1), the next block's conv1, in one launch
 * ([EXT d2: modeling/backbone/resnet.py BottleneckBlock.forward]).  All maps NHWC fp16 with a zero halo of 1: t1 / t1n
 * [n][h+2][w+2][64], x / out [n][h+2][w+2][256].  w2 [64][576] (kh, kw, cin); w3p [256][64] and w1p [64][256] with their K columns
 * in the register-chaining order (weights.py _perm_k64); biases fp32.  Projection-shortcut form (first block of the stage): x =
 * NULL and instead x0 [n][h+2][w+2][64], the shortcut's input, with wsc [256][64] (natural K order); b3 then is conv3's plus the
 * shortcut's bias: out = relu(w3 . t2 + wsc . x0 + b3). */
int rs_op_bneck_tail(const void* t1, const void* w2, const float* b2, const void* w3p, const float* b3, const void* x, void* out,
                     const void* w1p, const float* b1, void* t1n, const void* x0, const void* wsc, int n, int h, int w, int width,
                     void* stream);     /* width: 64 (as written above) or 128 (every 64 / 256 / 576 above doubled; no projection form) */

/* The identity-shortcut form of rs_op_bneck_tail in the split-operand precision mode (csrc/bneck_split.hip; rs_spec.precision == 2): every map is
 * a hi plane with its lo plane `*_lo` ELEMENTS behind it (value = hi + lo); every weight is [2 * rows][K] fp16 -- hi rows, then lo rows, of the
 * fp32 weight with each row scaled by a power of two (weights.py split_planes) -- and s2 / s3 / s1 hold the inverse scales per row.  Same shapes,
 * K orders and halo as rs_op_bneck_tail; w1p (with s1, b1, t1n) may be NULL.  Projection-shortcut form (width 64 only): x = NULL and x0
 * [n][h+2][w+2][64] (+ x0_lo) is the shortcut's input; w3p then is [2 * 256][128] -- the K columns of conv3 in the chained order followed by
 * the shortcut's 64 in natural order, ONE scale per row over both (s3) -- and b3 the sum of the two biases:
 * out = relu((w3 . t2 + wsc . x0) * s3 + b3). */
int rs_op_bneck_tail_split(const void* t1, int64_t t1_lo, const void* w2, const float* s2, const float* b2, const void* w3p, const float* s3,
                           const float* b3, const void* x, int64_t x_lo, void* out, int64_t out_lo, const void* w1p, const float* s1,
                           const float* b1, void* t1n, int64_t t1n_lo, const void* x0, int64_t x0_lo, int n, int h, int w, int width,
                           void* stream);

/* Raster voting: the overlay of R:scripts/road_segmentation/determine_class.py:97-120 on the tile grid.
 * det_masks [n_det][h][ceil(w/8)] and label_masks [n_labels][same] are bit-packed device buffers (rs_dets.masks layout; h*ceil(w/8)
 * must be a multiple of 4); inter [n_labels][n_det] receives the pixel count of every (label, detection) intersection, label_area
 * [n_labels] every label's own pixel count (device buffers).  rs_engine_label_overlap runs it on the masks of tile `tile` of the
 * engine's last forward (n_det = D slots; only the first count[tile] columns are meaningful) on the engine's stream. */
int rs_op_mask_overlap(const uint8_t* det_masks, int n_det, const uint8_t* label_masks, int n_labels, int h, int w, int32_t* inter,
                       int32_t* label_area, void* stream);
int rs_engine_label_overlap(rs_engine* e, int tile, const uint8_t* label_masks_dev, int n_labels, int32_t* inter_dev, int32_t* label_area_dev);

/* The tile variant launch_conv would pick for a conv / linear layer of this shape (no launch, no GPU needed): m = batch *
 * Ho * Wo output pixels, k x k taps over cin channels (+ cin2 channels of a second 1x1 K source, 0 = none), cout channels,
 * deconv2x as in rs_op_conv2d.  The choice depends on m, i.e. on the batch size: tests enumerate it per layer and batch
 * (tests/test_host_cpu.py::test_conv_variant_table).  Also returns the number of LDS K-step buffers in *stages_out. */
int rs_op_conv_variant(int m, int cin, int k, int cout, int cin2, int deconv2x, int out_f32, int* stages_out);

/* Diagnostic builds only (csrc compiled with -DRS_CLOCK_PROBE, tools/ubench/clock_probe.py): device buffer of 2 int64 per
 * workgroup that rs_op_conv2d's 256x256 deep-prefetch kernel fills with {shader clocks, 100 MHz ticks} spent in its K loop.
 * No effect in the production build. */
int rs_debug_set_conv_probe(void* buffer);

/* rs_op_conv2d in the split-operand precision mode (rs_spec.precision == 2): `in`, `w`, `out`, `residual`, `upsample_add` point to
 * the hi plane of an fp16 tensor whose lo plane lies `*_lo` ELEMENTS behind it (value = hi + lo); w holds the weight rows scaled
 * by a power of two per row, wscale[rows] the inverse scales.  out_f32 = 1: one fp32 output as in rs_op_conv2d. */
int rs_op_conv2d_split(const void* in, int64_t in_lo, const void* w, int64_t w_lo, const float* wscale, const float* bias, void* out, int64_t out_lo,
                       const void* residual, int64_t res_lo, const void* upsample_add, int64_t up_lo,
                       int n, int hi, int wi, int cin, int in_halo, int kh, int kw, int stride, int pad, int cout, int kpad,
                       int out_halo, int relu, int out_f32, int deconv2x, int variant, void* stream);

/* Bottleneck output with a projection shortcut as ONE GEMM over two activation sources:
 *   out = act( conv_khxkw(in; W[:, :kh*kw*cin]) + conv_1x1_stride2(in2; W[:, kh*kw*cin:]) + bias )
 * i.e. BottleneckBlock.forward's `out = conv3(out); shortcut = self.shortcut(x); out += shortcut; relu`
 * ([EXT d2: modeling/backbone/resnet.py BottleneckBlock.forward]) with both FrozenBN-folded convolutions'
 * weights concatenated along K and their biases added.  in2: [n][h2+2*in2_halo][w2+2*in2_halo][cin2] fp16;
 * output pixel (y, x) reads in2 pixel (y*stride2, x*stride2). */
int rs_op_conv2d_dual(const void* in, const void* in2, const void* w, const float* bias, void* out,
                      int n, int hi, int wi, int cin, int in_halo, int kh, int kw, int stride, int pad,
                      int h2, int w2, int cin2, int in2_halo, int stride2,
                      int cout, int kpad, int out_halo, int relu, int variant, void* stream);

/* Training path: input gradient of rs_op_conv2d's convolution, with the backward epilogue fused:
 *   dx = relu'(mask) * ( conv_transpose(dy, W) + res + res32 + sum2x2(down) )
 * dy: [n][ho+2h][wo+2h][cout] fp16; w_t: the transposed, tap-flipped weight [cin][(kh-1-i, kw-1-j, co)] fp16 (kpad columns);
 * dx/res/mask: [n][hi+2h][wi+2h][cin] fp16, res32 the same geometry in fp32, down: [n][2hi+2h][2wi+2h][cin] fp16 (all
 * optional except dx).  stride 1, or stride s with a 1x1 kernel (STRIDE_IN_1X1, R:config/detectron2_config_3bands.yaml:111):
 * then dx is only written at every s-th pixel and the caller provides zeros elsewhere.  What autograd reaches through
 * cuDNN backward-data + the ReLU / add / nearest-upsample backward nodes ([EXT d2: modeling/backbone/{resnet,fpn}.py]). */
int rs_op_conv2d_dgrad(const void* dy, const void* w_t, void* dx, const void* res, const float* res32, const void* mask,
                       const void* down, int n, int hi, int wi, int cin, int ho, int wo, int cout, int kh, int kw, int stride,
                       int pad, int kpad, int halo, int variant, void* stream);

/* Training path: weight gradient of rs_op_conv2d's convolution (conv_wgrad.hip),
 *   grad[co][(kh,kw,ci)] = scale[co] * sum_{n,y,x} dy[n][y][x][co] * in[n][y*stride+kh-pad][x*stride+kw-pad][ci]
 * -- what autograd computes for Conv2d.weight / Linear.weight ([EXT d2: layers/wrappers.py Conv2d]; with FrozenBN the
 * trainable tensor is the unfolded weight, hence the optional per-channel `scale`).  dy: [n][ho+2*dy_halo][wo+2*dy_halo][cout]
 * fp16 with a zero halo; grad: fp32 [cout][kpad] in the forward weight layout.  splits <= 0: chosen by the library. */
int rs_op_conv2d_wgrad(const void* dy, const void* in, float* grad, const float* scale, int n, int hi, int wi, int cin,
                       int in_halo, int kh, int kw, int stride, int pad, int cout, int kpad, int dy_halo, int splits,
                       void* stream);
/* the same from fp32 operands (dy, x point to float): the reference-precision trainer's conv_wgrad_f32_kernel */
int rs_op_conv2d_wgrad_f32(const void* dy, const void* in, float* grad, const float* scale, int n, int hi, int wi, int cin,
                       int in_halo, int kh, int kw, int stride, int pad, int cout, int kpad, int dy_halo, int splits,
                       void* stream);

/* Greedy NMS over `segments` independent lists of up to 1024 boxes in priority order
 * (torchvision.ops.nms semantics: IoU > thresh suppresses). keep: [segments][cap] 0/1. */
int rs_op_nms(const float* boxes, const int32_t* counts, const uint8_t* valid, uint8_t* keep,
              int segments, int cap, float thresh, void* stream);

/* ROIPooler + ROIAlign(aligned=True, sampling_ratio=0) over up to 4 NHWC fp16 levels (halo 1,
 * 256 channels). rois: [n_rois][4] image coordinates, batch_index = roi / rois_per_image.
 * out: [n_rois][P+2*out_halo][P+2*out_halo][256] fp16; levels_out optional [n_rois]. */
int rs_op_roi_align(const void* const feats[4], const int32_t heights[4], const int32_t widths[4],
                    const float scales[4], int nlevels, const float* rois, int n_rois, int rois_per_image,
                    int P, int out_halo, void* out, int32_t* levels_out, void* stream);

/* Training path: adjoint of rs_op_roi_align.  dout: gradient of the pooled output, same layout as `out` above;
 * dfeats[l]: fp32 gradient maps with the geometry of feats[l] ([N][H_l+2][W_l+2][256]), ACCUMULATED into.  Owner-computes: one workgroup
 * per 8 x 8-cell region adds the RoIs that reach it in entry order (fixed summation order, reproducible bits); RoIs whose window exceeds
 * the weight tables go through float atomics, as every RoI does in torchvision's roi_align_backward_kernel
 * ([EXT tv: csrc/ops/cuda/roi_align_kernel.cu]).  Synchronises `stream` (its table workspace lives for the call). */
int rs_op_roi_align_bwd(float* const dfeats[4], const int32_t heights[4], const int32_t widths[4], const float scales[4],
                        int nlevels, const float* rois, int n_rois, int rois_per_image, int P, int out_halo, const void* dout,
                        void* stream);

/* Training path, losses: each writes the gradient of the total loss w.r.t. its inputs (fp16, times loss_scale) and adds the
 * loss values to loss_out (fp32, for logging).  Label assignment / sampling happen before (oracle/train_oracle.py restates
 * them; their kernels are the next step).
 *  rs_op_rpn_loss : RPN.losses for ONE feature level.  head/dhead: [n][hw][cs] ([0,A) logits, [A,5A) deltas); labels:
 *                   [n][total_anchors] in {1,0,-1} after subsampling; anchors [total_anchors][4]; matched_gt
 *                   [n][total_anchors][4]; this level's anchors start at level_off.  loss_out[0] += loss_rpn_cls,
 *                   loss_out[1] += loss_rpn_loc, both / normalizer (= RPN.BATCH_SIZE_PER_IMAGE * n, R:223).
 *  rs_op_box_loss : FastRCNNOutputLayers.losses.  pred/dpred [n_rois][cs] ([0,K] logits, K = background, then 4K deltas);
 *                   gt_classes in {0..K, -1 = empty slot}; n_valid = number of sampled RoIs.  loss_out[0] += loss_cls,
 *                   loss_out[1] += loss_box_reg.
 *  rs_op_mask_loss: mask_rcnn_loss.  logits/dlogits [n_masks][side*side][cs], targets [n_masks][side*side] 0/1.
 *  rs_op_sgd_momentum: one torch.optim.SGD step on a flat fp32 tensor (grad is divided by the loss scale first).
 *  rs_op_fold_weights: fp32 master weight [cout][kpad] -> fp16 forward weight (same layout, optional per-channel scale
 *                   folded) and its transposed, tap-flipped copy [cin][kpad_t] for rs_op_conv2d_dgrad (w_bwd may be NULL). */
/* Training path, label assignment.
 *  rs_op_match    : Matcher on pairwise IoU ([EXT d2: modeling/matcher.py, structures/boxes.py]).  boxes: [n_boxes][4] shared
 *                   by all images (anchors) or [n_images][n_boxes][4] (per_image_boxes = 1, box_count[n] valid rows);
 *                   gt [n_images][gt_cap][4] (gt_cap <= 256).  matched: index of the best gt (lowest index on ties);
 *                   labels: lbl_lo / lbl_mid / lbl_hi for best IoU < t_lo / in [t_lo, t_hi) / >= t_hi (RPN: 0,-1,1 at
 *                   0.3/0.7, R:237-243; ROI heads: 0,1,1 at 0.5/0.5, R:184-188); allow_low_quality: boxes whose IoU with a
 *                   gt equals that gt's highest IoU become 1 (RPN only).
 *  rs_op_subsample: subsample_labels ([EXT d2: modeling/sampling.py]).  Positives: label != -1 && != bg_label.  rpn_mode 1
 *                   rewrites labels in place (1 / 0 / -1); otherwise `sampled` [n_images][num_samples] receives the sampled
 *                   indices (positives first, ascending inside each group, -1 padded).  The random choice is a keyed hash of
 *                   (seed, image, index): deterministic for a seed, a different uniform sample per seed. */
int rs_op_match(const float* boxes, int per_image_boxes, const int32_t* box_count, const float* gt, const int32_t* gt_count,
                int32_t* matched, int32_t* labels, float* best_iou, int n_images, int n_boxes, int gt_cap, float t_lo, float t_hi,
                int lbl_lo, int lbl_mid, int lbl_hi, int allow_low_quality, void* stream);
int rs_op_subsample(int32_t* labels, int32_t* sampled, int32_t* sampled_count, int n_images, int n, int num_samples,
                    float positive_fraction, int bg_label, int rpn_mode, uint32_t seed, void* stream);
int rs_op_rpn_loss(const float* head, void* dhead, const int32_t* labels, const float* anchors, const float* matched_gt,
                   float* loss_out, int n, int hw, int num_anchors, int cs, int level_off, int total_anchors, float normalizer,
                   float loss_scale, void* stream);
int rs_op_box_loss(const float* pred, void* dpred, const int32_t* gt_classes, const float* proposals, const float* gt_boxes,
                   float* loss_out, int n_rois, int num_classes, int cs, float n_valid, const float reg_weights[4], float loss_scale,
                   void* stream);
int rs_op_mask_loss(const float* logits, void* dlogits, const uint8_t* targets, const int32_t* gt_classes, float* loss_out, int n_masks,
                    int side, int cs, float loss_scale, void* stream);
int rs_op_sgd_momentum(float* w, float* momentum_buf, const float* grad, int64_t n, float lr, float momentum, float weight_decay,
                       float inv_loss_scale, int first_step, void* stream);
int rs_op_fold_weights(const float* w32, const float* scale, void* w_fwd, void* w_bwd, int cout, int cin, int kh, int kw, int kpad,
                       int kpad_t, void* stream);

/* ------------------------------------------------------------------ training engine
 * Owns a forward engine (same blob, packed with train=True: fp32 master weights `<layer>.m32`, FrozenBN scales `<layer>.s`),
 * one flat fp32 master / gradient / momentum buffer in the forward GEMM layout, a gradient buffer per activation, and the
 * backward stage list.  One step = rs_trainer_set_targets, rs_trainer_forward_trunk (preprocess .. FPN), rs_trainer_rpn_forward,
 * rs_trainer_roi_step (box head: sampling, losses, backward), rs_trainer_mask_forward / _mask_backward, rs_trainer_rpn_step
 * (RPN losses + backward), rs_trainer_backward_trunk (FPN + res5..res3 from the gradients of p2..p6), rs_trainer_apply_sgd
 * (torch.optim.SGD step on every trainable tensor + refold of the fp16 operands): what detectron2 reaches through autograd +
 * SimpleTrainer.run_step
 * ([EXT d2: engine/train_loop.py]).  Tensors: "d:<forward tensor>" activation gradients (fp16, times loss_scale),
 * "g:<layer>.w|.b" gradients and "m:<layer>.w|.b" master weights (fp32, forward layout).
 * rs_spec.precision selects the arithmetic of the whole step: 1 = REFERENCE PRECISION -- activations, activation gradients and GEMM
 * operands fp32 on v_mfma_f32_16x16x4_f32 (what detectron2's SimpleTrainer computes for the reference's YAML, which has no SOLVER.AMP
 * key, R:config/detectron2_config_3bands.yaml:268-305; "d:" tensors are then fp32, blob packed with the `.w32` operands, loss_scale 1);
 * 0 = fp16 operands with fp32 master weights and a loss scale (AMPTrainer's place). */
typedef struct rs_trainer rs_trainer;
int rs_trainer_create(const rs_spec* spec, const void* weights, size_t nbytes, int device_ordinal, int batch, int tile_h,
                      int tile_w, int tile_c, float loss_scale, rs_trainer** out);
void rs_trainer_destroy(rs_trainer* t);
rs_engine* rs_trainer_engine(rs_trainer* t);
int rs_trainer_forward_trunk(rs_trainer* t, const uint8_t* tiles_dev, int n);
int rs_trainer_backward_trunk(rs_trainer* t, int n);
/* Batches of mixed sizes (INPUT.MIN_SIZE_TRAIN with MIN_SIZE_TRAIN_SAMPLING "choice" drawn per image,
 * R:config/detectron2_config_3bands.yaml:31-38; [EXT d2: data/dataset_mapper.py + structures/image_list.py ImageList.from_tensors]):
 * image i of the coming batches is resized to new_h[i] x new_w[i] (<= the trainer's network-input size, which is the canvas = the
 * largest size of the batch), the canvas is zero beyond it, and its proposals are clipped to that size.  n = 0 restores one size
 * for all.  Ground truth passed to rs_trainer_set_targets is in each image's own resized pixels. */
int rs_trainer_set_image_sizes(rs_trainer* t, const int32_t* new_h, const int32_t* new_w, int n);
/* Ground truth of the batch (host pointers): boxes in network-input pixels [n][cap][4], classes [n][cap], counts [n]. */
int rs_trainer_set_targets(rs_trainer* t, const float* gt_boxes, const int32_t* gt_classes, const int32_t* gt_count, int n, int cap);
/* RPN of the training forward + RPN.losses + its backward: heads on the current FPN maps, anchor Matcher (0.3/0.7, low-quality
 * matches) and subsample_labels (256 @ 0.5) keyed by `seed`, loss_rpn_cls / loss_rpn_loc into tensor "losses"[0..1], gradients of
 * the shared head accumulated over the levels, d:p2..d:p6 overwritten with the head's input gradient.  external_labels = 1 keeps
 * the caller's "rpn_labels" / "rpn_matched" (parity tests feed the oracle's sample). */
int rs_trainer_rpn_step(rs_trainer* t, int n, uint32_t seed, int external_labels);
/* Optional: the RPN's targets of the coming step (anchor Matcher + label subsampling: functions of the ground truth and the seed only) computed
 * ahead on the trainer's side stream.  Call after rs_trainer_set_targets, before rs_trainer_forward_trunk; the rs_trainer_rpn_step of the same
 * seed then only waits for them.  Same kernels and seed as inside the step: identical labels. */
int rs_trainer_rpn_targets_async(rs_trainer* t, int n, uint32_t seed);
/* RPN head forward alone (rs_trainer_rpn_step runs it itself), then the RoI box head of the training step:
 * RPN proposals in training mode (PRE_NMS_TOPK_TRAIN 2000 per level, NMS 0.7, POST_NMS_TOPK_TRAIN 1000 per image: R:245-250;
 * rs_trainer_set_rpn_topk overrides) + gt boxes (R:193), Matcher at 0.5, subsample_labels (1024 @ 0.25, R:178,192), box-head forward on the sample,
 * loss_cls / loss_box_reg into "losses"[2..3], backward of predictor/fc2/fc1 and RoIAlign into "d32:p2".."d32:p5".  Call order of a
 * full step: forward_trunk, rpn_forward, roi_step, rpn_step, backward_trunk, apply_sgd. */
int rs_trainer_rpn_forward(rs_trainer* t, int n);
int rs_trainer_roi_step(rs_trainer* t, int n, uint32_t seed);
/* Mask head of the training step (after rs_trainer_roi_step, before rs_trainer_rpn_step): forward on the sampled foreground RoIs
 * -- entry e runs over images, then over each image's sampled foreground in order ("mask_slots", "mask_total") --; then, with the
 * host-rasterised gt masks targets[n_entries][28*28] (rs_rasterize_polygons_within_box on "roi_boxes" / "roi_gt_index"),
 * mask_rcnn_loss -> "losses"[4] and the backward of predictor / deconv / 4 convs / RoIAlign into "d32:p2".."d32:p5".  detectron2
 * rasterises the polygons on the host at the same point ([EXT d2: structures/masks.py PolygonMasks.crop_and_resize]). */
int rs_trainer_mask_forward(rs_trainer* t, int n);
int rs_trainer_mask_backward(rs_trainer* t, int n, const uint8_t* targets_host, int n_entries);
/* Read-back of the RoIs sampled by the current rs_trainer_roi_step for the host-side mask targets: boxes [n][1024][4], gt index
 * [n][1024], counts [n][2] (foreground, total).  Waits for the sampling only (own copy stream, event recorded in roi_step): the
 * box head and an already enqueued rs_trainer_mask_forward keep running while the host rasterises. */
int rs_trainer_fetch_rois(rs_trainer* t, int n, float* boxes_host, int32_t* gt_index_host, int32_t* counts_host);
int rs_trainer_set_rpn_topk(rs_trainer* t, int pre_nms_topk_train, int post_nms_topk_train);
/* Sampler sizes (defaults = the reference YAML: 256 @ 0.5 anchors, 1024 @ 0.25 RoIs per image). */
int rs_trainer_set_sampling(rs_trainer* t, int rpn_batch, float rpn_positive_fraction, int roi_batch, float roi_positive_fraction);
/* SGD with momentum on the flat buffers + refold.  The gradient is first checked for inf / nan (fp16 loss scale too large for
 * this batch): then the step is skipped on the device -- weights and momentum unchanged -- and the tensor "grad_overflow" [1]
 * reads 1; lower the scale with rs_trainer_set_loss_scale before the next forward (GradScaler semantics, no host sync here). */
int rs_trainer_apply_sgd(rs_trainer* t, float lr, float momentum, float weight_decay);
int rs_trainer_set_loss_scale(rs_trainer* t, float loss_scale);
int rs_trainer_sync(rs_trainer* t);
int rs_trainer_tensor(rs_trainer* t, const char* name, void** dev_ptr, int* dtype, int* ndim, int64_t dims[5], int* halo);
int rs_trainer_tensor_count(rs_trainer* t);
int rs_trainer_tensor_name(rs_trainer* t, int i, char* name_out);
/* Multi-scale training (INPUT.MIN_SIZE_TRAIN sampling "choice", R:31-38): build one trainer per network-input size (same
 * weights, spec.min_size_test = that size) and, when the size of the next batch differs from the previous one's, carry the
 * optimiser state over: master weights + momentum are copied device to device and dst's fp16 operands refolded. */
int rs_trainer_copy_state(rs_trainer* dst, rs_trainer* src);
/* Data parallel: every rank all-reduces (SUM) the flat gradient buffer (rs_trainer_grad_buffer, rs_trainer_param_count floats,
 * device memory) and sets the divisor to the world size, which rs_trainer_apply_sgd applies together with the loss scale
 * (DistributedDataParallel's gradient averaging: [EXT d2: engine/defaults.py create_ddp_model]). */
int rs_trainer_set_grad_divisor(rs_trainer* t, float divisor);
void* rs_trainer_master_buffer(rs_trainer* t);
int64_t rs_trainer_param_count(rs_trainer* t);
void* rs_trainer_grad_buffer(rs_trainer* t);
/* Bucketed, overlapped gradient all-reduce (what DistributedDataParallel's bucketing does with autograd hooks).  The flat
 * gradient buffer is cut into contiguous buckets listed in the order a step COMPLETES them: "heads" (box / mask / RPN heads,
 * complete after rs_trainer_rpn_step), "fpn", "res5", "res4", "res3" (completed one after the other inside
 * rs_trainer_backward_trunk).  rs_trainer_bucket_info: offset / count in floats from rs_trainer_grad_buffer.
 * rs_trainer_bucket_wait(i, stream): work enqueued on `stream` afterwards (the collective of bucket i) runs behind the
 * kernels that produce bucket i's gradients of the step enqueued so far -- a device-side wait, the caller does not block, so
 * bucket i's all-reduce overlaps the rest of the backward pass.  rs_trainer_bucket_sync(i): the host-side form (collectives
 * through host memory).  rs_trainer_wait_stream(stream): the trainer's own stream waits for everything enqueued on `stream` so
 * far; call it after the last collective and before rs_trainer_apply_sgd. */
/* Per-stage timing of the training step (bench.py --train): HIP events around every stage -- the forward stages of the training
 * step, then the box / mask / RPN / trunk backward lists -- on the stream the stage runs on, no host wait.  on = 1 clears the
 * accumulators and starts; rs_trainer_stage_count() resolves what has been recorded (it synchronises the trainer's streams).
 * flops = algorithmic FLOP of the stage's last execution (0 for non-GEMM stages); on_side_stream = 1 for weight / bias gradients
 * and RoIAlign backward, which run next to the input-gradient chain. */
int rs_trainer_set_profiling(rs_trainer* t, int on);
int rs_trainer_stage_count(rs_trainer* t);
int rs_trainer_stage_info(rs_trainer* t, int i, char* name_out /* >= 96 bytes */, double* ms_total, int* calls, double* flops, int* on_side_stream);
int rs_trainer_bucket_count(rs_trainer* t);
int rs_trainer_bucket_info(rs_trainer* t, int i, char* name_out /* >= 96 bytes */, int64_t* offset, int64_t* count);
int rs_trainer_bucket_wait(rs_trainer* t, int i, void* stream);
int rs_trainer_bucket_sync(rs_trainer* t, int i);
int rs_trainer_wait_stream(rs_trainer* t, void* stream);

/* -------- host-only helpers (no GPU needed) -------- */
/* detectron2 ResizeShortestEdge.get_output_shape. */
void rs_resize_shape(int h, int w, int short_edge, int max_size, int* new_h, int* new_w);
/* Pillow bilinear coefficient tables for in_size -> out_size: bounds[out][2] = (first, count),
 * coeffs[out][ksize] 22-bit fixed point.  Returns ksize; pass NULL pointers to query it. */
int rs_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* coeffs);

int rs_memcpy_d2h(void* dst_host, const void* src_dev, size_t nbytes);
/* out[i] = a[i] / b[i] (device pointers) through csrc/common.h rs_fdiv, the division every kernel of the library uses instead of the compiler's
 * v_div_scale / v_div_fmas / v_div_fixup sequence; exported for its test. */
int rs_op_fdiv(const float* a, const float* b, float* out, int64_t n, void* stream);
int rs_memcpy_h2d(void* dst_dev, const void* src_host, size_t nbytes);
const char* rs_last_error(void);
int rs_abi_version(void);

/* Weight blob: little-endian; header {u32 magic 'RSEW', u32 version, u32 n, u32 data_offset};
 * n entries {char name[96]; u32 dtype; u32 ndim; u64 dims[4]; u64 offset; u64 nbytes}; data
 * 256-byte aligned.  Entries "<layer>.w" (fp16 [Cout_pad][Kpad], FrozenBN folded) and
 * "<layer>.b" (fp32) for every conv/linear layer, detectron2 layer names. */

/* ------------------------------------------------------------------ detections -> polygons (host code)
 * What the object-detector's detectron2dets_to_features does per instance after the predictor returns
 * ([EXT od: helpers/detectron2.py]; R:config/config_obj_detec.yaml:87-89): rasterio.features.shapes on the
 * instance mask (4-connected regions, rings along pixel edges, holes) and Ramer-Douglas-Peucker (epsilon in
 * pixels; <= 0 disables).  masks = rs_dets.masks layout, [n][h][(w+7)/8] bit-packed LSB first.  Instances are
 * spread over `threads` host threads (0 = all cores).  The result is a flat ragged structure:
 *   inst_poly_count[n]      polygons of every instance
 *   poly_ring_count[np]     rings of every polygon (exterior first, then holes)
 *   ring_len[nr]            vertices of every ring (closed: first == last)
 *   xy[2*nv]                pixel-corner coordinates (x = column, y = row), float64
 * Python restatement with identical vertex output: proj_roadsurf_amd/vectorize.py. */
typedef struct rs_vec_result rs_vec_result;
rs_vec_result* rs_vectorize_masks(const uint8_t* masks, int n, int h, int w, double rdp_epsilon, int threads);
/* The same for masks that arrive as crops (rs_mask_crops): rects [n][4], offsets [n] into data. */
rs_vec_result* rs_vectorize_mask_crops(const uint8_t* data, const int32_t* rects, const uint32_t* offsets, int n, int h, int w,
                                       double rdp_epsilon, int threads);
void rs_vec_counts(const rs_vec_result* r, int64_t* n_instances, int64_t* n_polygons, int64_t* n_rings, int64_t* n_vertices);
int rs_vec_copy(const rs_vec_result* r, int32_t* inst_poly_count, int32_t* poly_ring_count, int32_t* ring_len, double* xy);
void rs_vec_free(rs_vec_result* r);
/* GeoPackage geometry blobs ('GP' header + little-endian WKB Polygon, OGC 12-128r15) of every polygon of a result, coordinates
 * georeferenced per instance: X = xform[i][0] + x*xform[i][2], Y = xform[i][1] - y*xform[i][3] (xform NULL = pixel coordinates,
 * Y = y) -- what the reference writes through geopandas.to_file(driver="GPKG") (R:config/config_obj_detec.yaml:100-103).
 * out NULL: returns the bytes needed; else fills out, offsets[n_polygons+1] and bbox[minx,miny,maxx,maxy]. */
int64_t rs_vec_gpkg_blobs(const rs_vec_result* r, const double* xform, int32_t srs_id, uint8_t* out, int64_t out_cap,
                          int64_t* offsets, double bbox[4]);

/* Training targets of the mask head (host code): the polygons of ONE ground-truth instance cropped to `box` and rasterised at
 * mask_size x mask_size -- PolygonMasks.crop_and_resize ([EXT d2: structures/masks.py rasterize_polygons_within_box]; rasteriser =
 * pycocotools' rleFrPoly restated, parity unpinned).  polys: concatenated [x0,y0,x1,y1,...] of the polygons, poly_len[i] doubles
 * each; out: [mask_size][mask_size] 0/1. */
int rs_rasterize_polygons_within_box(const double* polys, const int32_t* poly_len, int n_polys, const double box[4], int mask_size,
                                     uint8_t* out);
/* The same for every sampled foreground RoI of a training step in one call (host threads over the entries; threads <= 0: up
 * to 16): instance g owns polygons inst_first[g] .. inst_first[g+1]-1, polygon q = poly_len[q] doubles at polys + poly_off[q];
 * entry e rasterises instance entry_inst[e] inside boxes[e] (x1,y1,x2,y2 as read back from "roi_boxes") -> out[e][S*S]. */
int rs_rasterize_entries(const double* polys, const int64_t* poly_off, const int32_t* poly_len, const int32_t* inst_first, int n_inst,
                         const int32_t* entry_inst, const float* boxes, int n_entries, int mask_size, uint8_t* out, int threads);

#ifdef __cplusplus
}
#endif
#endif /* RS_ENGINE_H */
