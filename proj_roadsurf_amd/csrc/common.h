// Shared declarations for the gfx950 engine (device + host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 half_t;
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RS_OK 0
#define RS_ERR_ARG -1
#define RS_ERR_HIP -2
#define RS_ERR_BLOB -3
#define RS_ERR_UNSUPPORTED -4

// Set by the failing call; read through rs_last_error().
void rs_set_error(const char* fmt, ...);

#define RS_HIP(expr)                                                                       \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      rs_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));   \
      return RS_ERR_HIP;                                                                   \
    }                                                                                      \
  } while (0)

#define RS_CHECK(cond, code, ...)  \
  do {                             \
    if (!(cond)) {                 \
      rs_set_error(__VA_ARGS__);   \
      return (code);               \
    }                              \
  } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// IEEE-correct fp32 a / b WITHOUT v_div_scale_f32 / v_div_fmas_f32 / v_div_fixup_f32.  The compiler's expansion of `/` returns wrong quotients in a wave whose
// SIMD is shared with waves of ANOTHER kernel that issue MFMAs (found in round 4 with two engines on independent streams: the paste kernel's masks; reproduced
// stand-alone by tools/ubench/coexec_probe.hip, where the same arithmetic with this function is bit-stable and bit-identical to the undisturbed `/`).  Same iteration as
// the compiler's (reciprocal refined once, quotient refined twice with the exact remainder), without its exponent scaling -- exact for the O(1) magnitudes of box
// coordinates, probabilities and losses (tests/test_gpu_conv.py: bit-identical to IEEE division on 8 million pairs over 1e-12 .. 1e12); zero / infinite divisors and zero
// (its sign) / infinite / NaN dividends get IEEE's answer from the plain product with the hardware reciprocal.
// Every fp32 division of device code that can run beside another stream's convolutions goes through it (detection glue, pre-processing, losses).
#if defined(__HIPCC__)
__device__ __forceinline__ float rs_fdiv(float a, float b) {
  const float y0 = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, y0, 1.0f);
  const float y = __builtin_fmaf(e, y0, y0);
  const float q0 = a * y;
  const float r0 = __builtin_fmaf(-b, q0, a);
  const float q1 = __builtin_fmaf(r0, y, q0);
  const float r1 = __builtin_fmaf(-b, q1, a);
  const float q = __builtin_fmaf(r1, y, q1);
  const bool special = !(__builtin_fabsf(b) > 0.f && __builtin_fabsf(b) < __builtin_inff() && __builtin_fabsf(a) > 0.f && __builtin_fabsf(a) < __builtin_inff());
  return special ? a * y0 : q;
}
#endif

// Debug / experiment switches.  Every RS_* environment variable the library understands is read in ONE place
// (engine.hip: rs_debug_reload(), run by rs_engine_create / rs_trainer_create and on first use); production sets none
// of them and the defaults below ARE the shipped configuration.  tools/ubench/* and a few tests flip them.
struct RsDebug {
  int conv_single_stage_nk = 4;   // RS_CONV_SINGLE_STAGE_NK  K steps up to which a conv runs with ONE LDS buffer
  int conv_persist = 0;           // RS_CONV_PERSIST          1: shallow-K layers persistent, 2: all (measured: not faster)
  int conv_tuned = 1;             // RS_CONV_TUNED            0: first-round tile rule (128x128 / 256x64 only)
  int conv_deep = 1;              // RS_CONV_DEEP             0: conv_igemm 256x256 instead of conv_deep
  int conv_wide_px = 128;         // RS_CONV_WIDE_PX          pixels of the all-256-channel tile of the HBM-bound 1x1 layers: 128 (variant 14) or 64 (10)
  int wreg_dbg = 0;               // RS_WREG_DBG              conv_wreg timing ablations (bit 0: no epilogue / stores, bit 2: no operand loads; results wrong)
  int wreg_waves = 2;             // RS_WREG_WAVES            conv_wreg form: 2 = 32-pixel tiles, two workgroups of four waves per CU (ships); 4 / 8 = 64-pixel tiles, one workgroup of four / eight waves
  int conv_wreg = 1;              // RS_CONV_WREG             0: never the persistent register-weight kernel (variant 22, conv_wreg.hip)
  int stem_small_tile = 1;        // RS_STEM_SMALL_TILE       0: 256x64 stem tile
  int deep_dbg = 0;               // RS_DEEP_DBG              -DRS_DEEP_CEILING builds only
  int deconv_variant = 22;        // RS_DECONV_VARIANT        kernel of the fused deconv + predictor: conv_wreg (22), or conv_igemm's 128x256 (14), 64x256 (10), 128x128 (0) tile
  int fuse_mask_predictor = 1;    // RS_FUSE_MASK_PREDICTOR
  int side_stream = 1;            // RS_SIDE_STREAM           detection glue on a side stream
  int narrow_roialign = 0;        // RS_NARROW_ROIALIGN
  int use_glds = 1;               // RS_USE_GLDS              0: register staging instead of LDS-DMA
  int fuse_shortcut = 1;          // RS_FUSE_SHORTCUT
  int fuse_stem = 1;              // RS_FUSE_STEM             stem conv + ReLU + max-pool in one launch (stem_fused.hip)
  int fuse_bneck = 1;             // RS_FUSE_BNECK            conv2 + conv3 + next conv1 of the res2 identity blocks in one launch
  int deep_tile_px = 1;           // RS_DEEP_TILE_PX          conv_deep with 160 / 192 / 224-pixel tiles where 256 fill the CUs badly (variants 15-17)
  int deep_tail = 1;              // RS_DEEP_TAIL             conv_deep: split the tiles of a last round that fills at most half the chip
  int fuse_rpn_heads = 1;         // RS_FUSE_RPN_HEADS        objectness + delta heads inside the epilogue of the merged RPN 3x3 launch
  int merge_levels = 1;           // RS_MERGE_LEVELS          FPN output convs of all levels / the RPN 3x3 over all levels as one launch each
  int graph_small = 1;            // RS_GRAPH_SMALL           forwards of ONE tile replay a hipGraph (the reference's predictor(im) loop: launch gaps are 5 % of its latency)
  int use_graph = 0;              // RS_USE_GRAPH
  int train_roi_side = -1;        // RS_TRAIN_ROI_SIDE        -1: trainer default
  int train_side = -1;            // RS_TRAIN_SIDE            -1: trainer default (on)
  int wgrad_target = 512;         // RS_WGRAD_TARGET
  int wgrad_cb = 0;               // RS_WGRAD_CB
  int select_debug = 0;           // RS_SELECT_DEBUG
  int nms_debug = 0;              // RS_NMS_DEBUG
  int roi_window = 1;             // RS_ROI_WINDOW
  int roi_bwd_atomic = 0;         // RS_ROI_BWD_ATOMIC        1: RoIAlign backward by float atomics for every RoI (rounds 1-2) instead of owner-computes regions
  int roi_order = 1;              // RS_ROI_ORDER             box.roi_align visits the proposals sorted by (level, row, column) instead of by score
};
const RsDebug& rs_debug();
void rs_debug_reload();

// ------------------------------------------------------------------ conv / GEMM
// Activations are NHWC fp16 with a zero halo: a tensor of logical size (N,H,W,C) is stored as
// [N][H+2*pad][W+2*pad][Cs] and only the interior is ever written, so 3x3/7x7 convolutions need
// no bounds checks and the loader is a pure strided gather.
// One map of a multi-map launch (conv_deep only): the FPN output convolutions of all levels, or the shared RPN 3x3 over p2..p6, as ONE
// grid -- same Cin / Cout / kernel size / channel pitches / halos, own geometry, tensors and weights per map.  Tiles never straddle maps.
struct ConvSeg {
  const half_t* in;
  const half_t* w;
  const float* bias;
  void* out;
  float* head_out;   // fused head (ConvParams::head_w): [N][Ho][Wo][16] fp32 of this map
  int Ho, Wo, in_Hp, in_Wp, out_Hp, out_Wp;
  int M;         // rows of this map (set per launch: images * Ho * Wo)
  int tile0;     // first logical tile of this map (set per launch)
  long long in_lo, out_lo;   // split-operand mode (ConvParams::split): element offsets of the lo planes of this map's input / output
  const float* wscale;       // ... and the per-row weight descale of this map's filter
};
#define RS_MAX_SEGS 5

struct ConvParams {
  const half_t* in;
  const half_t* w;      // [Cout_pad][Kpad] fp16, K = (kh,kw,cin) with cin fastest
  const float* bias;    // [Cout_pad]
  void* out;            // fp16 (or fp32 if out_f32)
  const half_t* res;    // residual, same geometry as out (nullptr = none)
  const half_t* up;     // coarser map added at (y/2,x/2) (FPN top-down), nullptr = none
  const int* m_count;   // optional device counter: rows = min(M, *m_count * m_mul)
  const int* koff;      // SMALLC only: element offset of every 8-element K chunk
  int m_mul;
  int M;                // N*Ho*Wo
  int Ho, Wo;
  int in_Hp, in_Wp, in_Cs, in_off;   // in_off = in_pad - conv_pad
  int stride;
  int KH, KW, Cin;
  int Kpad;
  int Cout;             // channels actually stored
  int out_Hp, out_Wp, out_Cs, out_pad;
  int up_Hp, up_Wp, up_Cs, up_pad;
  int relu;
  int mode;             // 0 = conv, 1 = 2x2 stride-2 transposed conv as 4 GEMMs + pixel shuffle,
                        // 2 = mode 1 + fused mask predictor: out is not written, per output pixel the dot product
                        //     of relu(deconv) with dot_w[class] is accumulated into dot_out (see conv_igemm.hip)
  int out_f32;
  const float* dot_w;   // mode 2: [K][Cout] predictor weights (fp32)
  const int* dot_cls;   // mode 2: [slots] predicted class
  const int* dot_slot;  // mode 2: [entries] entry -> slot
  float* dot_out;       // mode 2: [slots][2*Ho][2*Wo] logits (without bias), zeroed by the caller
  int dot_k;            // mode 2: rows of dot_w (classes); 0 = not given (conv_wreg.hip then leaves the layer to conv_igemm)
  // Second activation source appended along K (bottleneck conv3 + projection shortcut in ONE GEMM:
  //   out = relu([W3 | Wsc] * [conv2 out ; block input(stride s)] + b3 + bsc)): after the KH*KW*Cin/64 steps of
  // `in`, Cin2/64 more K steps read pixel (y*stride2, x*stride2) of `in2` (1x1 taps).  Weight row = [K of in | Cin2].
  const half_t* in2;    // nullptr = single source
  int in2_Hp, in2_Wp, in2_Cs, in2_off, stride2, Cin2;
  // ---- training-path epilogue options (input-gradient convolutions, conv_igemm.hip only) ----
  const half_t* down;   // finer map (2Ho x 2Wo, out channel count) whose 2x2 sums are added: backward of the FPN top-down
                        // nearest upsample; geometry down_Hp/down_Wp/down_Cs/down_pad
  int down_Hp, down_Wp, down_Cs, down_pad;
  const float* res32;   // fp32 addend with the OUTPUT's geometry (RoIAlign-backward scatter target), nullptr = none
  const half_t* mask;   // ReLU backward: output is zeroed where mask (output geometry, the saved forward activation) <= 0;
                        // applied after bias/res/up/down/res32, before `relu`
  int out_stride;       // 0/1 = dense; s > 1: output pixel (y, x) is stored at (y*s, x*s) of the out/res/res32/mask maps
                        // (input gradient of a stride-s 1x1 convolution; the other positions are the caller's zeros)
  long long* probe;     // diagnostic builds only (-DRS_CLOCK_PROBE): per workgroup {shader clocks, 100 MHz ticks} around the K loop
  int persist;          // >0: persistent launch with this many workgroups per CU; -1: per-variant default; 0: one per tile
  int stages;           // LDS K-step buffers: 0/2 = double buffered, 1 = single (set by launch_conv for shallow K)
  // conv_deep, Cout == 256: a 16-row 1x1 head applied to relu(conv + bias) inside the epilogue (RPN objectness + anchor deltas); `out` is
  // then NOT written.  head_w: [16][256] fp16 with its K columns in the register-chaining order (weights.py _perm_k64), head_b [16].
  const half_t* head_w;
  const float* head_b;
  float* head_out;      // single-map launch: [N][Ho][Wo][16] fp32
  int nseg;             // > 0: multi-map launch (launch_conv_deep_multi); in / w / bias / out / geometry / M come from seg[]
  int seg_tiles;        // total logical tiles of all maps
  int tail_tiles;       // conv_deep: this many of the LAST logical tiles run as two 128-pixel tiles each (set by the launcher's tail rule)
  // ---- split-operand mode (rs_spec.precision == 2): every fp16 tensor is TWO planes, x = hi + lo with
  // hi = fp16(x), lo = fp16(x - hi) (22 significand bits), the lo plane `*_lo` ELEMENTS behind the hi plane.  The GEMM runs three
  // passes per 64-channel slice -- W_hi.X_hi, W_hi.X_lo, W_lo.X_hi (the lo.lo term, <= 2^-22 of the product, is dropped) -- into the
  // same fp32 accumulators; weight rows are pre-scaled by a power of two per row (so that their lo parts stay fp16-normal) and the
  // epilogue multiplies the accumulator by wscale[row] (the inverse, exact) before the bias.  Outputs are written as hi / lo planes;
  // res / up are read as hi + lo.  fp32 outputs (out_f32, mode 2) as in the fp16 mode.
  int split;
  long long in_lo, w_lo, out_lo, res_lo, up_lo, in2_lo;
  const float* wscale;  // [rows] fp32, power of two
  long long head_w_lo;        // fused head (head_w) in the split-operand mode: offset of its lo plane, and
  const float* head_scale;    // its inverse row scales [16]
  ConvSeg seg[RS_MAX_SEGS];
};

int launch_conv(const ConvParams& p, hipStream_t stream, int force_variant /* -1 auto */, int use_glds);
int conv_choose_variant(ConvParams& p, int force_variant, int use_glds);   // the dispatch rule (also sets p.stages / p.persist)
extern thread_local int g_last_conv_variant;

// ------------------------------------------------------------------ fused stem (stem_fused.hip): conv 7x7 s2 + ReLU + max-pool 3x3 s2
struct StemPoolParams {
  const half_t* in;     // [N][in_Hp][in_Wp][4] fp16, halo 3
  const half_t* wf;     // the [64][256] stem matrix (k = kh*32 + kw*4 + c: 8 taps per filter row, the 8th zero) in MFMA A-fragment order:
                        // [kh 7][16-row block 4][lane 64][8] (weights.py "stem.conv1f")
  const float* bias;    // [64]
  half_t* out;          // pooled map [N][Hq+2][Wq+2][64], halo 1
  int N, in_Hp, in_Wp;
  int Hc, Wc;           // conv output size
  int Hq, Wq;           // pooled size
};
int launch_stem_pool(const StemPoolParams& p, hipStream_t stream);
// The same launch in the split-operand precision mode: input and pooled output as hi + lo planes, the weight fragments as [2 planes][7][4][64][8]
// (hi plane first) of the row-scaled matrix with the inverse scales in wscale.
struct StemPoolSplitParams {
  const half_t* in; long long in_lo;
  const half_t* wf;
  const float* wscale;
  const float* bias;
  half_t* out; long long out_lo;
  int N, in_Hp, in_Wp;
  int Hc, Wc;
  int Hq, Wq;
};
int launch_stem_pool_split(const StemPoolSplitParams& p, hipStream_t stream);

// ------------------------------------------------------------------ fused bottleneck tail (bneck_fused.hip)
// conv2 (3x3 64->64) + conv3 (1x1 64->256, + residual + ReLU) [+ the next block's conv1 (1x1 256->64)] of an identity-shortcut
// bottleneck block in one launch.  Every map is NHWC fp16 with a zero halo of 1 and the same H x W.
struct BneckParams {
  const half_t* t1;     // conv2 input  [N][H+2][W+2][64]
  const half_t* w2;     // [64][576] (kh, kw, cin)
  const float* b2;
  const half_t* w3p;    // conv3 weight [256][64], K columns in the register-chaining order (weights.py _perm_k64)
  const float* b3;
  const half_t* x;      // block input = residual [N][H+2][W+2][256] (identity shortcut); nullptr with x0 / wsc
  const half_t* x0;     // projection shortcut: its 64-channel input [N][H+2][W+2][64] at the same resolution (res2.0: the stem output)
  const half_t* wsc;    //   and its weight [256][64], natural K order; b3 then holds conv3's + the shortcut's bias
  half_t* out;          // block output            [N][H+2][W+2][256]
  const half_t* w1p;    // next block's conv1 [64][256], K columns permuted per group of 64; nullptr = stop after conv3
  const float* b1;
  half_t* t1n;          // next block's conv1 output [N][H+2][W+2][64]
  int M, H, W, Hp, Wp;  // M = N*H*W pixels
  int CB;               // bottleneck width / 64: 1 (64 -> 256, res2) or 2 (128 -> 512, res3); every "64" / "256" above scales with it
};
int launch_bneck_tail(const BneckParams& p, hipStream_t stream);

// The same tail in the split-operand precision mode (bneck_split.hip): every tensor a hi plane with its lo plane `*_lo` ELEMENTS behind it, weights
// row-scaled with the inverse scales s2 / s3 / s1 (ConvParams::split); identity shortcut only.
struct BneckSplitParams {
  const half_t* t1; long long t1_lo;        // conv2 input  [N][H+2][W+2][64 CB]
  const half_t* w2; long long w2_lo;        // [64 CB][9 * 64 CB]
  const float* b2; const float* s2;
  const half_t* w3p; long long w3_lo;       // conv3 weight [256 CB][64 CB], K columns in the register-chaining order (weights.py _perm_k64, group = 64 CB)
  const float* b3; const float* s3;
  const half_t* x; long long x_lo;          // block input = residual [N][H+2][W+2][256 CB]
  half_t* out; long long out_lo;            // block output
  const half_t* w1p; long long w1_lo;       // next block's conv1 [64 CB][256 CB], K columns permuted per group of 64; nullptr = stop after conv3
  const float* b1; const float* s1;
  half_t* t1n; long long t1n_lo;            // next block's conv1 output [N][H+2][W+2][64 CB]
  // projection-shortcut form (first block of res2; CB = 1): x = nullptr and x0 [N][H+2][W+2][64] is the shortcut's input; w3p then is [256][128] =
  // [conv3 (chained K order) | shortcut (natural K order)] with ONE power-of-two scale per row over both parts (w3_lo = 256 * 128), b3 the sum of the
  // two biases: out = relu((W3 . t2 + Wsc . x0) * s3 + b3)
  const half_t* x0; long long x0_lo;
  int M, H, W, Hp, Wp;
  int CB;
};
int launch_bneck_tail_split(const BneckSplitParams& p, hipStream_t stream);

// ------------------------------------------------------------------ raster voting (raster_vote.hip)
int launch_mask_overlap(const uint8_t* det, int n_det, const uint8_t* lab, int n_lab, int h, int w, int* inter, int* lab_area, hipStream_t s);

// ------------------------------------------------------------------ training: weight gradient (conv_wgrad.hip)
struct WgradParams {
  const half_t* dy;     // gradient of the layer output, [N][Ho+2*dy_pad][Wo+2*dy_pad][dy_Cs] fp16, zero halo
  const half_t* x;      // layer input as the forward saw it (zero halo)
  float* partial;       // [splits][Cout][Kpad] fp32 scratch
  float* grad;          // [Cout][Kpad] fp32, K = (kh,kw,cin) with cin fastest (the forward weight layout)
  const half_t* zeros;  // >= Cout zero halfs: source of the dY rows past M (so the tail contributes nothing)
  const float* scale;   // optional [Cout]: FrozenBN scale folded into the forward weight (grad is w.r.t. the unfolded one)
  int M, Ho, Wo;
  int dy_Hp, dy_Wp, dy_Cs, dy_pad;
  int in_Hp, in_Wp, in_Cs, in_off;   // as ConvParams
  int stride, KH, KW, Cin, Cout, Kpad;
  const int* m_count;   // optional device counter: rows = min(M, *m_count * m_mul) (RoI heads: valid entries only)
  int m_mul;
  int splits;           // pixel-range splits (gridDim.z); wgrad_splits() proposes one
  int accumulate;       // 1: grad += result
  int f32;              // 1: dy and x point to fp32 data (reference-precision trainer): conv_wgrad_f32_kernel
};
int wgrad_splits(const WgradParams& p);
int launch_conv_wgrad(const WgradParams& p, hipStream_t stream);

// ------------------------------------------------------------------ tile ingest / pooling
struct PreprocParams {
  const uint8_t* tiles;   // [N][H][W][C] uint8, channel order as cv2.imread (BGR)
  half_t* out;            // [N][Hp+6][Wp+6][4] fp16, halo 3
  const int* hb;          // [new_w][2]  (first source column, tap count)
  const int* hk;          // [new_w][ksh] Pillow fixed-point coefficients (22 bit)
  const int* vb;          // [new_h][2]
  const int* vk;          // [new_h][ksv]
  int N, H, W, C;
  int new_h, new_w;       // resized image size
  int out_Hp, out_Wp;     // padded dims incl. halo
  int ksh, ksv;
  int need_h, need_v;
  int flip;               // 1: model channel c reads source channel C-1-c
  int out_f32;            // fp32 validation mode: out is float; 2 = split-operand mode: out is the hi plane, the lo plane out_lo elements behind it
  long long out_lo;
  float mean[4], stdv[4];
};
int launch_preprocess(const PreprocParams& p, hipStream_t s);
int launch_maxpool(const half_t* in, half_t* out, int N, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s);
int launch_subsample2(const half_t* in, half_t* out, int N, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s);
int launch_maxpool_f32(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s);
// split-operand mode: in / out are hi planes, the lo planes in_lo / out_lo elements behind them
int launch_maxpool_split(const half_t* in, long long in_lo, half_t* out, long long out_lo, int N, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s);
int launch_subsample2_split(const half_t* in, long long in_lo, half_t* out, long long out_lo, int N, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s);
int launch_subsample2_f32(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t s);
int launch_conv_f32(const ConvParams& p, hipStream_t stream, int force_valu = 0, int tile = 0);   // ref_f32.hip: fp32 MFMA (or the VALU cross-check)
int launch_conv_deep(const ConvParams& p, hipStream_t stream, int tile_px = 256);
int rs_device_cu_count();                                        // CUs of the current device (256 without one), conv_deep.hip
bool conv_wreg_ok(const ConvParams& p);                          // conv_wreg.hip: persistent 1x1 (Cin 256) with register-resident weights, variant 22
int launch_conv_wreg(const ConvParams& p, hipStream_t stream, int waves = 0);   // waves: 4 / 8 per workgroup, 0 = RS_WREG_WAVES
int launch_conv_deep_multi(const ConvParams& common, const ConvSeg* segs, const int* m_per_image, int nseg, int images, hipStream_t stream);
