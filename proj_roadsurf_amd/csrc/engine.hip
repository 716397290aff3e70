// librs_engine.so host side: C ABI (include/rs_engine.h), weight blob, workspace and the op list
// of one GeneralizedRCNN.inference pass ([EXT d2: modeling/meta_arch/rcnn.py]; topology fixed by
// R:config/detectron2_config_3bands.yaml).  All device work is enqueued on one HIP stream with
// fixed-capacity buffers and device-side counts: no host synchronisation inside a forward.
#include <stdarg.h>
#include <string.h>

#include <cmath>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/rs_engine.h"
#include "detect.h"
#include "train.h"

// ------------------------------------------------------------------------------------- errors
static thread_local char g_err[1024] = "";
static bool g_trainer_unfused_shortcut = false;   // set by rs_trainer_create while it builds its forward engine

// ---------------------------------------------------------------- debug switches: the one place that reads the environment
static RsDebug g_debug;
static bool g_debug_loaded = false;
void rs_debug_reload() {
  RsDebug d;
  auto rd = [](const char* name, int* v) { const char* e = getenv(name); if (e && *e) *v = atoi(e); };
  rd("RS_CONV_SINGLE_STAGE_NK", &d.conv_single_stage_nk); rd("RS_CONV_PERSIST", &d.conv_persist); rd("RS_CONV_TUNED", &d.conv_tuned);
  rd("RS_CONV_DEEP", &d.conv_deep); rd("RS_CONV_WIDE_PX", &d.conv_wide_px); rd("RS_CONV_WREG", &d.conv_wreg); rd("RS_WREG_DBG", &d.wreg_dbg); rd("RS_ROI_BWD_ATOMIC", &d.roi_bwd_atomic); rd("RS_WREG_WAVES", &d.wreg_waves); rd("RS_STEM_SMALL_TILE", &d.stem_small_tile); rd("RS_DEEP_DBG", &d.deep_dbg);
  rd("RS_DECONV_VARIANT", &d.deconv_variant); rd("RS_FUSE_MASK_PREDICTOR", &d.fuse_mask_predictor); rd("RS_SIDE_STREAM", &d.side_stream);
  rd("RS_NARROW_ROIALIGN", &d.narrow_roialign); rd("RS_USE_GLDS", &d.use_glds); rd("RS_FUSE_SHORTCUT", &d.fuse_shortcut); rd("RS_MERGE_LEVELS", &d.merge_levels); rd("RS_FUSE_RPN_HEADS", &d.fuse_rpn_heads); rd("RS_DEEP_TAIL", &d.deep_tail); rd("RS_DEEP_TILE_PX", &d.deep_tile_px); rd("RS_FUSE_BNECK", &d.fuse_bneck); rd("RS_FUSE_STEM", &d.fuse_stem);
  rd("RS_USE_GRAPH", &d.use_graph); rd("RS_GRAPH_SMALL", &d.graph_small); rd("RS_TRAIN_ROI_SIDE", &d.train_roi_side);
  rd("RS_TRAIN_SIDE", &d.train_side); rd("RS_WGRAD_TARGET", &d.wgrad_target); rd("RS_WGRAD_CB", &d.wgrad_cb);
  rd("RS_SELECT_DEBUG", &d.select_debug); rd("RS_NMS_DEBUG", &d.nms_debug); rd("RS_ROI_WINDOW", &d.roi_window); rd("RS_ROI_ORDER", &d.roi_order);
  g_debug = d;
  g_debug_loaded = true;
}
const RsDebug& rs_debug() {
  if (!g_debug_loaded) rs_debug_reload();
  return g_debug;
}
void rs_set_error(const char* fmt, ...) {
  va_list a;
  va_start(a, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, a);
  va_end(a);
}

namespace {

// DT_SPLIT16: an activation of the split-operand mode -- two fp16 planes of the registered shape back to back (hi, then lo); value = hi + lo
enum { DT_F16 = 1, DT_F32 = 2, DT_I32 = 3, DT_U8 = 4, DT_SPLIT16 = 5 };
static size_t dt_size(int dt) { return (dt == DT_F16 || dt == DT_SPLIT16) ? 2 : (dt == DT_U8 ? 1 : 4); }

struct TensorInfo {
  std::string name;
  void* p = nullptr;
  int dtype = 0, ndim = 0, halo = 0;
  int64_t dims[5] = {1, 1, 1, 1, 1};
  size_t bytes = 0;
};

struct Act {   // NHWC fp16 activation with halo
  half_t* p = nullptr;
  long long lo = 0;   // split-operand mode: element offset of the lo plane behind p (0 = single plane)
  int N = 0, H = 0, W = 0, C = 0, pad = 0;
  int Hp() const { return H + 2 * pad; }
  int Wp() const { return W + 2 * pad; }
};

struct Stage {
  std::string name;
  std::function<int(int, hipStream_t)> fn;
  double flops_per_image = 0, bytes_per_image = 0;   // algorithmic, per tile (0 = n/a)
  double ms_total = 0;
  int calls = 0;
  double last_flops = 0, last_bytes = 0;
  int variant = -2;      // conv tile variant of the last call (-2 = not a conv stage)
  bool narrow = false;   // latency-bound detection glue (few workgroups): runs on the engine's side stream
  bool grad_side = false;   // trainer: weight / bias gradient, off the input-gradient chain (may run on the trainer's side stream)
  int bucket = -1;          // trainer: gradient bucket this stage writes into (rs_trainer::buckets), -1 = none
  int phase = 0;         // 0 = preprocess..RPN proposals, 1 = box head..detections, 2 = mask head + paste
  hipEvent_t handoff = nullptr;   // recorded on the previous stage's stream when this stage switches streams
};

struct BlobEntry { const void* host; void* dev; int dtype; int ndim; int64_t dims[4]; size_t nbytes; };

}  // namespace

// host-only helpers -------------------------------------------------------------------------
extern "C" void rs_resize_shape(int h, int w, int short_edge, int max_size, int* new_h, int* new_w) {
  // [EXT d2: data/transforms/augmentation_impl.py ResizeShortestEdge.get_output_shape]
  double scale = (double)short_edge * 1.0 / (double)(h < w ? h : w);
  double newh, neww;
  if (h < w) { newh = short_edge; neww = scale * w; } else { newh = scale * h; neww = short_edge; }
  const double mx = newh > neww ? newh : neww;
  if (mx > max_size) {
    scale = (double)max_size * 1.0 / mx;
    newh = newh * scale;
    neww = neww * scale;
  }
  *new_w = (int)(neww + 0.5);
  *new_h = (int)(newh + 0.5);
}

extern "C" int rs_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* coeffs) {
  // Pillow src/libImaging/Resample.c precompute_coeffs (bilinear, support 1) + normalize_coeffs_8bpc
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  if (!bounds || !coeffs) return ksize;
  const double ss = 1.0 / filterscale;
  std::vector<double> w(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < ksize; ++x) w[x] = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0) a = -a;
      const double v = a < 1.0 ? 1.0 - a : 0.0;
      w[x] = v;
      ww += v;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) w[x] /= ww;
    for (int x = 0; x < ksize; ++x) {
      const double v = w[x] * (double)(1 << 22);
      coeffs[xx * ksize + x] = v < 0 ? (int)(v - 0.5) : (int)(v + 0.5);
    }
    bounds[xx * 2] = xmin;
    bounds[xx * 2 + 1] = xmax;
  }
  return ksize;
}

// =================================================================================== engine
struct rs_engine {
  rs_spec spec;
  int device = 0;
  hipStream_t stream = nullptr;         // "wide" stream: every kernel that fills the chip (may be shared between engines)
  bool own_stream = false;
  hipStream_t copy_stream = nullptr;    // device-to-host result copies (rs_engine_fetch_async), overlapping the next batch
  hipEvent_t ev_results = nullptr;      // recorded on `stream` when a forward's results are complete
  hipEvent_t ev_copied = nullptr;       // recorded on `copy_stream` after the last result copy; the next forward's box head waits for it
  bool copy_pending = false;
  hipStream_t narrow = nullptr;         // side stream for the latency-bound glue kernels (null = everything on `stream`)
  bool on_narrow = false;               // which stream the most recently enqueued stage went to
  hipEvent_t ev_join = nullptr;         // narrow -> wide join at the end of a forward that ends on the side stream
  bool cur_record = false;              // profiling decision of the forward in flight (taken at phase 0)
  int max_batch = 0, tile_h = 0, tile_w = 0, tile_c = 0;
  int net_h = 0, net_w = 0, pad_h = 0, pad_w = 0;
  int use_glds = 1;    // -1 = fp32 validation path (launch_conv forwards to launch_conv_f32)
  bool f32 = false;    // rs_spec.precision == 1: activations and weights are float
  bool split = false;  // rs_spec.precision == 2: split-operand mode -- activations and weights as hi + lo fp16 planes, three MFMA passes (common.h ConvParams::split)
  int profiling = 0;   // 0 off, 1 = events + host sync per stage, 2 = events only (resolved later)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;   // mode 2
  std::vector<int> ev_stage;
  std::vector<int> ev_batch;
  size_t ev_used = 0;
  int resolve_profile();

  void* blob_dev = nullptr;
  std::map<std::string, BlobEntry> blob;
  std::vector<void*> allocs;
  std::vector<TensorInfo> tensors;
  std::vector<Stage> stages;

  uint8_t* tiles_dev = nullptr;
  // per-image resized sizes inside the net_h x net_w canvas (training: INPUT.MIN_SIZE_TRAIN drawn per image, the batch padded to the
  // largest -- [EXT d2: data/dataset_mapper.py, structures/image_list.py]); empty = every image fills the canvas
  struct ResizeTab { int* b = nullptr; int* k = nullptr; int ks = 0; };
  std::map<int, ResizeTab> tab_h, tab_v;   // by output size
  std::vector<int> img_new_h, img_new_w;
  float* img_hw_dev = nullptr;             // [max_batch][2] clip size of the proposals per image (h, w)
  int resize_tab(int in_size, int out_size, std::map<int, ResizeTab>& cache, ResizeTab* out);
  int set_image_sizes(const int32_t* new_h, const int32_t* new_w, int n);
  // results (device)
  int* det_count = nullptr;
  float* det_boxes = nullptr;
  float* det_boxes_net = nullptr;
  float* det_scores = nullptr;
  int* det_classes = nullptr;
  uint8_t* masks = nullptr;
  float* mask_probs = nullptr;
  int D = 0;
  // mask crops for the host (rs_engine_fetch_crops_*): table + compacted data on the device, byte count read back pinned
  int* crop_rects = nullptr;
  unsigned int* crop_offsets = nullptr;
  unsigned long long* crop_total = nullptr;
  uint8_t* crop_data = nullptr;
  unsigned long long* h_crop_total = nullptr;   // pinned
  hipEvent_t ev_crop_hdr = nullptr;

  int alloc(void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~(size_t)255;
    RS_HIP(hipMalloc(p, bytes));
    allocs.push_back(*p);
    RS_HIP(hipMemsetAsync(*p, 0, bytes, stream));
    return RS_OK;
  }
  void reg(const std::string& name, void* p, int dtype, std::vector<int64_t> dims, int halo) {
    TensorInfo t;
    t.name = name; t.p = p; t.dtype = dtype; t.ndim = (int)dims.size(); t.halo = halo;
    size_t nb = dt_size(dtype);
    for (size_t i = 0; i < dims.size(); ++i) { t.dims[i] = dims[i]; nb *= (size_t)dims[i]; }
    t.bytes = dtype == DT_SPLIT16 ? 2 * nb : nb;
    tensors.push_back(t);
  }
  int new_act(Act* a, const std::string& name, int N, int H, int W, int C, int pad) {
    a->N = N; a->H = H; a->W = W; a->C = C; a->pad = pad;
    const size_t bytes = (size_t)N * a->Hp() * a->Wp() * C * (f32 ? 4 : 2) * (split ? 2 : 1);
    int rc = alloc((void**)&a->p, bytes);
    if (rc) return rc;
    a->lo = split ? (long long)N * a->Hp() * a->Wp() * C : 0;
    reg(name, a->p, f32 ? DT_F32 : (split ? DT_SPLIT16 : DT_F16), {N, a->Hp(), a->Wp(), C}, pad);
    return RS_OK;
  }
  const BlobEntry* find(const std::string& n) {
    auto it = blob.find(n);
    return it == blob.end() ? nullptr : &it->second;
  }
  // GEMM weights of a layer: "<layer>.w" (fp16) or "<layer>.w32" in the fp32 validation mode
  // split-operand mode: "<layer>.ws" = fp16 [2][rows][Kpad] (hi rows, then lo rows, of the row-scaled weight) + "<layer>.wsi" fp32 [rows] (inverse scales)
  const BlobEntry* findw(const std::string& layer) { return find(layer + (f32 ? ".w32" : (split ? ".ws" : ".w"))); }
  int wrows(const BlobEntry* w) const { return (int)(split ? w->dims[0] / 2 : w->dims[0]); }
  // fills the split-operand fields of a conv whose weight entry is w (no-op in the other modes)
  int set_split(ConvParams* p, const std::string& wname, const BlobEntry* w, const Act* in, const Act* out, const Act* res, const Act* up, const Act* in2) {
    if (!split) return RS_OK;
    const BlobEntry* si = find(wname + ".wsi");
    RS_CHECK(si && si->dtype == DT_F32 && si->dims[0] >= w->dims[0] / 2 && (w->dims[0] & 1) == 0, RS_ERR_BLOB, "row scales of %s missing from blob (split-operand mode)", wname.c_str());
    p->split = 1;
    p->wscale = (const float*)si->dev;
    p->w_lo = (long long)(w->dims[0] / 2) * w->dims[1];
    if (in) p->in_lo = in->lo;
    if (out) p->out_lo = out->lo;
    if (res) p->res_lo = res->lo;
    if (up) p->up_lo = up->lo;
    if (in2) p->in2_lo = in2->lo;
    return RS_OK;
  }
  int parse_blob(const void* data, size_t nbytes);
  int build();
  struct DeferredConv { ConvParams p; int m_per_image = 0; double flops = 0, bytes = 0; };
  int add_conv(const std::string& name, const std::string& wname, const Act& in, const Act& out, int k, int stride,
               int pad, bool relu, const Act* res, const Act* up, int cin_real, int units_per_tile = 1,
               const int* m_count = nullptr, const Act* in2 = nullptr, int stride2 = 1, DeferredConv* defer = nullptr);
  int add_merged_convs(const std::string& name, const std::vector<DeferredConv>& d);
  int run(const uint8_t* tiles, int n, int phase = -1);
  int run_stages(int n, bool record, int phase = -1, bool all_wide = false);
  int assign_phases();
  int use_graph = 0;
  int fuse_shortcut = 1;
  int fuse_bneck = 1;
  int fuse_stem = 1;      // stem conv + ReLU + max-pool as one launch (inference engines, fp16 path)
  bool frozen_fusions_only = false;   // a trainer's forward engine: layer fusions only where nothing is differentiated (stem + res2 at FREEZE_AT 2)
  int merge_levels = 1;   // FPN output convs / RPN 3x3 of all levels as one multi-map launch each (inference engines, fp16 path)
  long long forward_index = 0;
  std::set<int> warmed;
  std::map<int, hipGraphExec_t> graphs;
};

int rs_engine::parse_blob(const void* data, size_t nbytes) {
  const uint8_t* b = (const uint8_t*)data;
  RS_CHECK(nbytes >= 16, RS_ERR_BLOB, "weight blob too small");
  uint32_t hdr[4];
  memcpy(hdr, b, 16);
  RS_CHECK(hdr[0] == 0x52534557u && hdr[1] == 1u, RS_ERR_BLOB, "bad weight blob magic/version %08x/%u", hdr[0], hdr[1]);
  const size_t ent = 96 + 4 + 4 + 32 + 8 + 8;
  RS_CHECK(16 + ent * hdr[2] <= nbytes, RS_ERR_BLOB, "weight blob truncated");
  RS_HIP(hipMalloc(&blob_dev, nbytes));
  RS_HIP(hipMemcpy(blob_dev, data, nbytes, hipMemcpyHostToDevice));
  for (uint32_t i = 0; i < hdr[2]; ++i) {
    const uint8_t* e = b + 16 + ent * i;
    char name[97];
    memcpy(name, e, 96);
    name[96] = 0;
    BlobEntry be;
    uint32_t dt, nd;
    uint64_t dims[4], off, nb;
    memcpy(&dt, e + 96, 4);
    memcpy(&nd, e + 100, 4);
    memcpy(dims, e + 104, 32);
    memcpy(&off, e + 136, 8);
    memcpy(&nb, e + 144, 8);
    RS_CHECK(off + nb <= nbytes, RS_ERR_BLOB, "weight blob entry %s out of range", name);
    be.host = b + off;
    be.dev = (char*)blob_dev + off;
    be.dtype = (int)dt;
    be.ndim = (int)nd;
    for (int d = 0; d < 4; ++d) be.dims[d] = (int64_t)dims[d];
    be.nbytes = nb;
    blob[name] = be;
  }
  return RS_OK;
}

// One conv / linear stage.  `cin_real` only feeds the FLOP count (stem: 3 of 8 padded channels).
// `units_per_tile` = images of the conv per input tile (1 for feature maps, D for per-RoI maps);
// `m_count` = optional device-side count of units actually present.
int rs_engine::add_conv(const std::string& name, const std::string& wname, const Act& in, const Act& out, int k,
                        int stride, int pad, bool relu, const Act* res, const Act* up, int cin_real, int units_per_tile,
                        const int* m_count, const Act* in2, int stride2, DeferredConv* defer) {
  const BlobEntry* w = findw(wname);
  const BlobEntry* b = find(wname + ".b");
  RS_CHECK(w && b, RS_ERR_BLOB, "weights for %s missing from blob", wname.c_str());
  RS_CHECK(w->dtype == (f32 ? DT_F32 : DT_F16) && b->dtype == DT_F32, RS_ERR_BLOB, "weights for %s have wrong dtype", wname.c_str());
  ConvParams p;
  memset(&p, 0, sizeof p);
  { int rc = set_split(&p, wname, w, &in, &out, res, up, in2); if (rc) return rc; }
  p.in = in.p; p.w = (const half_t*)w->dev; p.bias = (const float*)b->dev; p.out = out.p;
  p.res = res ? res->p : nullptr;
  p.up = up ? up->p : nullptr;
  p.Ho = out.H; p.Wo = out.W;
  p.in_Hp = in.Hp(); p.in_Wp = in.Wp(); p.in_Cs = in.C; p.in_off = in.pad - pad;
  p.stride = stride; p.KH = k; p.KW = k; p.Cin = in.C;
  p.Kpad = (int)w->dims[1];
  p.Cout = out.C;
  p.out_Hp = out.Hp(); p.out_Wp = out.Wp(); p.out_Cs = out.C; p.out_pad = out.pad;
  if (up) { p.up_Hp = up->Hp(); p.up_Wp = up->Wp(); p.up_Cs = up->C; p.up_pad = up->pad; }
  p.relu = relu ? 1 : 0;
  if (in2) {   // second K source: 1x1 taps at stride2 (projection shortcut folded into conv3)
    RS_CHECK((out.H - 1) * stride2 < in2->H && (out.W - 1) * stride2 < in2->W && in2->C % 64 == 0, RS_ERR_ARG, "%s: second source geometry", name.c_str());
    p.in2 = in2->p; p.in2_Hp = in2->Hp(); p.in2_Wp = in2->Wp(); p.in2_Cs = in2->C; p.in2_off = in2->pad;
    p.stride2 = stride2; p.Cin2 = in2->C;
  }
  RS_CHECK(in.pad >= pad, RS_ERR_ARG, "%s: input halo %d < conv pad %d", name.c_str(), in.pad, pad);
  RS_CHECK(wrows(w) >= out.C, RS_ERR_BLOB, "%s: weight rows %d < Cout %d", name.c_str(), wrows(w), out.C);
  RS_CHECK((out.H - 1) * stride + k - 2 * pad <= in.H + (stride - 1), RS_ERR_ARG, "%s: geometry", name.c_str());
  if (res) RS_CHECK(res->H == out.H && res->W == out.W && res->C == out.C && res->pad == out.pad, RS_ERR_ARG, "%s: residual geometry", name.c_str());
  if (in.C < 64) {
    // small-Cin (stem) path: per-16-byte-chunk element offsets.  C == 8: one tap per chunk.  C == 4: the tap row
    // is padded to 8 taps (the 8th has zero weights) and a chunk holds two horizontally adjacent taps, so
    // K = kh*8*4 = 224 -> 256 instead of 49*8 = 392 -> 448.
    RS_CHECK(in.C == 8 || in.C == 4, RS_ERR_UNSUPPORTED, "%s: Cin %d", name.c_str(), in.C);
    std::vector<int> koff(p.Kpad / 8, 0);
    if (in.C == 8) {
      for (int t = 0; t < k * k && t < (int)koff.size(); ++t) koff[t] = ((t / k) * p.in_Wp + (t % k)) * in.C;
    } else {
      RS_CHECK(k == 7 && (p.in_Wp & 1) == 0 && ((in.pad - pad) & 1) == 0 && stride == 2, RS_ERR_UNSUPPORTED, "%s: C=4 stem needs 7x7 s2 and even pitch", name.c_str());
      p.KW = 8;
      for (int t = 0; t < k * 4 && t < (int)koff.size(); ++t) koff[t] = ((t / 4) * p.in_Wp + (t % 4) * 2) * in.C;
    }
    int* d = nullptr;
    int rc = alloc((void**)&d, koff.size() * 4);
    if (rc) return rc;
    RS_HIP(hipMemcpyAsync(d, koff.data(), koff.size() * 4, hipMemcpyHostToDevice, stream));
    RS_HIP(hipStreamSynchronize(stream));
    p.koff = d;
  }
  const int m_per_image = out.H * out.W * units_per_tile;
  p.m_count = m_count;
  p.m_mul = out.H * out.W;
  Stage st;
  st.name = name;
  st.flops_per_image = 2.0 * m_per_image * ((double)k * k * cin_real + (in2 ? in2->C : 0)) * out.C;
  // algorithmic bytes: the input pixels the convolution actually reads (a stride-s 1x1 touches every s-th pixel of every
  // s-th row only), the output once, the residual once, the second K source at the output's pixel count
  const double in_px = k >= stride ? (double)in.H * in.W : (double)out.H * out.W * k * k;
  st.bytes_per_image = (split ? 4.0 : 2.0) * (in_px * in.C * units_per_tile + (double)m_per_image * out.C * (1 + (res ? 1 : 0)) +
                              (in2 ? (double)m_per_image * in2->C : 0.0));
  const int glds = use_glds;
  if (defer) {           // the caller merges this convolution into a multi-map launch (add_merged_convs)
    defer->p = p; defer->m_per_image = m_per_image; defer->flops = st.flops_per_image; defer->bytes = st.bytes_per_image;
    return RS_OK;
  }
  st.fn = [p, m_per_image, glds](int n, hipStream_t s) mutable {
    p.M = n * m_per_image;
    return launch_conv(p, s, -1, glds);
  };
  stages.push_back(st);
  return RS_OK;
}

// One stage = one conv_deep launch over several maps (launch_conv_deep_multi)
int rs_engine::add_merged_convs(const std::string& name, const std::vector<DeferredConv>& d) {
  RS_CHECK(!d.empty() && d.size() <= RS_MAX_SEGS, RS_ERR_ARG, "%s: %d maps", name.c_str(), (int)d.size());
  ConvParams common = d[0].p;
  std::vector<ConvSeg> segs(d.size());
  std::vector<int> mpi(d.size());
  Stage st;
  st.name = name;
  for (size_t i = 0; i < d.size(); ++i) {
    const ConvParams& q = d[i].p;
    RS_CHECK(q.Cin == common.Cin && q.Cout == common.Cout && q.KH == common.KH && q.KW == common.KW && q.stride == 1 && q.in_Cs == common.in_Cs &&
                 q.in_off == common.in_off && q.out_Cs == common.out_Cs && q.out_pad == common.out_pad && q.Kpad == common.Kpad && q.relu == common.relu &&
                 !q.res && !q.up && !q.in2 && !q.m_count && q.mode == 0 && !q.out_f32,
             RS_ERR_ARG, "%s: map %d does not share the launch parameters of map 0", name.c_str(), (int)i);
    RS_CHECK(q.head_w == common.head_w && q.head_b == common.head_b && q.head_scale == common.head_scale, RS_ERR_ARG, "%s: map %d has another fused head", name.c_str(), (int)i);
    RS_CHECK(q.split == common.split && q.w_lo == common.w_lo, RS_ERR_ARG, "%s: map %d differs in the split-operand fields", name.c_str(), (int)i);
    segs[i].in = q.in; segs[i].w = q.w; segs[i].bias = q.bias; segs[i].out = q.out; segs[i].head_out = q.head_out;
    segs[i].in_lo = q.in_lo; segs[i].out_lo = q.out_lo; segs[i].wscale = q.wscale;
    segs[i].Ho = q.Ho; segs[i].Wo = q.Wo; segs[i].in_Hp = q.in_Hp; segs[i].in_Wp = q.in_Wp; segs[i].out_Hp = q.out_Hp; segs[i].out_Wp = q.out_Wp;
    mpi[i] = d[i].m_per_image;
    st.flops_per_image += d[i].flops;
    st.bytes_per_image += d[i].bytes;
  }
  st.fn = [common, segs, mpi](int n, hipStream_t s) {
    g_last_conv_variant = 12;
    return launch_conv_deep_multi(common, segs.data(), mpi.data(), (int)segs.size(), n, s);
  };
  stages.push_back(st);
  return RS_OK;
}

int rs_engine::resize_tab(int in_size, int out_size, std::map<int, ResizeTab>& cache, ResizeTab* out) {
  auto it = cache.find(out_size);
  if (it == cache.end()) {
    ResizeTab t;
    t.ks = rs_resize_coeffs(in_size, out_size, nullptr, nullptr);
    std::vector<int32_t> b((size_t)out_size * 2), k((size_t)out_size * t.ks);
    rs_resize_coeffs(in_size, out_size, b.data(), k.data());
    int rc;
    if ((rc = alloc((void**)&t.b, b.size() * 4))) return rc;
    if ((rc = alloc((void**)&t.k, k.size() * 4))) return rc;
    RS_HIP(hipMemcpyAsync(t.b, b.data(), b.size() * 4, hipMemcpyHostToDevice, stream));
    RS_HIP(hipMemcpyAsync(t.k, k.data(), k.size() * 4, hipMemcpyHostToDevice, stream));
    RS_HIP(hipStreamSynchronize(stream));          // the host vectors go out of scope
    it = cache.emplace(out_size, t).first;
  }
  *out = it->second;
  return RS_OK;
}

// n = 0: every image fills the canvas again
int rs_engine::set_image_sizes(const int32_t* new_h, const int32_t* new_w, int n) {
  RS_CHECK(n >= 0 && n <= max_batch, RS_ERR_ARG, "set_image_sizes: %d images, engine built for %d", n, max_batch);
  std::vector<float> hw((size_t)max_batch * 2);
  for (int i = 0; i < max_batch; ++i) { hw[2 * i] = (float)net_h; hw[2 * i + 1] = (float)net_w; }
  for (int i = 0; i < n; ++i) {
    RS_CHECK(new_h[i] >= 1 && new_h[i] <= net_h && new_w[i] >= 1 && new_w[i] <= net_w, RS_ERR_ARG,
             "image %d: %d x %d does not fit the %d x %d canvas", i, new_h[i], new_w[i], net_h, net_w);
    hw[2 * i] = (float)new_h[i]; hw[2 * i + 1] = (float)new_w[i];
  }
  for (int i = 0; i < n; ++i) {                     // build the tables now: the stage itself must not synchronise
    ResizeTab t;
    int rc;
    if ((rc = resize_tab(tile_w, new_w[i], tab_h, &t))) return rc;
    if ((rc = resize_tab(tile_h, new_h[i], tab_v, &t))) return rc;
  }
  img_new_h.assign(new_h, new_h + n);
  img_new_w.assign(new_w, new_w + n);
  RS_HIP(hipMemcpyAsync(img_hw_dev, hw.data(), hw.size() * 4, hipMemcpyHostToDevice, stream));
  RS_HIP(hipStreamSynchronize(stream));
  return RS_OK;
}

int rs_engine::build() {
  const rs_spec& S = spec;
  const int NB = max_batch;
  rs_resize_shape(tile_h, tile_w, S.min_size_test, S.max_size_test, &net_h, &net_w);
  const int dv = S.size_divisibility;
  pad_h = (net_h + dv - 1) / dv * dv;
  pad_w = (net_w + dv - 1) / dv * dv;
  RS_CHECK(S.fpn_out_channels == 256 && S.mask_conv_dim == 256, RS_ERR_UNSUPPORTED, "FPN/mask width must be 256");
  RS_CHECK(S.num_levels == 5, RS_ERR_UNSUPPORTED, "RPN must use p2..p6");
  RS_CHECK(S.rpn_pre_nms_topk <= 1024 && S.rpn_post_nms_topk <= 1024, RS_ERR_UNSUPPORTED, "RPN top-k > 1024");
  RS_CHECK(S.num_classes >= 1 && S.num_classes <= 8, RS_ERR_UNSUPPORTED, "NUM_CLASSES %d outside [1,8]", S.num_classes);
  RS_CHECK(S.in_channels >= 1 && S.in_channels <= 4 && S.in_channels == tile_c, RS_ERR_ARG, "tile channels %d vs PIXEL_MEAN %d", tile_c, S.in_channels);
  RS_CHECK(!S.mask_on || S.mask_pooler_resolution * 2 == RS_MASK_SIDE, RS_ERR_UNSUPPORTED, "mask side must be 28");
  int rc;

  // ---- resize tables + input staging
  RS_CHECK(alloc((void**)&tiles_dev, (size_t)NB * tile_h * tile_w * tile_c) == RS_OK, RS_ERR_HIP, "alloc tiles");
  reg("tiles", tiles_dev, DT_U8, {NB, tile_h, tile_w, tile_c}, 0);
  PreprocParams pp;
  memset(&pp, 0, sizeof pp);
  pp.need_h = net_w != tile_w;
  pp.need_v = net_h != tile_h;
  {
    const int ksh = rs_resize_coeffs(tile_w, net_w, nullptr, nullptr);
    const int ksv = rs_resize_coeffs(tile_h, net_h, nullptr, nullptr);
    std::vector<int32_t> hb(net_w * 2), hk((size_t)net_w * ksh), vb(net_h * 2), vk((size_t)net_h * ksv);
    rs_resize_coeffs(tile_w, net_w, hb.data(), hk.data());
    rs_resize_coeffs(tile_h, net_h, vb.data(), vk.data());
    int *dhb, *dhk, *dvb, *dvk;
    if ((rc = alloc((void**)&dhb, hb.size() * 4))) return rc;
    if ((rc = alloc((void**)&dhk, hk.size() * 4))) return rc;
    if ((rc = alloc((void**)&dvb, vb.size() * 4))) return rc;
    if ((rc = alloc((void**)&dvk, vk.size() * 4))) return rc;
    RS_HIP(hipMemcpyAsync(dhb, hb.data(), hb.size() * 4, hipMemcpyHostToDevice, stream));
    RS_HIP(hipMemcpyAsync(dhk, hk.data(), hk.size() * 4, hipMemcpyHostToDevice, stream));
    RS_HIP(hipMemcpyAsync(dvb, vb.data(), vb.size() * 4, hipMemcpyHostToDevice, stream));
    RS_HIP(hipMemcpyAsync(dvk, vk.data(), vk.size() * 4, hipMemcpyHostToDevice, stream));
    RS_HIP(hipStreamSynchronize(stream));
    pp.hb = dhb; pp.hk = dhk; pp.vb = dvb; pp.vk = dvk; pp.ksh = ksh; pp.ksv = ksv;
  }
  Act x0;
  if ((rc = new_act(&x0, "net_input", NB, pad_h, pad_w, 4, 3))) return rc;
  pp.out = x0.p; pp.H = tile_h; pp.W = tile_w; pp.C = tile_c; pp.new_h = net_h; pp.new_w = net_w;
  pp.out_Hp = x0.Hp(); pp.out_Wp = x0.Wp(); pp.flip = S.flip_channels; pp.out_f32 = f32 ? 1 : (split ? 2 : 0);
  pp.out_lo = x0.lo;
  for (int c = 0; c < 4; ++c) { pp.mean[c] = S.pixel_mean[c]; pp.stdv[c] = S.pixel_std[c] == 0.f ? 1.f : S.pixel_std[c]; }
  {
    Stage st;
    st.name = "preprocess";
    st.bytes_per_image = (double)tile_h * tile_w * tile_c + (double)net_h * net_w * 8;
    pp.tiles = tiles_dev;
    const size_t x0_bytes = (size_t)NB * x0.Hp() * x0.Wp() * 4 * (f32 ? 4 : 2);       // one plane
    const int planes = split ? 2 : 1;
    st.fn = [this, pp, x0_bytes, planes](int n, hipStream_t s) mutable {
      if (img_new_h.empty()) {
        pp.N = n;
        return launch_preprocess(pp, s);
      }
      // images of different sizes in one canvas: zeros (the padding value of ImageList.from_tensors) outside each image
      RS_CHECK((int)img_new_h.size() >= n, RS_ERR_ARG, "per-image sizes set for %d images, batch of %d", (int)img_new_h.size(), n);
      RS_HIP(hipMemsetAsync((void*)pp.out, 0, x0_bytes * planes, s));
      for (int i = 0; i < n; ++i) {
        PreprocParams q = pp;
        ResizeTab th, tv;
        int rc;
        if ((rc = resize_tab(tile_w, img_new_w[i], tab_h, &th))) return rc;
        if ((rc = resize_tab(tile_h, img_new_h[i], tab_v, &tv))) return rc;
        q.N = 1;
        q.tiles = pp.tiles + (size_t)i * tile_h * tile_w * tile_c;
        q.out = (half_t*)((char*)pp.out + (size_t)i * (x0_bytes / max_batch));
        q.new_h = img_new_h[i]; q.new_w = img_new_w[i];
        q.need_h = q.new_w != tile_w; q.need_v = q.new_h != tile_h;
        q.hb = th.b; q.hk = th.k; q.ksh = th.ks; q.vb = tv.b; q.vk = tv.k; q.ksv = tv.ks;
        if ((rc = launch_preprocess(q, s))) return rc;
      }
      return RS_OK;
    };
    stages.push_back(st);
    if ((rc = alloc((void**)&img_hw_dev, (size_t)NB * 8))) return rc;
    if ((rc = set_image_sizes(nullptr, nullptr, 0))) return rc;
  }

  // ---- stem
  const std::string bu = "backbone.bottom_up.";
  const int h2 = pad_h / 2, w2 = pad_w / 2, h4 = pad_h / 4, w4 = pad_w / 4;
  Act stem, c1;
  if ((rc = new_act(&stem, "stem_conv", NB, h2, w2, S.stem_out_channels, 1))) return rc;
  if ((rc = new_act(&c1, "stem", NB, h4, w4, S.stem_out_channels, 1))) return rc;
  const BlobEntry* stem_w = findw(bu + "stem.conv1f");          // fragment-ordered copy of the 64 x 256 stem matrix
  const BlobEntry* stem_b = find(bu + "stem.conv1.b");
  if (fuse_stem && !f32 && !split && use_glds > 0 && S.stem_out_channels == 64 && x0.C == 4 && x0.pad == 3 && stem_w && stem_b &&
      (long long)stem_w->dims[0] * stem_w->dims[1] == 7 * 4 * 64 * 8 && (pad_h & 3) == 0 && (pad_w & 3) == 0) {
    // conv 7x7 s2 + FrozenBN + ReLU + max-pool 3x3 s2 in one launch (stem_fused.hip): the 400 x 400 x 64 map never reaches HBM
    StemPoolParams sp;
    memset(&sp, 0, sizeof sp);
    sp.in = x0.p; sp.wf = (const half_t*)stem_w->dev; sp.bias = (const float*)stem_b->dev; sp.out = c1.p;
    sp.in_Hp = x0.Hp(); sp.in_Wp = x0.Wp();
    sp.Hc = h2; sp.Wc = w2; sp.Hq = h4; sp.Wq = w4;
    Stage st;
    st.name = "stem.conv1+maxpool";
    st.flops_per_image = 2.0 * h2 * w2 * 49.0 * S.in_channels * 64;
    st.bytes_per_image = 2.0 * ((double)pad_h * pad_w * 4 + (double)h4 * w4 * 64);
    st.fn = [sp](int n, hipStream_t s) mutable { sp.N = n; g_last_conv_variant = 21; return launch_stem_pool(sp, s); };
    stages.push_back(st);
  } else if (fuse_stem && split && use_glds > 0 && S.stem_out_channels == 64 && x0.C == 4 && x0.pad == 3 && stem_w && stem_b && find(bu + "stem.conv1.wsi") &&
             (long long)stem_w->dims[0] * stem_w->dims[1] == 2 * 7 * 4 * 64 * 8 && (pad_h & 3) == 0 && (pad_w & 3) == 0) {
    // the same launch on hi + lo planes (stem_fused.hip stem_pool_split_kernel): bit-identical to the stand-alone split stem + split max-pool
    StemPoolSplitParams sp;
    memset(&sp, 0, sizeof sp);
    sp.in = x0.p; sp.in_lo = x0.lo; sp.wf = (const half_t*)stem_w->dev; sp.wscale = (const float*)find(bu + "stem.conv1.wsi")->dev;
    sp.bias = (const float*)stem_b->dev; sp.out = c1.p; sp.out_lo = c1.lo;
    sp.in_Hp = x0.Hp(); sp.in_Wp = x0.Wp();
    sp.Hc = h2; sp.Wc = w2; sp.Hq = h4; sp.Wq = w4;
    Stage st;
    st.name = "stem.conv1+maxpool";
    st.flops_per_image = 2.0 * h2 * w2 * 49.0 * S.in_channels * 64;
    st.bytes_per_image = 4.0 * ((double)pad_h * pad_w * 4 + (double)h4 * w4 * 64);
    st.fn = [sp](int n, hipStream_t s) mutable { sp.N = n; g_last_conv_variant = 21; return launch_stem_pool_split(sp, s); };
    stages.push_back(st);
  } else {
  if ((rc = add_conv("stem.conv1", bu + "stem.conv1", x0, stem, 7, 2, 3, true, nullptr, nullptr, S.in_channels))) return rc;
  {
    Stage st;
    st.name = "stem.maxpool";
    st.bytes_per_image = 2.0 * ((double)h2 * w2 + (double)h4 * w4) * S.stem_out_channels;
    const bool f = f32, sp = split;
    st.fn = [stem, c1, f, sp](int n, hipStream_t s) {
      if (sp) return launch_maxpool_split(stem.p, stem.lo, c1.p, c1.lo, n, stem.H, stem.W, c1.H, c1.W, c1.C, s);
      return f ? launch_maxpool_f32((const float*)stem.p, (float*)c1.p, n, stem.H, stem.W, c1.H, c1.W, c1.C, s)
               : launch_maxpool(stem.p, c1.p, n, stem.H, stem.W, c1.H, c1.W, c1.C, s);
    };
    stages.push_back(st);
  }
  }

  // ---- res2..res5
  Act cur = c1;
  Act res_out[4];
  Act t1_pre;              // conv1 output of the NEXT block when the previous block's fused tail already produced it
  bool have_t1 = false;
  int bott = 64, cout = S.res2_out_channels;
  int ch = h4, cw = w4;
  for (int si = 0; si < 4; ++si) {
    for (int bi = 0; bi < S.res_blocks[si]; ++bi) {
      const std::string nm = "res" + std::to_string(si + 2) + "." + std::to_string(bi);
      const std::string wn = bu + nm;
      const int stride = (bi == 0 && si > 0) ? 2 : 1;
      const int s1 = S.stride_in_1x1 ? stride : 1, s3 = S.stride_in_1x1 ? 1 : stride;
      const int oh = ch / stride, ow = cw / stride;
      Act t1, t2, sc, out;
      // Fused tail (bneck_fused.hip): identity-shortcut blocks of the 64- and 128-wide stages run conv2 + conv3 (+ the next block's conv1)
      // in one launch -- t2 is never materialised and the next conv1 reads `out` from registers.  fp16 inference engine only.
      // Block 0 of the stage has a projection shortcut from the 64-channel stem output at the same resolution: the tail then adds
      // Wsc . x0 as two more K steps instead of the identity residual (needs the folded conv3sc bias = conv3's + the shortcut's).
      const bool may_fuse = !frozen_fusions_only || si == 0;          // a trainer fuses only inside the frozen res2
      const bool fuse_bneck = this->fuse_bneck && may_fuse, fuse_shortcut = this->fuse_shortcut && may_fuse;
      const bool tail0 = !f32 && !split && fuse_bneck && fuse_shortcut && bi == 0 && bott == 64 && cout == 256 && stride == 1 && cur.C == 64 &&
                         findw(wn + ".conv3p") != nullptr && findw(wn + ".shortcut") != nullptr && find(wn + ".conv3sc.b") != nullptr;
      const bool tail = tail0 || (!f32 && !split && fuse_bneck && bi > 0 && (bott == 64 || bott == 128) && cout == 4 * bott && stride == 1 && findw(wn + ".conv3p") != nullptr);
      // Split-operand mode: the same chain on hi + lo planes (bneck_split.hip), identity-shortcut blocks only.
      const bool tail_s = split && fuse_bneck && rs_debug().conv_deep && use_glds > 0 && bi > 0 && (bott == 64 || bott == 128) && cout == 4 * bott && stride == 1 &&
                          findw(wn + ".conv3p") != nullptr && find(wn + ".conv3p.wsi") != nullptr;
      // ... and res2.0, whose projection shortcut reads the 64-channel stem output at the same resolution (conv3 | shortcut as one [256][128] operand)
      const bool tail0_s = split && fuse_bneck && fuse_shortcut && rs_debug().conv_deep && use_glds > 0 && bi == 0 && bott == 64 && cout == 256 && stride == 1 &&
                           cur.C == 64 && findw(wn + ".conv3scp") != nullptr && find(wn + ".conv3scp.wsi") != nullptr && find(wn + ".conv3sc.b") != nullptr;
      const bool tail_next = (tail || tail_s || tail0_s) && bi + 1 < S.res_blocks[si] &&
                             findw(bu + "res" + std::to_string(si + 2) + "." + std::to_string(bi + 1) + ".conv1p") != nullptr;
      if (have_t1) t1 = t1_pre;
      else if ((rc = new_act(&t1, nm + ".conv1", NB, ch / s1, cw / s1, bott, 1))) return rc;
      if (!tail && !tail_s && !tail0_s) { if ((rc = new_act(&t2, nm + ".conv2", NB, oh, ow, bott, 1))) return rc; }
      if ((rc = new_act(&out, bi == S.res_blocks[si] - 1 ? "res" + std::to_string(si + 2) : nm + ".out", NB, oh, ow, cout, 1))) return rc;
      const Act* resid = &cur;
      // Projection shortcut: in the fp16 path it is folded into conv3 as a second K source (one GEMM over
      // [conv2 out ; block input], no shortcut tensor written or re-read); the fp32 validation path and
      // RS_FUSE_SHORTCUT=0 keep the reference's two-convolution form.
      const bool proj = cur.C != cout;
      const bool fuse_sc = proj && !tail0 && !tail0_s && !f32 && fuse_shortcut && s3 == 1 && cur.C % 64 == 0 && findw(wn + ".conv3sc") != nullptr;
      if (proj && !fuse_sc && !tail0 && !tail0_s) {
        if ((rc = new_act(&sc, nm + ".shortcut", NB, oh, ow, cout, 1))) return rc;
        if ((rc = add_conv(nm + ".shortcut", wn + ".shortcut", cur, sc, 1, stride, 0, false, nullptr, nullptr, cur.C))) return rc;
        resid = &sc;
      }
      if (!have_t1) { if ((rc = add_conv(nm + ".conv1", wn + ".conv1", cur, t1, 1, s1, 0, true, nullptr, nullptr, cur.C))) return rc; }
      have_t1 = false;
      if (tail) {
        const std::string nn = "res" + std::to_string(si + 2) + "." + std::to_string(bi + 1);
        if (tail_next) { if ((rc = new_act(&t1_pre, nn + ".conv1", NB, oh, ow, bott, 1))) return rc; }
        const BlobEntry *w2 = findw(wn + ".conv2"), *b2 = find(wn + ".conv2.b"), *w3 = findw(wn + ".conv3p"), *b3 = find(wn + (tail0 ? ".conv3sc.b" : ".conv3.b"));
        const BlobEntry* wsc = tail0 ? findw(wn + ".shortcut") : nullptr;
        const BlobEntry *w1 = tail_next ? findw(bu + nn + ".conv1p") : nullptr, *b1 = tail_next ? find(bu + nn + ".conv1.b") : nullptr;
        RS_CHECK(w2 && b2 && w3 && b3 && (!tail_next || (w1 && b1)), RS_ERR_BLOB, "weights of the fused tail of %s missing", nm.c_str());
        RS_CHECK(w2->dims[0] == bott && w2->dims[1] == 9 * bott && w3->dims[0] == cout && w3->dims[1] == bott && (!w1 || (w1->dims[0] == bott && w1->dims[1] == cout)),
                 RS_ERR_BLOB, "fused tail of %s: weight shapes", nm.c_str());
        BneckParams bp;
        memset(&bp, 0, sizeof bp);
        bp.t1 = t1.p; bp.w2 = (const half_t*)w2->dev; bp.b2 = (const float*)b2->dev; bp.w3p = (const half_t*)w3->dev; bp.b3 = (const float*)b3->dev;
        bp.out = out.p;
        if (tail0) {
          RS_CHECK(wsc && wsc->dims[0] == 256 && wsc->dims[1] == 64, RS_ERR_BLOB, "fused tail of %s: shortcut weight shape", nm.c_str());
          bp.x0 = cur.p; bp.wsc = (const half_t*)wsc->dev;
        } else {
          bp.x = cur.p;
        }
        if (tail_next) { bp.w1p = (const half_t*)w1->dev; bp.b1 = (const float*)b1->dev; bp.t1n = t1_pre.p; }
        bp.H = oh; bp.W = ow; bp.Hp = out.Hp(); bp.Wp = out.Wp(); bp.CB = bott / 64;
        RS_CHECK(t1.pad == 1 && cur.pad == 1 && out.pad == 1 && t1.H == oh && cur.H == oh && t1.C == bott && cur.C == (tail0 ? 64 : cout), RS_ERR_ARG, "fused tail of %s: geometry", nm.c_str());
        const int mpi = oh * ow;
        Stage st;
        st.name = nm + (tail_next ? ".conv2+conv3+next.conv1" : ".conv2+conv3");
        st.flops_per_image = 2.0 * mpi * (9.0 * bott * bott + (double)bott * cout + (tail0 ? 64.0 * cout : 0.0) + (tail_next ? (double)cout * bott : 0.0));
        st.bytes_per_image = 2.0 * mpi * ((double)bott + (tail0 ? 64.0 : (double)cout) + cout + (tail_next ? (double)bott : 0.0));       // t1 + x (or x0) in, out (+ t1n) out
        st.fn = [bp, mpi](int n, hipStream_t s) mutable { bp.M = n * mpi; g_last_conv_variant = 13; return launch_bneck_tail(bp, s); };
        stages.push_back(st);
        have_t1 = tail_next;
      } else if (tail_s || tail0_s) {
        const std::string nn = "res" + std::to_string(si + 2) + "." + std::to_string(bi + 1);
        if (tail_next) { if ((rc = new_act(&t1_pre, nn + ".conv1", NB, oh, ow, bott, 1))) return rc; }
        const BlobEntry *w2 = findw(wn + ".conv2"), *b2 = find(wn + ".conv2.b"), *s2 = find(wn + ".conv2.wsi");
        const BlobEntry *w3 = findw(wn + (tail0_s ? ".conv3scp" : ".conv3p")), *b3 = find(wn + (tail0_s ? ".conv3sc.b" : ".conv3.b")),
                        *sc3 = find(wn + (tail0_s ? ".conv3scp.wsi" : ".conv3p.wsi"));
        const BlobEntry *w1 = tail_next ? findw(bu + nn + ".conv1p") : nullptr, *b1 = tail_next ? find(bu + nn + ".conv1.b") : nullptr,
                        *sc1 = tail_next ? find(bu + nn + ".conv1p.wsi") : nullptr;
        RS_CHECK(w2 && b2 && s2 && w3 && b3 && sc3 && (!tail_next || (w1 && b1 && sc1)), RS_ERR_BLOB, "weights of the fused split tail of %s missing", nm.c_str());
        const int k3 = tail0_s ? bott + 64 : bott;
        RS_CHECK(wrows(w2) == bott && w2->dims[1] == 9 * bott && wrows(w3) == cout && w3->dims[1] == k3 && (!w1 || (wrows(w1) == bott && w1->dims[1] == cout)),
                 RS_ERR_BLOB, "fused split tail of %s: weight shapes", nm.c_str());
        RS_CHECK(t1.pad == 1 && cur.pad == 1 && out.pad == 1 && t1.H == oh && cur.H == oh && t1.C == bott && cur.C == (tail0_s ? 64 : cout), RS_ERR_ARG, "fused split tail of %s: geometry", nm.c_str());
        BneckSplitParams bp;
        memset(&bp, 0, sizeof bp);
        bp.t1 = t1.p; bp.t1_lo = t1.lo;
        bp.w2 = (const half_t*)w2->dev; bp.w2_lo = (long long)bott * 9 * bott; bp.b2 = (const float*)b2->dev; bp.s2 = (const float*)s2->dev;
        bp.w3p = (const half_t*)w3->dev; bp.w3_lo = (long long)cout * k3; bp.b3 = (const float*)b3->dev; bp.s3 = (const float*)sc3->dev;
        if (tail0_s) { bp.x0 = cur.p; bp.x0_lo = cur.lo; } else { bp.x = cur.p; bp.x_lo = cur.lo; }
        bp.out = out.p; bp.out_lo = out.lo;
        if (tail_next) {
          bp.w1p = (const half_t*)w1->dev; bp.w1_lo = (long long)bott * cout; bp.b1 = (const float*)b1->dev; bp.s1 = (const float*)sc1->dev;
          bp.t1n = t1_pre.p; bp.t1n_lo = t1_pre.lo;
        }
        bp.H = oh; bp.W = ow; bp.Hp = out.Hp(); bp.Wp = out.Wp(); bp.CB = bott / 64;
        const int mpi = oh * ow;
        Stage st;
        st.name = nm + (tail_next ? ".conv2+conv3+next.conv1" : ".conv2+conv3");
        st.flops_per_image = 2.0 * mpi * (9.0 * bott * bott + (double)k3 * cout + (tail_next ? (double)cout * bott : 0.0));
        st.bytes_per_image = 4.0 * mpi * ((double)bott + (tail0_s ? 64.0 : (double)cout) + cout + (tail_next ? (double)bott : 0.0));       // two planes of t1 + x (or x0) in, out (+ t1n) out
        st.fn = [bp, mpi](int n, hipStream_t s) mutable { bp.M = n * mpi; g_last_conv_variant = 23; return launch_bneck_tail_split(bp, s); };
        stages.push_back(st);
        have_t1 = tail_next;
      } else if (fuse_sc) {
        if ((rc = add_conv(nm + ".conv2", wn + ".conv2", t1, t2, 3, s3, 1, true, nullptr, nullptr, bott))) return rc;
        if ((rc = add_conv(nm + ".conv3", wn + ".conv3sc", t2, out, 1, 1, 0, true, nullptr, nullptr, bott, 1, nullptr, &cur, stride))) return rc;
      } else {
        if ((rc = add_conv(nm + ".conv2", wn + ".conv2", t1, t2, 3, s3, 1, true, nullptr, nullptr, bott))) return rc;
        if ((rc = add_conv(nm + ".conv3", wn + ".conv3", t2, out, 1, 1, 0, true, resid, nullptr, bott))) return rc;
      }
      cur = out;
      ch = oh; cw = ow;
    }
    res_out[si] = cur;
    bott *= 2;
    cout *= 2;
  }

  // ---- FPN
  Act inner[4], P[5];
  const bool merge = merge_levels && !f32 && rs_debug().conv_deep && use_glds > 0;
  std::vector<DeferredConv> fpn_out(4), rpn_conv(S.num_levels);
  for (int l = 3; l >= 0; --l) {
    const std::string ln = std::to_string(l + 2);
    if ((rc = new_act(&inner[l], "inner" + ln, NB, res_out[l].H, res_out[l].W, 256, 1))) return rc;
    if ((rc = new_act(&P[l], "p" + ln, NB, res_out[l].H, res_out[l].W, 256, 1))) return rc;
    if ((rc = add_conv("fpn_lateral" + ln, "backbone.fpn_lateral" + ln, res_out[l], inner[l], 1, 1, 0, false, nullptr,
                       l < 3 ? &inner[l + 1] : nullptr, res_out[l].C))) return rc;
    if (merge) {
      if ((rc = add_conv("fpn_output" + ln, "backbone.fpn_output" + ln, inner[l], P[l], 3, 1, 1, false, nullptr, nullptr, 256, 1, nullptr, nullptr, 1, &fpn_out[l]))) return rc;
    } else {
      if ((rc = add_conv("fpn_output" + ln, "backbone.fpn_output" + ln, inner[l], P[l], 3, 1, 1, false, nullptr, nullptr, 256))) return rc;
    }
  }
  // the four output convolutions depend on the laterals only: one launch, largest map first
  if (merge && (rc = add_merged_convs("fpn_output2-5", fpn_out))) return rc;
  {
    const int h6 = (P[3].H - 1) / 2 + 1, w6 = (P[3].W - 1) / 2 + 1;
    if ((rc = new_act(&P[4], "p6", NB, h6, w6, 256, 1))) return rc;
    Stage st;
    st.name = "fpn.p6";
    Act a = P[3], b = P[4];
    const bool f = f32, sp = split;
    st.fn = [a, b, f, sp](int n, hipStream_t s) {
      if (sp) return launch_subsample2_split(a.p, a.lo, b.p, b.lo, n, a.H, a.W, b.H, b.W, 256, s);
      return f ? launch_subsample2_f32((const float*)a.p, (float*)b.p, n, a.H, a.W, b.H, b.W, 256, s)
               : launch_subsample2(a.p, b.p, n, a.H, a.W, b.H, b.W, 256, s);
    };
    stages.push_back(st);
  }

  // ---- RPN head
  const int A = S.num_anchors, L = S.num_levels;
  const int head_cs = (5 * A + 15) / 16 * 16;
  RpnParams rp;
  memset(&rp, 0, sizeof rp);
  Act rpn_t[RS_MAX_LEVELS];
  float* rpn_ho[RS_MAX_LEVELS];
  for (int l = 0; l < L; ++l) {
    float* ho = nullptr;
    if ((rc = alloc((void**)&ho, (size_t)NB * P[l].H * P[l].W * head_cs * 4))) return rc;
    reg("rpn_head" + std::to_string(l + 2), ho, DT_F32, {NB, P[l].H, P[l].W, head_cs}, 0);
    rpn_ho[l] = ho;
  }
  // inference engines: the 16-row head runs inside the epilogue of the merged 3x3 launch (conv_deep.hip, ConvParams::head_w), so the
  // 256-channel "rpn_conv" maps are never written
  const BlobEntry* headsp = findw("proposal_generator.rpn_head.headsp");
  const BlobEntry* headsp_si = split ? find("proposal_generator.rpn_head.headsp.wsi") : nullptr;
  const bool fuse_heads = merge && !frozen_fusions_only && rs_debug().fuse_rpn_heads && head_cs == 16 && headsp != nullptr && (!split || headsp_si != nullptr);
  for (int l = 0; l < L; ++l) {
    const std::string ln = std::to_string(l + 2);
    Act t;
    if ((rc = new_act(&t, "rpn_conv" + ln, NB, P[l].H, P[l].W, 256, 1))) return rc;   // halo 1: its gradient is the input of a 3x3 (training)
    if (merge) {
      if ((rc = add_conv("rpn.conv" + ln, "proposal_generator.rpn_head.conv", P[l], t, 3, 1, 1, true, nullptr, nullptr, 256, 1, nullptr, nullptr, 1, &rpn_conv[l]))) return rc;
      if (fuse_heads) {
        const BlobEntry* hb = find("proposal_generator.rpn_head.heads.b");
        RS_CHECK(hb && wrows(headsp) == 16 && (int)headsp->dims[1] == 256, RS_ERR_BLOB, "rpn head weights (chained order) missing or not 16 x 256");
        rpn_conv[l].p.head_w = (const half_t*)headsp->dev; rpn_conv[l].p.head_b = (const float*)hb->dev; rpn_conv[l].p.head_out = rpn_ho[l];
        if (split) { rpn_conv[l].p.head_w_lo = 16 * 256; rpn_conv[l].p.head_scale = (const float*)headsp_si->dev; }
        rpn_conv[l].flops += 2.0 * P[l].H * P[l].W * 256 * 5 * A;
        rpn_conv[l].bytes += (double)P[l].H * P[l].W * (head_cs * 4 - 256 * 2 * (split ? 2 : 1));       // the heads' output instead of the 256-channel map
      }
      if (l == L - 1 && (rc = add_merged_convs(fuse_heads ? "rpn.conv+heads2-6" : "rpn.conv2-6", rpn_conv))) return rc;
      rpn_t[l] = t;
      continue;
    }
    if ((rc = add_conv("rpn.conv" + ln, "proposal_generator.rpn_head.conv", P[l], t, 3, 1, 1, true, nullptr, nullptr, 256))) return rc;
    rpn_t[l] = t;
  }
  for (int l = 0; l < L; ++l) {
    const std::string ln = std::to_string(l + 2);
    const Act t = rpn_t[l];
    float* ho = rpn_ho[l];
    // 1x1 heads (objectness + deltas fused), fp32 out
    if (!fuse_heads) {
      const BlobEntry* w = findw("proposal_generator.rpn_head.heads");
      const BlobEntry* b = find("proposal_generator.rpn_head.heads.b");
      RS_CHECK(w && b, RS_ERR_BLOB, "rpn head weights missing");
      RS_CHECK(wrows(w) == head_cs, RS_ERR_BLOB, "rpn head rows %d != %d", wrows(w), head_cs);
      ConvParams p;
      memset(&p, 0, sizeof p);
      if ((rc = set_split(&p, "proposal_generator.rpn_head.heads", w, &t, nullptr, nullptr, nullptr, nullptr))) return rc;
      p.in = t.p; p.w = (const half_t*)w->dev; p.bias = (const float*)b->dev; p.out = ho;
      p.Ho = t.H; p.Wo = t.W; p.in_Hp = t.Hp(); p.in_Wp = t.Wp(); p.in_Cs = 256; p.in_off = t.pad; p.stride = 1;
      p.KH = p.KW = 1; p.Cin = 256; p.Kpad = (int)w->dims[1]; p.Cout = head_cs;
      p.out_Hp = t.H; p.out_Wp = t.W; p.out_Cs = head_cs; p.out_pad = 0; p.out_f32 = 1;
      const int mpi = t.H * t.W;
      const int glds = use_glds;
      Stage st;
      st.name = "rpn.heads" + ln;
      st.flops_per_image = 2.0 * mpi * 256 * 5 * A;
      st.bytes_per_image = (double)mpi * (256 * 2 + head_cs * 4);
      st.fn = [p, mpi, glds](int n, hipStream_t s) mutable { p.M = n * mpi; return launch_conv(p, s, 2, glds); };
      stages.push_back(st);
    }
    rp.head[l] = ho;
    rp.H[l] = P[l].H; rp.W[l] = P[l].W; rp.stride[l] = 4 << l;
    uint32_t* keys = nullptr;
    if ((rc = alloc((void**)&keys, (size_t)NB * P[l].H * P[l].W * A * 4 * 2))) return rc;   // keys + candidate list
    rp.keys[l] = keys;
    for (int a = 0; a < A; ++a)
      for (int d = 0; d < 4; ++d) rp.base[l][a][d] = S.cell_anchors[l][a][d];
    RS_CHECK((long long)P[l].H * P[l].W * A < (1 << 24), RS_ERR_UNSUPPORTED, "feature map too large for the RPN index select");
  }
  rp.offset = S.anchor_offset; rp.L = L; rp.A = A; rp.cs = head_cs; rp.topk = S.rpn_pre_nms_topk;
  rp.img_h = (float)net_h; rp.img_w = (float)net_w;
  rp.wx = S.rpn_bbox_reg_weights[0]; rp.wy = S.rpn_bbox_reg_weights[1]; rp.ww = S.rpn_bbox_reg_weights[2]; rp.wh = S.rpn_bbox_reg_weights[3];
  rp.scale_clamp = S.scale_clamp; rp.min_size = S.rpn_min_size;
  uint8_t* cand_keep = nullptr;
  if ((rc = alloc((void**)&rp.cand_boxes, (size_t)NB * L * 1024 * 16))) return rc;
  if ((rc = alloc((void**)&rp.cand_scores, (size_t)NB * L * 1024 * 4))) return rc;
  if ((rc = alloc((void**)&rp.cand_valid, (size_t)NB * L * 1024))) return rc;
  if ((rc = alloc((void**)&rp.cand_count, (size_t)NB * L * 4))) return rc;
  if ((rc = alloc((void**)&rp.cand_index, (size_t)NB * L * 1024 * 4))) return rc;
  if ((rc = alloc((void**)&cand_keep, (size_t)NB * L * 1024))) return rc;
  reg("rpn_cand_boxes", rp.cand_boxes, DT_F32, {NB, L, 1024, 4}, 0);
  reg("rpn_cand_scores", rp.cand_scores, DT_F32, {NB, L, 1024}, 0);
  reg("rpn_cand_valid", rp.cand_valid, DT_U8, {NB, L, 1024}, 0);
  reg("rpn_cand_count", rp.cand_count, DT_I32, {NB, L}, 0);
  reg("rpn_cand_index", rp.cand_index, DT_I32, {NB, L, 1024}, 0);
  reg("rpn_cand_keep", cand_keep, DT_U8, {NB, L, 1024}, 0);
  {
    Stage st;
    st.name = "rpn.select_decode";
    st.fn = [rp](int n, hipStream_t s) mutable { rp.N = n; return launch_rpn_select(rp, s); };
    stages.push_back(st);
  }
  {
    NmsParams np = {};
    np.boxes = rp.cand_boxes; np.count = rp.cand_count; np.valid = rp.cand_valid; np.keep = cand_keep; np.cap = 1024;
    np.thresh = S.rpn_nms_thresh;
    // suppression masks in global memory for launches of few segments (batch 1-3: launch_nms shares a segment's mask build between workgroups)
    if ((rc = alloc((void**)&np.scratch, (size_t)(NB * L < 32 ? NB * L : 32) * 1024 * 16 * 8))) return rc;
    Stage st;
    st.name = "rpn.nms";
    st.fn = [np, L](int n, hipStream_t s) { return launch_nms(np, n * L, s); };
    stages.push_back(st);
  }
  const int PC = 1024;   // proposal slots per image
  float *prop_boxes, *prop_scores;
  int *prop_count, *prop_level;
  if ((rc = alloc((void**)&prop_boxes, (size_t)NB * PC * 16))) return rc;
  if ((rc = alloc((void**)&prop_scores, (size_t)NB * PC * 4))) return rc;
  if ((rc = alloc((void**)&prop_level, (size_t)NB * PC * 4))) return rc;
  if ((rc = alloc((void**)&prop_count, (size_t)NB * 4))) return rc;
  reg("proposal_boxes", prop_boxes, DT_F32, {NB, PC, 4}, 0);
  reg("proposal_logits", prop_scores, DT_F32, {NB, PC}, 0);
  reg("proposal_level", prop_level, DT_I32, {NB, PC}, 0);
  reg("proposal_count", prop_count, DT_I32, {NB}, 0);
  // visiting order of box.roi_align (RpnMergeParams::prop_order): inference engines on the windowed fp16 kernel only -- a training
  // engine overwrites the proposal buffer with its sampled RoIs after this stage
  int* prop_order = nullptr;
  const bool roi_order = !f32 && !(split && rs_debug().roi_window != 1) && !g_trainer_unfused_shortcut && rs_debug().roi_order != 0;
  if (roi_order) {
    if ((rc = alloc((void**)&prop_order, (size_t)NB * PC * 4))) return rc;
    reg("proposal_order", prop_order, DT_I32, {NB, PC}, 0);
  }
  {
    RpnMergeParams mp = {};
    mp.cand_boxes = rp.cand_boxes; mp.cand_scores = rp.cand_scores; mp.keep = cand_keep; mp.cand_count = rp.cand_count;
    mp.L = L; mp.post_topk = S.rpn_post_nms_topk; mp.cap = PC;
    mp.prop_boxes = prop_boxes; mp.prop_scores = prop_scores; mp.prop_level = prop_level; mp.prop_count = prop_count;
    mp.prop_order = prop_order;
    Stage st;
    st.name = "rpn.merge";
    st.fn = [mp](int n, hipStream_t s) { return launch_rpn_merge(mp, n, s); };
    stages.push_back(st);
  }

  // ---- box head
  const int PR = S.box_pooler_resolution;
  RoiAlignParams ra;
  memset(&ra, 0, sizeof ra);
  for (int l = 0; l < 4; ++l) { ra.feat[l] = P[l].p; ra.H[l] = P[l].H; ra.W[l] = P[l].W; ra.scale[l] = 1.0f / (float)(4 << l); }
  ra.nlevels = 4; ra.C = 256; ra.f32 = f32 ? 1 : (split ? 2 : 0);
  for (int l = 0; l < 4; ++l) ra.feat_lo[l] = P[l].lo;
  Act boxfeat;   // [NB*PC] "images" of PR x PR x 256
  if ((rc = new_act(&boxfeat, "box_pooled", NB * PC, PR, PR, 256, 0))) return rc;
  int* box_level = nullptr;
  if ((rc = alloc((void**)&box_level, (size_t)NB * PC * 4))) return rc;
  reg("box_roi_level", box_level, DT_I32, {NB, PC}, 0);
  {
    RoiAlignParams q = ra;
    q.rois = prop_boxes; q.per_image_count = prop_count; q.slots_per_image = PC; q.out = boxfeat.p; q.out_lo = boxfeat.lo; q.P = PR; q.out_pad = 0;
    q.out_level = box_level;
    q.order = prop_order;
    Stage st;
    st.name = "box.roi_align";
    st.bytes_per_image = (double)PC * PR * PR * 256 * 2 * 2 * (split ? 2 : 1);
    st.fn = [q](int n, hipStream_t s) mutable { q.S = n * PC; return launch_roi_align(q, s); };
    stages.push_back(st);
  }
  // FC layers as 1x1 "convs" over a (M x 1) image
  const int FC = S.box_fc_dim;
  Act fin, f1, f2;
  fin.p = boxfeat.p; fin.lo = boxfeat.lo; fin.N = 1; fin.H = NB * PC; fin.W = 1; fin.C = PR * PR * 256; fin.pad = 0;
  if ((rc = new_act(&f1, "box_fc1", 1, NB * PC, 1, FC, 0))) return rc;
  if ((rc = new_act(&f2, "box_fc2", 1, NB * PC, 1, FC, 0))) return rc;
  auto add_fc = [&](const std::string& name, const std::string& wn, const Act& in, const Act& out, bool relu, float* out32, int rows32) -> int {
    const BlobEntry* w = findw(wn);
    const BlobEntry* b = find(wn + ".b");
    RS_CHECK(w && b, RS_ERR_BLOB, "weights for %s missing", wn.c_str());
    ConvParams p;
    memset(&p, 0, sizeof p);
    { int rc2 = set_split(&p, wn, w, &in, out32 ? nullptr : &out, nullptr, nullptr, nullptr); if (rc2) return rc2; }
    p.in = in.p; p.w = (const half_t*)w->dev; p.bias = (const float*)b->dev;
    p.Ho = NB * PC; p.Wo = 1; p.in_Hp = NB * PC; p.in_Wp = 1; p.in_Cs = in.C; p.stride = 1; p.KH = p.KW = 1; p.Cin = in.C;
    p.Kpad = (int)w->dims[1];
    p.out_Hp = NB * PC; p.out_Wp = 1; p.relu = relu;
    if (out32) { p.out = out32; p.Cout = rows32; p.out_Cs = rows32; p.out_f32 = 1; }
    else { p.out = out.p; p.Cout = out.C; p.out_Cs = out.C; }
    RS_CHECK(wrows(w) >= p.Cout && p.Kpad >= in.C, RS_ERR_BLOB, "%s: weight shape", wn.c_str());
    const int glds = use_glds;
    const int variant = out32 ? 2 : -1;
    Stage st;
    st.name = name;
    st.flops_per_image = 2.0 * PC * (double)in.C * p.Cout;
    st.bytes_per_image = (double)PC * (in.C * 2 * (split ? 2 : 1) + p.Cout * (out32 ? 4 : (split ? 4 : 2)));
    st.fn = [p, glds, variant](int n, hipStream_t s) mutable { p.M = n * PC; return launch_conv(p, s, variant, glds); };
    stages.push_back(st);
    return RS_OK;
  };
  if ((rc = add_fc("box.fc1", "roi_heads.box_head.fc1", fin, f1, true, nullptr, 0))) return rc;
  if ((rc = add_fc("box.fc2", "roi_heads.box_head.fc2", f1, f2, true, nullptr, 0))) return rc;
  const int K = S.num_classes;
  const int pred_cs = (5 * K + 1 + 15) / 16 * 16;
  float* pred = nullptr;
  if ((rc = alloc((void**)&pred, (size_t)NB * PC * pred_cs * 4))) return rc;
  reg("box_pred", pred, DT_F32, {NB, PC, pred_cs}, 0);
  if ((rc = add_fc("box.predictor", "roi_heads.box_predictor", f2, f2, false, pred, pred_cs))) return rc;

  D = S.detections_per_image;
  BoxCandParams bc;
  memset(&bc, 0, sizeof bc);
  bc.pred = pred; bc.prop_boxes = prop_boxes; bc.prop_count = prop_count; bc.K = K; bc.cap = PC; bc.cs = pred_cs;
  bc.wx = S.box_reg_weights[0]; bc.wy = S.box_reg_weights[1]; bc.ww = S.box_reg_weights[2]; bc.wh = S.box_reg_weights[3];
  bc.scale_clamp = S.scale_clamp; bc.img_h = (float)net_h; bc.img_w = (float)net_w; bc.score_thresh = S.score_thresh_test;
  uint8_t* seg_keep = nullptr;
  if ((rc = alloc((void**)&bc.dec_boxes, (size_t)NB * PC * K * 16))) return rc;
  if ((rc = alloc((void**)&bc.dec_scores, (size_t)NB * PC * K * 4))) return rc;
  if ((rc = alloc((void**)&bc.seg_boxes, (size_t)NB * K * 1024 * 16))) return rc;
  if ((rc = alloc((void**)&bc.seg_roi, (size_t)NB * K * 1024 * 4))) return rc;
  if ((rc = alloc((void**)&bc.seg_count, (size_t)NB * K * 4))) return rc;
  if ((rc = alloc((void**)&seg_keep, (size_t)NB * K * 1024))) return rc;
  reg("box_dec_boxes", bc.dec_boxes, DT_F32, {NB, PC, K, 4}, 0);
  reg("box_dec_scores", bc.dec_scores, DT_F32, {NB, PC, K}, 0);
  reg("box_seg_boxes", bc.seg_boxes, DT_F32, {NB, K, 1024, 4}, 0);
  reg("box_seg_roi", bc.seg_roi, DT_I32, {NB, K, 1024}, 0);
  reg("box_seg_count", bc.seg_count, DT_I32, {NB, K}, 0);
  reg("box_seg_keep", seg_keep, DT_U8, {NB, K, 1024}, 0);
  {
    Stage st;
    st.name = "box.candidates";
    st.fn = [bc](int n, hipStream_t s) { return launch_box_candidates(bc, n, s); };
    stages.push_back(st);
  }
  {
    NmsParams np = {};
    np.boxes = bc.seg_boxes; np.count = bc.seg_count; np.valid = nullptr; np.keep = seg_keep; np.cap = 1024; np.thresh = S.nms_thresh_test;
    if ((rc = alloc((void**)&np.scratch, (size_t)(NB * K < 32 ? NB * K : 32) * 1024 * 16 * 8))) return rc;
    Stage st;
    st.name = "box.nms";
    st.fn = [np, K](int n, hipStream_t s) { return launch_nms(np, n * K, s); };
    stages.push_back(st);
  }
  int* det_roi = nullptr;
  if ((rc = alloc((void**)&det_boxes_net, (size_t)NB * D * 16))) return rc;
  if ((rc = alloc((void**)&det_boxes, (size_t)NB * D * 16))) return rc;
  if ((rc = alloc((void**)&det_scores, (size_t)NB * D * 4))) return rc;
  if ((rc = alloc((void**)&det_classes, (size_t)NB * D * 4))) return rc;
  if ((rc = alloc((void**)&det_roi, (size_t)NB * D * 4))) return rc;
  if ((rc = alloc((void**)&det_count, (size_t)NB * 4))) return rc;
  reg("det_boxes_net", det_boxes_net, DT_F32, {NB, D, 4}, 0);
  reg("det_boxes", det_boxes, DT_F32, {NB, D, 4}, 0);
  reg("det_scores", det_scores, DT_F32, {NB, D}, 0);
  reg("det_classes", det_classes, DT_I32, {NB, D}, 0);
  reg("det_roi", det_roi, DT_I32, {NB, D}, 0);
  reg("det_count", det_count, DT_I32, {NB}, 0);
  {
    DetMergeParams dm;
    memset(&dm, 0, sizeof dm);
    dm.dec_boxes = bc.dec_boxes; dm.dec_scores = bc.dec_scores; dm.seg_roi = bc.seg_roi; dm.seg_count = bc.seg_count; dm.keep = seg_keep;
    dm.K = K; dm.cap = PC; dm.dets_per_image = D;
    dm.scale_x = (float)((double)tile_w / (double)net_w);
    dm.scale_y = (float)((double)tile_h / (double)net_h);
    dm.out_w = (float)tile_w; dm.out_h = (float)tile_h;
    dm.det_boxes_net = det_boxes_net; dm.det_boxes = det_boxes; dm.det_scores = det_scores; dm.det_classes = det_classes;
    dm.det_roi = det_roi; dm.det_count = det_count;
    Stage st;
    st.name = "box.merge_postprocess";
    st.fn = [dm](int n, hipStream_t s) { return launch_det_merge(dm, n, s); };
    stages.push_back(st);
  }

  // ---- mask head
  if (S.mask_on) {
    const int MR = S.mask_pooler_resolution, R = NB * D;
    int *slot_list, *det_total;
    if ((rc = alloc((void**)&slot_list, (size_t)R * 4))) return rc;
    if ((rc = alloc((void**)&det_total, 16))) return rc;
    reg("det_slot_list", slot_list, DT_I32, {R}, 0);
    reg("det_total", det_total, DT_I32, {1}, 0);
    {
      Stage st;
      st.name = "mask.compact";
      int* dc = det_count;
      const int Dc = D;
      st.fn = [dc, Dc, slot_list, det_total](int n, hipStream_t s) { return launch_det_compact(dc, n, Dc, slot_list, det_total, s); };
      stages.push_back(st);
    }
    Act mx;
    if ((rc = new_act(&mx, "mask_pooled", R, MR, MR, 256, 1))) return rc;
    {
      RoiAlignParams q = ra;
      q.rois = det_boxes_net; q.slot_list = slot_list; q.n_entries = det_total; q.slots_per_image = D;
      q.out = mx.p; q.out_lo = mx.lo; q.P = MR; q.out_pad = 1;
      Stage st;
      st.name = "mask.roi_align";
      st.bytes_per_image = (double)D * MR * MR * 256 * 2 * 2 * (split ? 2 : 1);
      const int Dc = D;
      st.fn = [q, Dc](int n, hipStream_t s) mutable { q.S = n * Dc; return launch_roi_align(q, s); };
      stages.push_back(st);
    }
    Act curm = mx;
    for (int i = 0; i < S.mask_num_conv; ++i) {
      Act o;
      const std::string nm = "mask_fcn" + std::to_string(i + 1);
      if ((rc = new_act(&o, nm, R, MR, MR, 256, 1))) return rc;
      if ((rc = add_conv("mask.fcn" + std::to_string(i + 1), "roi_heads.mask_head." + nm, curm, o, 3, 1, 1, true, nullptr, nullptr, 256, D, det_total))) return rc;
      curm = o;
    }
    if ((rc = alloc((void**)&mask_probs, (size_t)R * RS_MASK_SIDE * RS_MASK_SIDE * 4))) return rc;
    reg("mask_probs", mask_probs, DT_F32, {NB, D, RS_MASK_SIDE, RS_MASK_SIDE}, 0);
    const BlobEntry* dw = findw("roi_heads.mask_head.deconv");
    const BlobEntry* db = find("roi_heads.mask_head.deconv.b");
    const BlobEntry* pw = find("roi_heads.mask_head.predictor.w");
    const BlobEntry* pb = find("roi_heads.mask_head.predictor.b");
    RS_CHECK(dw && db && wrows(dw) == 1024, RS_ERR_BLOB, "deconv weights missing / wrong rows");
    RS_CHECK(pw && pb && pw->dtype == DT_F32, RS_ERR_BLOB, "mask predictor weights missing");
    const bool fuse = f32 ? false : rs_debug().fuse_mask_predictor != 0;
    ConvParams dp;
    memset(&dp, 0, sizeof dp);
    if ((rc = set_split(&dp, "roi_heads.mask_head.deconv", dw, &curm, nullptr, nullptr, nullptr, nullptr))) return rc;
    dp.in = curm.p; dp.w = (const half_t*)dw->dev; dp.bias = (const float*)db->dev;
    dp.Ho = MR; dp.Wo = MR; dp.in_Hp = curm.Hp(); dp.in_Wp = curm.Wp(); dp.in_Cs = 256; dp.in_off = 1; dp.stride = 1; dp.KH = dp.KW = 1;
    dp.Cin = 256; dp.Kpad = (int)dw->dims[1]; dp.Cout = 256; dp.out_Hp = 2 * MR; dp.out_Wp = 2 * MR; dp.out_Cs = 256; dp.out_pad = 0; dp.relu = 1;
    dp.m_count = det_total; dp.m_mul = MR * MR;
    const int per_roi = MR * MR;
    if (fuse) {
      // deconv 2x2 s2 + ReLU + predictor 1x1 (predicted class only) in one launch: the 28x28x256 map (40 MB per
      // tile) is never written; a small kernel then adds the class bias and applies the sigmoid in place.
      dp.mode = 2; dp.out = nullptr;
      dp.dot_w = (const float*)pw->dev; dp.dot_cls = det_classes; dp.dot_slot = slot_list; dp.dot_out = mask_probs;
      dp.dot_k = (int)pw->dims[0];
      const int glds = use_glds, Dc = D;
      const bool sp = split;
      float* probs = mask_probs;
      const size_t zero_per_tile = (size_t)D * RS_MASK_SIDE * RS_MASK_SIDE * 4;
      Stage st;
      st.name = "mask.deconv_predict";
      st.flops_per_image = 2.0 * D * per_roi * 256 * 1024 + 2.0 * D * RS_MASK_SIDE * RS_MASK_SIDE * 256;
      st.bytes_per_image = (double)D * per_roi * 256 * 2 * (split ? 2 : 1) + (double)D * RS_MASK_SIDE * RS_MASK_SIDE * 4;
      st.fn = [dp, per_roi, Dc, glds, probs, zero_per_tile, sp](int n, hipStream_t s) mutable {
        RS_HIP(hipMemsetAsync(probs, 0, zero_per_tile * n, s));
        dp.M = n * Dc * per_roi;
        // conv_wreg.hip (22: persistent, a (dy, dx) group's weights in registers) or conv_igemm's tile where one workgroup holds all 256
        // channels of a group (14 / 10); the two give the same bits
        const int dv = rs_debug().deconv_variant;
        return launch_conv(dp, s, (dv == 22 && (sp || !(rs_debug().conv_wreg && glds > 0))) ? 14 : dv, glds);
      };
      stages.push_back(st);
      MaskPredictParams mp;
      mp.f32 = 0;
      mp.in = nullptr; mp.w = nullptr; mp.b = (const float*)pb->dev; mp.slot_list = slot_list; mp.det_classes = det_classes;
      mp.n_entries = det_total; mp.out = mask_probs; mp.S = RS_MASK_SIDE;
      Stage st2;
      st2.name = "mask.bias_sigmoid";
      st2.fn = [mp, Dc](int n, hipStream_t s) { return launch_mask_sigmoid(mp, n * Dc, s); };
      stages.push_back(st2);
    } else {
      Act dec;
      if ((rc = new_act(&dec, "mask_deconv", R, 2 * MR, 2 * MR, 256, 0))) return rc;
      dp.mode = 1; dp.out = dec.p; dp.out_lo = dec.lo;
      {
        const int glds = use_glds, Dc = D;
        Stage st;
        st.name = "mask.deconv";
        st.flops_per_image = 2.0 * D * per_roi * 256 * 1024;
        st.bytes_per_image = (double)D * per_roi * 256 * 2 * 5;
        st.fn = [dp, per_roi, Dc, glds](int n, hipStream_t s) mutable { dp.M = n * Dc * per_roi; return launch_conv(dp, s, -1, glds); };
        stages.push_back(st);
      }
      MaskPredictParams mp;
      mp.in = dec.p; mp.w = (const float*)pw->dev; mp.b = (const float*)pb->dev; mp.slot_list = slot_list; mp.det_classes = det_classes;
      mp.n_entries = det_total; mp.out = mask_probs; mp.S = RS_MASK_SIDE; mp.f32 = f32 ? 1 : (split ? 2 : 0); mp.in_lo = dec.lo;
      Stage st;
      st.name = "mask.predict_sigmoid";
      st.bytes_per_image = (double)D * RS_MASK_SIDE * RS_MASK_SIDE * (256 * 2 + 4);
      const int Dc = D;
      st.fn = [mp, Dc](int n, hipStream_t s) { return launch_mask_predict(mp, n * Dc, s); };
      stages.push_back(st);
    }
    const int Wb = (tile_w + 7) / 8;
    if ((rc = alloc((void**)&masks, (size_t)R * tile_h * Wb))) return rc;
    reg("masks", masks, DT_U8, {NB, D, tile_h, Wb}, 0);
    if ((rc = alloc((void**)&crop_data, (size_t)R * tile_h * Wb))) return rc;
    if ((rc = alloc((void**)&crop_rects, (size_t)R * 16))) return rc;
    if ((rc = alloc((void**)&crop_offsets, (size_t)R * 4))) return rc;
    if ((rc = alloc((void**)&crop_total, 16))) return rc;
    {
      PasteParams pm;
      pm.probs = mask_probs; pm.det_boxes = det_boxes; pm.slot_list = slot_list; pm.n_entries = det_total; pm.out = masks;
      pm.S = RS_MASK_SIDE; pm.out_h = tile_h; pm.out_w = tile_w; pm.threshold = S.mask_threshold;
      Stage st;
      st.name = "mask.paste";
      st.bytes_per_image = (double)D * tile_h * Wb;
      const int Dc = D;
      st.fn = [pm, Dc](int n, hipStream_t s) { return launch_paste_masks(pm, n * Dc, s); };
      stages.push_back(st);
    }
  }
  RS_HIP(hipStreamSynchronize(stream));
  return RS_OK;
}

// Phase and stream of every stage.  With RS_SIDE_STREAM=1 (default) the detection glue runs on a side stream:
// alone it changes nothing (the hand-off events keep the order), but two engines that share the wide stream can
// then hide one batch's glue behind the other batch's convolutions (rs_engine_infer_phase).
int rs_engine::assign_phases() {
  const bool side = rs_debug().side_stream != 0 && !use_graph;
  static const char* kNarrow[] = {"rpn.select_decode", "rpn.nms", "rpn.merge", "box.candidates", "box.nms",
                                  "box.merge_postprocess", "mask.compact"};
  int phase = 0;
  const bool roi_narrow = rs_debug().narrow_roialign != 0;
  for (Stage& st : stages) {
    if (st.name.rfind("box.", 0) == 0 && phase < 1) phase = 1;
    if (st.name.rfind("mask.", 0) == 0 && st.name != "mask.compact" && phase < 2) phase = 2;
    st.phase = phase;
    for (const char* nm : kNarrow) if (side && st.name == nm) st.narrow = true;
    // RoIAlign is gather-bound (L1/L2 request rate), not MFMA-bound: with RS_NARROW_ROIALIGN=1 it runs on the side stream and
    // shares the chip with the other lane's convolutions (+1 % tiles/s measured).  Off by default: the convolutions it
    // overlaps run ~12 % longer each, which blurs the per-kernel roofline measurement for a 1 % gain.
    if (side && roi_narrow && (st.name == "box.roi_align" || st.name == "mask.roi_align")) st.narrow = true;
    // (measured and left on the wide stream: mask.paste, preprocess, box.predictor, mask.bias_sigmoid -- 0 to -1 %)
  }
  if (!side) return RS_OK;
  RS_HIP(hipStreamCreateWithFlags(&narrow, hipStreamNonBlocking));
  RS_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  bool prev = false;
  for (Stage& st : stages) {
    if (st.narrow != prev) RS_HIP(hipEventCreateWithFlags(&st.handoff, hipEventDisableTiming));
    prev = st.narrow;
  }
  return RS_OK;
}

int rs_engine::run_stages(int n, bool record, int phase, bool all_wide) {
  for (size_t si = 0; si < stages.size(); ++si) {
    Stage& st = stages[si];
    if (phase >= 0 && st.phase != phase) continue;
    if (all_wide) {                     // one-tile graph capture: every stage on the wide stream, in list order (the side stream hides nothing at one tile)
      g_last_conv_variant = -2;
      int rc = st.fn(n, stream);
      if (rc) return rc;
      st.variant = g_last_conv_variant;
      continue;
    }
    if (st.narrow != on_narrow) {       // hand the dependency chain over to the other stream
      hipStream_t from = on_narrow ? narrow : stream, to = st.narrow ? narrow : stream;
      RS_HIP(hipEventRecord(st.handoff, from));
      RS_HIP(hipStreamWaitEvent(to, st.handoff, 0));
      on_narrow = st.narrow;
    }
    hipStream_t ss = st.narrow ? narrow : stream;
    const bool pooled = record && profiling >= 2 && ev_used < ev_pool.size();
    if (record && profiling == 1) RS_HIP(hipEventRecord(ev0, ss));
    if (pooled) RS_HIP(hipEventRecord(ev_pool[ev_used].first, ss));
    g_last_conv_variant = -2;
    int rc = st.fn(n, ss);
    if (rc) return rc;
    st.variant = g_last_conv_variant;
    if (pooled) {
      RS_HIP(hipEventRecord(ev_pool[ev_used].second, ss));
      ev_stage[ev_used] = (int)si;
      ev_batch[ev_used] = n;
      ++ev_used;
    }
    if (record && profiling == 1) {
      RS_HIP(hipEventRecord(ev1, ss));
      RS_HIP(hipEventSynchronize(ev1));
      float ms = 0.f;
      RS_HIP(hipEventElapsedTime(&ms, ev0, ev1));
      st.ms_total += ms;
      st.calls += 1;
      st.last_flops = st.flops_per_image * n;
      st.last_bytes = st.bytes_per_image * n;
    }
  }
  if ((phase < 0 || phase == RS_NUM_PHASES - 1) && on_narrow) {   // the forward ended on the side stream: join
    RS_HIP(hipEventRecord(ev_join, narrow));
    RS_HIP(hipStreamWaitEvent(stream, ev_join, 0));
    on_narrow = false;
  }
  return RS_OK;
}

// One forward.  The ~110 launches of a forward are replayed from a hipGraph (captured per batch size
// after one eager warm-up run, which also performs the one-time hipFuncSetAttribute calls); forwards
// that are being event-profiled run eagerly so every launch can be bracketed.
int rs_engine::run(const uint8_t* tiles, int n, int phase) {
  RS_CHECK(n >= 1 && n <= max_batch, RS_ERR_ARG, "batch %d outside [1, %d]", n, max_batch);
  RS_CHECK(phase >= -1 && phase < RS_NUM_PHASES, RS_ERR_ARG, "phase %d outside [-1, %d)", phase, RS_NUM_PHASES);
  if (phase <= 0) {
    if (tiles != tiles_dev) {
      // tiles already resident elsewhere on the device: stage them into the engine's input buffer
      RS_HIP(hipMemcpyAsync(tiles_dev, tiles, (size_t)n * tile_h * tile_w * tile_c, hipMemcpyDeviceToDevice, stream));
    }
    const long long idx = forward_index++;
    cur_record = profiling == 1 || profiling == 2 || (profiling == 3 && (idx & 3) == 0);
  }
  if (copy_pending && (phase < 0 || phase == 1)) {
    // the box head rewrites the result buffers: let an outstanding rs_engine_fetch_async of the previous forward finish first
    RS_HIP(hipStreamWaitEvent(stream, ev_copied, 0));
    copy_pending = false;
  }
  const bool record = cur_record;
  // One tile per call (what the reference's predictor(im) loop submits): the ~110 launches replay from a hipGraph captured with every stage on the wide
  // stream -- same kernels, same order, same bits; 2.25 -> 2.13 ms fp16, 3.76 -> 3.67 ms split (tools/ubench/single_tile_latency.py).  Larger batches stay eager.
  // (inference engines on the one network geometry only: a trainer's forward engine re-reads host state per step -- per-image sizes, their resize tables)
  const bool small = n == 1 && rs_debug().graph_small != 0 && !on_narrow && img_new_h.empty() && !frozen_fusions_only;
  if (phase >= 0 || record || !(use_graph || small) || !warmed.count(n)) {
    int rc = run_stages(n, record, phase);
    if (rc) return rc;
    if (phase < 0) warmed.insert(n);
    return RS_OK;
  }
  auto it = graphs.find(n);
  if (it == graphs.end()) {
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    RS_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    int rc = run_stages(n, false, -1, !use_graph);
    hipError_t he = hipStreamEndCapture(stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    RS_HIP(he);
    RS_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    it = graphs.emplace(n, ge).first;
  }
  RS_HIP(hipGraphLaunch(it->second, stream));
  return RS_OK;
}

int rs_engine::resolve_profile() {
  if (ev_used == 0) return RS_OK;
  RS_HIP(hipStreamSynchronize(stream));
  if (narrow) RS_HIP(hipStreamSynchronize(narrow));
  for (size_t i = 0; i < ev_used; ++i) {
    float ms = 0.f;
    RS_HIP(hipEventElapsedTime(&ms, ev_pool[i].first, ev_pool[i].second));
    Stage& st = stages[ev_stage[i]];
    st.ms_total += ms;
    st.calls += 1;
    st.last_flops = st.flops_per_image * ev_batch[i];
    st.last_bytes = st.bytes_per_image * ev_batch[i];
  }
  ev_used = 0;
  return RS_OK;
}

// ===================================================================================== C ABI
extern "C" {

const char* rs_last_error(void) { return g_err; }
int rs_abi_version(void) { return RS_ABI_VERSION; }

// rs_fdiv (common.h) as an operator, for its test: out[i] = a[i] / b[i] through the device code's division
__global__ void fdiv_kernel(const float* a, const float* b, float* out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = rs_fdiv(a[i], b[i]);
}
int rs_op_fdiv(const float* a, const float* b, float* out, int64_t n, void* stream) {
  RS_CHECK(a && b && out && n > 0, RS_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(fdiv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long long)n);
  RS_HIP(hipGetLastError());
  return RS_OK;
}

int rs_memcpy_d2h(void* dst, const void* src, size_t n) {
  RS_HIP(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
  return RS_OK;
}
int rs_memcpy_h2d(void* dst, const void* src, size_t n) {
  RS_HIP(hipMemcpy(dst, src, n, hipMemcpyHostToDevice));
  return RS_OK;
}

int rs_engine_create(const rs_spec* spec, const void* weights, size_t nbytes, int device_ordinal, int max_batch,
                     int tile_h, int tile_w, int tile_c, void* stream, rs_engine** out) {
  RS_CHECK(spec && weights && out, RS_ERR_ARG, "null argument");
  RS_CHECK(spec->struct_size == (int32_t)sizeof(rs_spec), RS_ERR_ARG, "rs_spec size mismatch: caller %d, library %d", spec->struct_size, (int)sizeof(rs_spec));
  RS_CHECK(max_batch >= 1 && tile_h >= 32 && tile_w >= 32, RS_ERR_ARG, "bad batch/tile shape");
  int ndev = 0;
  RS_HIP(hipGetDeviceCount(&ndev));
  RS_CHECK(ndev > 0, RS_ERR_HIP, "no HIP device visible: the engine has no CPU fallback");
  RS_CHECK(device_ordinal >= 0 && device_ordinal < ndev, RS_ERR_ARG, "device %d of %d", device_ordinal, ndev);
  RS_HIP(hipSetDevice(device_ordinal));
  rs_engine* e = new rs_engine();
  e->spec = *spec;
  e->device = device_ordinal;
  e->max_batch = max_batch; e->tile_h = tile_h; e->tile_w = tile_w; e->tile_c = tile_c;
  rs_debug_reload();
  e->use_glds = rs_debug().use_glds;
  e->f32 = spec->precision == 1;
  e->split = spec->precision == 2;
  if (e->f32) e->use_glds = -1;
  if (e->split && (e->use_glds <= 0 || g_trainer_unfused_shortcut)) {
    rs_set_error("precision 2 (split operands) needs LDS-DMA staging and is an inference mode");
    delete e;
    return RS_ERR_UNSUPPORTED;
  }
  e->fuse_shortcut = rs_debug().fuse_shortcut;
  e->fuse_bneck = rs_debug().fuse_bneck;
  e->fuse_stem = rs_debug().fuse_stem;
  e->merge_levels = rs_debug().merge_levels;
  // the training engine differentiates every convolution of res3..res5 / FPN / heads separately and needs every layer output there; the frozen
  // stem and res2 (FREEZE_AT 2, what rs_trainer implements) keep the inference engine's fused stem and fused bottleneck tails
  // (the multi-map launches of the FPN output convolutions and of the RPN 3x3 stay: every map still gets its own output tensor; only the RPN heads
  //  leave the 3x3's epilogue, because the trainer differentiates through the 256-channel rpn_conv maps)
  if (g_trainer_unfused_shortcut) e->frozen_fusions_only = true;
  e->use_graph = rs_debug().use_graph;   // measured: replay == eager (11.54 ms/step either way), so off by default
  if (stream) { e->stream = (hipStream_t)stream; e->own_stream = false; }
  else {
    hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (he != hipSuccess) { rs_set_error("hipStreamCreate: %s", hipGetErrorString(he)); delete e; return RS_ERR_HIP; }
    e->own_stream = true;
  }
  int rc = RS_OK;
  if (hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess) { rs_set_error("hipEventCreate failed"); rc = RS_ERR_HIP; }
  if (!rc) rc = e->parse_blob(weights, nbytes);
  if (!rc) rc = e->build();
  if (!rc) rc = e->assign_phases();
  if (rc) { rs_engine_destroy(e); return rc; }
  *out = e;
  return RS_OK;
}

void rs_engine_destroy(rs_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->narrow) { (void)hipStreamSynchronize(e->narrow); (void)hipStreamDestroy(e->narrow); }
  if (e->copy_stream) { (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamDestroy(e->copy_stream); }
  if (e->ev_crop_hdr) (void)hipEventDestroy(e->ev_crop_hdr);
  if (e->h_crop_total) (void)hipHostFree(e->h_crop_total);
  if (e->ev_results) (void)hipEventDestroy(e->ev_results);
  if (e->ev_copied) (void)hipEventDestroy(e->ev_copied);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  for (Stage& st : e->stages) if (st.handoff) (void)hipEventDestroy(st.handoff);
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->blob_dev) (void)hipFree(e->blob_dev);
  for (auto& kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
  for (auto& pr : e->ev_pool) { if (pr.first) (void)hipEventDestroy(pr.first); if (pr.second) (void)hipEventDestroy(pr.second); }
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int rs_engine_infer_device(rs_engine* e, const uint8_t* tiles_dev, int n) {
  RS_CHECK(e && tiles_dev, RS_ERR_ARG, "null argument");
  RS_HIP(hipSetDevice(e->device));
  return e->run(tiles_dev, n);
}

int rs_engine_infer_phase(rs_engine* e, const uint8_t* tiles_dev, int n, int phase) {
  RS_CHECK(e && tiles_dev, RS_ERR_ARG, "null argument");
  RS_CHECK(phase >= 0 && phase < RS_NUM_PHASES, RS_ERR_ARG, "phase %d outside [0, %d)", phase, RS_NUM_PHASES);
  RS_HIP(hipSetDevice(e->device));
  return e->run(tiles_dev, n, phase);
}
int rs_engine_phase_count(void) { return RS_NUM_PHASES; }

// Diagnostic: enqueue, on the engine's wide stream, only the stages whose name contains `substr` (in list order, on the data the last forward left).
int rs_debug_run_stages_matching(rs_engine* e, const char* substr, int n) {
  RS_CHECK(e && substr && n >= 1 && n <= e->max_batch, RS_ERR_ARG, "bad argument");
  RS_HIP(hipSetDevice(e->device));
  for (Stage& st : e->stages)
    if (st.name.find(substr) != std::string::npos) {
      int rc = st.fn(n, e->stream);
      if (rc) return rc;
    }
  return RS_OK;
}

int rs_engine_sync(rs_engine* e) {
  RS_CHECK(e, RS_ERR_ARG, "null engine");
  RS_HIP(hipStreamSynchronize(e->stream));
  if (e->narrow) RS_HIP(hipStreamSynchronize(e->narrow));
  return RS_OK;
}

int rs_engine_fetch(rs_engine* e, int n, rs_dets* o) {
  RS_CHECK(e && o && o->count, RS_ERR_ARG, "null argument");
  RS_CHECK(n >= 1 && n <= e->max_batch, RS_ERR_ARG, "batch %d", n);
  const int D = e->D;
  hipStream_t s = e->stream;
  RS_HIP(hipMemcpyAsync(o->count, e->det_count, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  if (o->boxes) RS_HIP(hipMemcpyAsync(o->boxes, e->det_boxes, (size_t)n * D * 16, hipMemcpyDeviceToHost, s));
  if (o->scores) RS_HIP(hipMemcpyAsync(o->scores, e->det_scores, (size_t)n * D * 4, hipMemcpyDeviceToHost, s));
  if (o->classes) RS_HIP(hipMemcpyAsync(o->classes, e->det_classes, (size_t)n * D * 4, hipMemcpyDeviceToHost, s));
  if (o->masks) {
    RS_CHECK(e->masks, RS_ERR_ARG, "masks requested but MASK_ON is false");
    RS_HIP(hipMemcpyAsync(o->masks, e->masks, (size_t)n * D * e->tile_h * ((e->tile_w + 7) / 8), hipMemcpyDeviceToHost, s));
  }
  if (o->mask_probs) {
    RS_CHECK(e->mask_probs, RS_ERR_ARG, "mask_probs requested but MASK_ON is false");
    RS_HIP(hipMemcpyAsync(o->mask_probs, e->mask_probs, (size_t)n * D * RS_MASK_SIDE * RS_MASK_SIDE * 4, hipMemcpyDeviceToHost, s));
  }
  RS_HIP(hipStreamSynchronize(s));
  return RS_OK;
}

// ---- asynchronous host interface: pinned buffers, H2D on the forward stream, D2H on a copy stream behind an event
void* rs_host_alloc(size_t nbytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, nbytes ? nbytes : 16, hipHostMallocDefault) != hipSuccess) { rs_set_error("hipHostMalloc(%zu) failed", nbytes); return nullptr; }
  return p;
}
void rs_host_free(void* p) { if (p) (void)hipHostFree(p); }

// Pin an existing host allocation (the CLI's shared-memory slab of decoded tiles) so that rs_engine_upload_async can copy straight out
// of it: a pageable source makes hipMemcpyAsync stage the bytes through the runtime on the calling thread.
int rs_host_register(void* p, size_t bytes) {
  RS_CHECK(p && bytes > 0, RS_ERR_ARG, "bad argument");
  RS_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
  return RS_OK;
}
int rs_host_unregister(void* p) {
  RS_CHECK(p, RS_ERR_ARG, "bad argument");
  RS_HIP(hipHostUnregister(p));
  return RS_OK;
}

int rs_engine_upload_async(rs_engine* e, const uint8_t* tiles_host, int n) {
  RS_CHECK(e && tiles_host && n >= 1 && n <= e->max_batch, RS_ERR_ARG, "bad argument");
  RS_HIP(hipMemcpyAsync(e->tiles_dev, tiles_host, (size_t)n * e->tile_h * e->tile_w * e->tile_c, hipMemcpyHostToDevice, e->stream));
  return RS_OK;
}

int rs_engine_fetch_async(rs_engine* e, int n, rs_dets* o) {
  RS_CHECK(e && o && o->count && n >= 1 && n <= e->max_batch, RS_ERR_ARG, "bad argument");
  if (!e->copy_stream) {
    RS_HIP(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    RS_HIP(hipEventCreateWithFlags(&e->ev_results, hipEventDisableTiming));
    RS_HIP(hipEventCreateWithFlags(&e->ev_copied, hipEventDisableTiming));
  }
  const int D = e->D;
  hipStream_t s = e->copy_stream;
  RS_HIP(hipEventRecord(e->ev_results, e->stream));          // everything enqueued so far for this engine (its last phase included)
  RS_HIP(hipStreamWaitEvent(s, e->ev_results, 0));
  RS_HIP(hipMemcpyAsync(o->count, e->det_count, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  if (o->boxes) RS_HIP(hipMemcpyAsync(o->boxes, e->det_boxes, (size_t)n * D * 16, hipMemcpyDeviceToHost, s));
  if (o->scores) RS_HIP(hipMemcpyAsync(o->scores, e->det_scores, (size_t)n * D * 4, hipMemcpyDeviceToHost, s));
  if (o->classes) RS_HIP(hipMemcpyAsync(o->classes, e->det_classes, (size_t)n * D * 4, hipMemcpyDeviceToHost, s));
  if (o->masks && e->masks) RS_HIP(hipMemcpyAsync(o->masks, e->masks, (size_t)n * D * e->tile_h * ((e->tile_w + 7) / 8), hipMemcpyDeviceToHost, s));
  RS_HIP(hipEventRecord(e->ev_copied, s));
  e->copy_pending = true;
  return RS_OK;
}

int rs_engine_fetch_crops_async(rs_engine* e, int n, rs_dets* o, rs_mask_crops* c) {
  RS_CHECK(e && o && o->count && c && c->rects && c->offsets && c->data && n >= 1 && n <= e->max_batch, RS_ERR_ARG, "bad argument");
  RS_CHECK(e->masks && e->crop_data, RS_ERR_ARG, "mask crops requested but MASK_ON is false");
  if (!e->copy_stream) {
    RS_HIP(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    RS_HIP(hipEventCreateWithFlags(&e->ev_results, hipEventDisableTiming));
    RS_HIP(hipEventCreateWithFlags(&e->ev_copied, hipEventDisableTiming));
  }
  if (!e->ev_crop_hdr) {
    RS_HIP(hipEventCreateWithFlags(&e->ev_crop_hdr, hipEventDisableTiming));
    RS_HIP(hipHostMalloc((void**)&e->h_crop_total, 16, hipHostMallocDefault));
  }
  const int D = e->D;
  hipStream_t s = e->copy_stream;
  RS_HIP(hipEventRecord(e->ev_results, e->stream));
  RS_HIP(hipStreamWaitEvent(s, e->ev_results, 0));
  CropParams cp;
  memset(&cp, 0, sizeof cp);
  cp.det_boxes = e->det_boxes; cp.det_count = e->det_count; cp.masks = e->masks; cp.n = n; cp.D = D; cp.h = e->tile_h; cp.w = e->tile_w;
  cp.Wb = (e->tile_w + 7) / 8; cp.rects = e->crop_rects; cp.offsets = e->crop_offsets; cp.total = e->crop_total; cp.data = e->crop_data;
  { int rc = launch_mask_crops(cp, s); if (rc) return rc; }
  RS_HIP(hipMemcpyAsync(o->count, e->det_count, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  if (o->boxes) RS_HIP(hipMemcpyAsync(o->boxes, e->det_boxes, (size_t)n * D * 16, hipMemcpyDeviceToHost, s));
  if (o->scores) RS_HIP(hipMemcpyAsync(o->scores, e->det_scores, (size_t)n * D * 4, hipMemcpyDeviceToHost, s));
  if (o->classes) RS_HIP(hipMemcpyAsync(o->classes, e->det_classes, (size_t)n * D * 4, hipMemcpyDeviceToHost, s));
  RS_HIP(hipMemcpyAsync(c->rects, e->crop_rects, (size_t)n * D * 16, hipMemcpyDeviceToHost, s));
  RS_HIP(hipMemcpyAsync(c->offsets, e->crop_offsets, (size_t)n * D * 4, hipMemcpyDeviceToHost, s));
  RS_HIP(hipMemcpyAsync(e->h_crop_total, e->crop_total, 8, hipMemcpyDeviceToHost, s));
  RS_HIP(hipEventRecord(e->ev_copied, s));       // detections and canvases are free for the next forward: the crops live in their own buffer
  RS_HIP(hipEventRecord(e->ev_crop_hdr, s));
  e->copy_pending = true;
  return RS_OK;
}

int rs_engine_fetch_crops_wait(rs_engine* e, rs_mask_crops* c) {
  RS_CHECK(e && c && c->data && e->ev_crop_hdr, RS_ERR_ARG, "rs_engine_fetch_crops_wait without rs_engine_fetch_crops_async");
  RS_HIP(hipEventSynchronize(e->ev_crop_hdr));
  const unsigned long long used = *e->h_crop_total;
  RS_CHECK(used <= c->capacity, RS_ERR_ARG, "mask crops need %llu bytes, the caller's buffer holds %llu", used, (unsigned long long)c->capacity);
  c->used = used;
  if (used) RS_HIP(hipMemcpyAsync(c->data, e->crop_data, (size_t)used, hipMemcpyDeviceToHost, e->copy_stream));
  RS_HIP(hipStreamSynchronize(e->copy_stream));
  return RS_OK;
}

int rs_engine_fetch_wait(rs_engine* e) {
  RS_CHECK(e, RS_ERR_ARG, "null engine");
  if (e->copy_stream) RS_HIP(hipStreamSynchronize(e->copy_stream));
  return RS_OK;
}

int rs_engine_infer(rs_engine* e, const uint8_t* tiles_host, int n, rs_dets* out_host) {
  RS_CHECK(e && tiles_host && out_host, RS_ERR_ARG, "null argument");
  RS_CHECK(n >= 1 && n <= e->max_batch, RS_ERR_ARG, "batch %d outside [1, %d]", n, e->max_batch);
  RS_HIP(hipSetDevice(e->device));
  RS_HIP(hipMemcpyAsync(e->tiles_dev, tiles_host, (size_t)n * e->tile_h * e->tile_w * e->tile_c, hipMemcpyHostToDevice, e->stream));
  int rc = e->run(e->tiles_dev, n);
  if (rc) return rc;
  return rs_engine_fetch(e, n, out_host);
}

void* rs_engine_stream(rs_engine* e) { return e ? (void*)e->stream : nullptr; }

int rs_engine_set_profiling(rs_engine* e, int enabled) {
  RS_CHECK(e, RS_ERR_ARG, "null engine");
  RS_CHECK(enabled >= 0 && enabled <= 3, RS_ERR_ARG, "profiling mode %d", enabled);
  if (e->ev_used) { int rc = e->resolve_profile(); if (rc) return rc; }
  e->forward_index = 0;
  e->ev_used = 0;
  e->profiling = enabled;
  for (Stage& s : e->stages) { s.ms_total = 0; s.calls = 0; }
  if (enabled >= 2 && e->ev_pool.empty()) {
    const size_t want = e->stages.size() * 32;   // 32 forwards' worth of event pairs
    e->ev_pool.resize(want);
    e->ev_stage.resize(want);
    e->ev_batch.resize(want);
    for (auto& pr : e->ev_pool) {
      RS_HIP(hipEventCreate(&pr.first));
      RS_HIP(hipEventCreate(&pr.second));
    }
  }
  return RS_OK;
}
int rs_engine_stage_count(rs_engine* e) { return e ? (int)e->stages.size() : 0; }
int rs_engine_stage_info(rs_engine* e, int i, char* name_out, double* ms_total, int* calls, double* flops, double* bytes) {
  RS_CHECK(e && i >= 0 && i < (int)e->stages.size(), RS_ERR_ARG, "stage index");
  if (e->ev_used) { int rc = e->resolve_profile(); if (rc) return rc; }
  const Stage& s = e->stages[i];
  if (name_out) { strncpy(name_out, s.name.c_str(), 95); name_out[95] = 0; }
  if (ms_total) *ms_total = s.ms_total;
  if (calls) *calls = s.calls;
  if (flops) *flops = s.last_flops;
  if (bytes) *bytes = s.last_bytes;
  return RS_OK;
}

int rs_engine_stage_kernel(rs_engine* e, int i, char* name_out) {
  RS_CHECK(e && i >= 0 && i < (int)e->stages.size() && name_out, RS_ERR_ARG, "stage index");
  static const char* names[] = {"conv_igemm_kernel<2,2,4,4> 128x128", "conv_igemm_kernel<4,1,4,4> 256x64", "conv_igemm_kernel<4,1,1,4> 256x16 f32-out",
                                "conv_igemm_kernel<4,2,4,4> 256x128", "conv_igemm_kernel<2,4,4,8> 256x256", "conv_igemm_kernel<4,1,4,4,smallC> 256x64 stem",
                                "(retired)",
                                "conv_igemm_kernel<2,2,4,2> 64x128", "conv_igemm_kernel<4,1,4,2> 128x64", "conv_igemm_kernel<2,2,4,1> 32x128",
                                "conv_igemm_kernel<2,4,4,2> 64x256", "(retired)",
                                "conv_deep_kernel 256x256 (3 activation + 2 weight LDS stages)",
                                "bneck_tail_kernel 128 px (conv2 + conv3 + next conv1 chained through registers)",
                                "conv_igemm_kernel<2,4,4,4> 128x256",
                                "conv_deep_kernel 160x256", "conv_deep_kernel 192x256", "conv_deep_kernel 224x256",
                                "conv_deep_kernel 64x256", "conv_deep_kernel 96x256", "conv_deep_kernel 128x256",
                                "stem_pool_kernel (conv 7x7 s2 + ReLU + max-pool 3x3 s2, 8x8 pooled pixels per workgroup)",
                                "conv1x1_wreg_kernel 32 px x 256 ch (persistent, weights in registers, operand tiles by LDS-DMA, two workgroups per CU)",
                                "bneck_tail_split_kernel 128 px (conv2 + conv3 + next conv1 chained through registers)"};
  const int v = e->stages[i].variant;
  const char* s = v == -1 ? "conv_f32_mfma_kernel" : (v >= 0 && v <= 23 ? names[v] : "");
  if (e->split && v >= 0 && v <= 23) {
    snprintf(name_out, 96, "%.62s [split: hi+lo planes, 3 MFMA blocks]", s);
    return RS_OK;
  }
  strncpy(name_out, s, 95);
  name_out[95] = 0;
  return RS_OK;
}

int rs_engine_stage_variant(rs_engine* e, int i) {
  if (!e || i < 0 || i >= (int)e->stages.size()) return -2;
  return e->stages[i].variant;
}

int rs_op_conv_variant(int m, int cin, int k, int cout, int cin2, int deconv2x, int out_f32, int* stages_out) {
  ConvParams p;
  memset(&p, 0, sizeof p);
  p.M = m; p.Cin = cin; p.KH = p.KW = k; p.Cout = cout; p.mode = deconv2x ? 1 : 0; p.out_f32 = out_f32;
  p.stride = 1; p.in_Cs = cin; p.out_Cs = cout;
  p.Kpad = (k * k * (cin < 64 ? 8 : cin) + cin2 + 63) / 64 * 64;
  if (cin < 64 && k == 7) p.Kpad = 256;       // the stem's padded tap rows (weights.py STEM_KW_PAD)
  static const half_t dummy = (half_t)0;
  if (cin2 > 0) { p.in2 = &dummy; p.Cin2 = cin2; }
  const int v = conv_choose_variant(p, -1, 1);
  if (stages_out) *stages_out = p.stages;
  return cin < 64 ? 5 : v;
}

int rs_engine_tensor(rs_engine* e, const char* name, void** dev_ptr, int* dtype, int* ndim, int64_t dims[5], int* halo) {
  RS_CHECK(e && name, RS_ERR_ARG, "null argument");
  for (const TensorInfo& t : e->tensors) {
    if (t.name == name) {
      if (dev_ptr) *dev_ptr = t.p;
      if (dtype) *dtype = t.dtype;
      if (ndim) *ndim = t.ndim;
      if (dims) for (int i = 0; i < 5; ++i) dims[i] = t.dims[i];
      if (halo) *halo = t.halo;
      return RS_OK;
    }
  }
  rs_set_error("no tensor named %s", name);
  return RS_ERR_ARG;
}
int rs_engine_tensor_count(rs_engine* e) { return e ? (int)e->tensors.size() : 0; }
int rs_engine_tensor_name(rs_engine* e, int i, char* name_out) {
  RS_CHECK(e && i >= 0 && i < (int)e->tensors.size() && name_out, RS_ERR_ARG, "tensor index");
  strncpy(name_out, e->tensors[i].name.c_str(), 95);
  name_out[95] = 0;
  return RS_OK;
}

int rs_engine_net_shape(rs_engine* e, int* rh, int* rw, int* ph, int* pw) {
  RS_CHECK(e, RS_ERR_ARG, "null engine");
  if (rh) *rh = e->net_h;
  if (rw) *rw = e->net_w;
  if (ph) *ph = e->pad_h;
  if (pw) *pw = e->pad_w;
  return RS_OK;
}

// ------------------------------------------------------------------------- stand-alone operators
static long long* g_conv_probe = nullptr;   // -DRS_CLOCK_PROBE diagnostic builds: see rs_debug_set_conv_probe
struct SplitArgs { long long in_lo, w_lo, out_lo, res_lo, up_lo; const float* wscale; };
static int op_conv2d(const void* in, const void* w, const float* bias, void* out, const void* residual, const void* upsample_add,
                     int n, int hi, int wi, int cin, int in_halo, int kh, int kw, int stride, int pad, int cout, int kpad,
                     int out_halo, int relu, int out_f32, int deconv2x, int variant, int use_glds, void* stream,
                     const void* in2, int h2, int w2, int cin2, int in2_halo, int stride2, const SplitArgs* sa = nullptr) {
  RS_CHECK(in && w && bias && out, RS_ERR_ARG, "null argument");
  RS_CHECK(in_halo >= pad, RS_ERR_ARG, "input halo %d < pad %d", in_halo, pad);
  const int ho = (hi + 2 * pad - kh) / stride + 1, wo = (wi + 2 * pad - kw) / stride + 1;
  ConvParams p;
  memset(&p, 0, sizeof p);
  p.in = (const half_t*)in; p.w = (const half_t*)w; p.bias = bias; p.out = out;
  p.res = (const half_t*)residual; p.up = (const half_t*)upsample_add;
  p.M = n * ho * wo; p.Ho = ho; p.Wo = wo;
  p.in_Hp = hi + 2 * in_halo; p.in_Wp = wi + 2 * in_halo; p.in_Cs = cin; p.in_off = in_halo - pad;
  p.stride = stride; p.KH = kh; p.KW = kw; p.Cin = cin; p.Kpad = kpad; p.Cout = cout;
  const int oh = deconv2x ? 2 * ho : ho, ow = deconv2x ? 2 * wo : wo;
  p.out_Hp = oh + 2 * out_halo; p.out_Wp = ow + 2 * out_halo; p.out_Cs = cout; p.out_pad = out_halo;
  if (upsample_add) { p.up_Hp = ho / 2 + 2 * out_halo; p.up_Wp = wo / 2 + 2 * out_halo; p.up_Cs = cout; p.up_pad = out_halo; }
  p.relu = relu; p.mode = deconv2x ? 1 : 0; p.out_f32 = out_f32;
  p.probe = g_conv_probe;
  if (sa) {
    RS_CHECK(sa->wscale && !in2, RS_ERR_ARG, "split-operand conv: row scales missing (or a second K source, which the operator does not take)");
    p.split = 1; p.in_lo = sa->in_lo; p.w_lo = sa->w_lo; p.out_lo = sa->out_lo; p.res_lo = sa->res_lo; p.up_lo = sa->up_lo; p.wscale = sa->wscale;
  }
  if (in2) {
    RS_CHECK(stride2 >= 1 && (ho - 1) * stride2 < h2 && (wo - 1) * stride2 < w2, RS_ERR_ARG, "second source geometry");
    p.in2 = (const half_t*)in2; p.in2_Hp = h2 + 2 * in2_halo; p.in2_Wp = w2 + 2 * in2_halo; p.in2_Cs = cin2;
    p.in2_off = in2_halo; p.stride2 = stride2; p.Cin2 = cin2;
  }
  int* koff_dev = nullptr;
  if (cin < 64 && use_glds >= 0) {
    RS_CHECK(cin == 8, RS_ERR_UNSUPPORTED, "small-Cin path needs cin == 8");
    std::vector<int> koff(kpad / 8, 0);
    for (int t = 0; t < kh * kw && t < (int)koff.size(); ++t) koff[t] = ((t / kw) * p.in_Wp + (t % kw)) * cin;
    RS_HIP(hipMalloc((void**)&koff_dev, koff.size() * 4));
    RS_HIP(hipMemcpy(koff_dev, koff.data(), koff.size() * 4, hipMemcpyHostToDevice));
    p.koff = koff_dev;
  }
  int rc = launch_conv(p, (hipStream_t)stream, variant, use_glds);
  if (koff_dev) {
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(koff_dev);
  }
  return rc;
}

int rs_debug_set_conv_probe(void* buffer) { g_conv_probe = (long long*)buffer; return RS_OK; }

int rs_op_conv2d(const void* in, const void* w, const float* bias, void* out, const void* residual, const void* upsample_add,
                 int n, int hi, int wi, int cin, int in_halo, int kh, int kw, int stride, int pad, int cout, int kpad,
                 int out_halo, int relu, int out_f32, int deconv2x, int variant, int use_glds, void* stream) {
  return op_conv2d(in, w, bias, out, residual, upsample_add, n, hi, wi, cin, in_halo, kh, kw, stride, pad, cout, kpad, out_halo,
                   relu, out_f32, deconv2x, variant, use_glds, stream, nullptr, 0, 0, 0, 0, 1);
}

// The same convolution in the split-operand precision mode (rs_spec.precision == 2): every fp16 tensor is a hi plane with its lo plane `*_lo`
// ELEMENTS behind it (value = hi + lo), the weight rows are scaled by a power of two per row and `wscale` holds the inverses.
int rs_op_conv2d_split(const void* in, int64_t in_lo, const void* w, int64_t w_lo, const float* wscale, const float* bias, void* out, int64_t out_lo,
                       const void* residual, int64_t res_lo, const void* upsample_add, int64_t up_lo,
                       int n, int hi, int wi, int cin, int in_halo, int kh, int kw, int stride, int pad, int cout, int kpad,
                       int out_halo, int relu, int out_f32, int deconv2x, int variant, void* stream) {
  SplitArgs sa = {in_lo, w_lo, out_lo, res_lo, up_lo, wscale};
  return op_conv2d(in, w, bias, out, residual, upsample_add, n, hi, wi, cin, in_halo, kh, kw, stride, pad, cout, kpad, out_halo,
                   relu, out_f32, deconv2x, variant, 1, stream, nullptr, 0, 0, 0, 0, 1, &sa);
}

int rs_op_conv2d_dual(const void* in, const void* in2, const void* w, const float* bias, void* out,
                      int n, int hi, int wi, int cin, int in_halo, int kh, int kw, int stride, int pad,
                      int h2, int w2, int cin2, int in2_halo, int stride2,
                      int cout, int kpad, int out_halo, int relu, int variant, void* stream) {
  RS_CHECK(in2, RS_ERR_ARG, "null argument");
  return op_conv2d(in, w, bias, out, nullptr, nullptr, n, hi, wi, cin, in_halo, kh, kw, stride, pad, cout, kpad, out_halo,
                   relu, 0, 0, variant, 1, stream, in2, h2, w2, cin2, in2_halo, stride2);
}

int rs_op_bneck_tail(const void* t1, const void* w2, const float* b2, const void* w3p, const float* b3, const void* x, void* out,
                     const void* w1p, const float* b1, void* t1n, const void* x0, const void* wsc, int n, int h, int w, int width, void* stream) {
  RS_CHECK(t1 && w2 && b2 && w3p && b3 && out && n > 0 && h > 0 && w > 0, RS_ERR_ARG, "bad argument");
  RS_CHECK(width == 64 || width == 128, RS_ERR_UNSUPPORTED, "bneck_tail: bottleneck width %d (64 or 128)", width);
  BneckParams p;
  memset(&p, 0, sizeof p);
  p.t1 = (const half_t*)t1; p.w2 = (const half_t*)w2; p.b2 = b2; p.w3p = (const half_t*)w3p; p.b3 = b3; p.x = (const half_t*)x; p.out = (half_t*)out;
  p.w1p = (const half_t*)w1p; p.b1 = b1; p.t1n = (half_t*)t1n; p.x0 = (const half_t*)x0; p.wsc = (const half_t*)wsc;
  p.M = n * h * w; p.H = h; p.W = w; p.Hp = h + 2; p.Wp = w + 2; p.CB = width / 64;
  return launch_bneck_tail(p, (hipStream_t)stream);
}

int rs_op_bneck_tail_split(const void* t1, int64_t t1_lo, const void* w2, const float* s2, const float* b2, const void* w3p, const float* s3, const float* b3,
                           const void* x, int64_t x_lo, void* out, int64_t out_lo, const void* w1p, const float* s1, const float* b1, void* t1n, int64_t t1n_lo,
                           const void* x0, int64_t x0_lo, int n, int h, int w, int width, void* stream) {
  RS_CHECK(t1 && w2 && s2 && b2 && w3p && s3 && b3 && (x || x0) && out && n > 0 && h > 0 && w > 0, RS_ERR_ARG, "bad argument");
  RS_CHECK(width == 64 || width == 128, RS_ERR_UNSUPPORTED, "bneck_tail_split: bottleneck width %d (64 or 128)", width);
  BneckSplitParams p;
  memset(&p, 0, sizeof p);
  p.t1 = (const half_t*)t1; p.t1_lo = t1_lo;
  p.w2 = (const half_t*)w2; p.w2_lo = (long long)width * 9 * width; p.s2 = s2; p.b2 = b2;
  p.w3p = (const half_t*)w3p; p.w3_lo = 4ll * width * (width + (x0 ? 64 : 0)); p.s3 = s3; p.b3 = b3;
  p.x = (const half_t*)x; p.x_lo = x_lo; p.x0 = (const half_t*)x0; p.x0_lo = x0_lo; p.out = (half_t*)out; p.out_lo = out_lo;
  p.w1p = (const half_t*)w1p; p.w1_lo = 4ll * width * width; p.s1 = s1; p.b1 = b1; p.t1n = (half_t*)t1n; p.t1n_lo = t1n_lo;
  p.M = n * h * w; p.H = h; p.W = w; p.Hp = h + 2; p.Wp = w + 2; p.CB = width / 64;
  return launch_bneck_tail_split(p, (hipStream_t)stream);
}

int rs_op_mask_overlap(const uint8_t* det_masks, int n_det, const uint8_t* label_masks, int n_labels, int h, int w, int32_t* inter,
                       int32_t* label_area, void* stream) {
  return launch_mask_overlap(det_masks, n_det, label_masks, n_labels, h, w, inter, label_area, (hipStream_t)stream);
}

// the same on the detection masks of tile `tile` of the engine's last forward (all D slots; slots >= count hold stale canvases,
// the caller reads the first count[tile] columns)
int rs_engine_label_overlap(rs_engine* e, int tile, const uint8_t* label_masks_dev, int n_labels, int32_t* inter_dev, int32_t* label_area_dev) {
  RS_CHECK(e && e->masks && tile >= 0 && tile < e->max_batch && label_masks_dev && inter_dev && label_area_dev, RS_ERR_ARG, "bad argument");
  const size_t per = (size_t)e->tile_h * ((e->tile_w + 7) / 8);
  return launch_mask_overlap(e->masks + (size_t)tile * e->D * per, e->D, label_masks_dev, n_labels, e->tile_h, e->tile_w, inter_dev, label_area_dev, e->stream);
}

int rs_op_conv2d_dgrad(const void* dy, const void* w_t, void* dx, const void* res, const float* res32, const void* mask,
                       const void* down, int n, int hi, int wi, int cin, int ho, int wo, int cout, int kh, int kw, int stride,
                       int pad, int kpad, int halo, int variant, void* stream) {
  RS_CHECK(dy && w_t && dx, RS_ERR_ARG, "null argument");
  RS_CHECK(stride == 1 || (kh == 1 && kw == 1), RS_ERR_UNSUPPORTED, "dgrad: stride %d needs a 1x1 kernel (STRIDE_IN_1X1)", stride);
  RS_CHECK(halo >= kh - 1 - pad && halo >= 0 && kh == kw, RS_ERR_ARG, "dgrad: halo %d too small", halo);
  // the input gradient of conv(x, W, stride 1, pad) is conv(dy, W^T flipped, stride 1, pad' = k-1-pad); of a stride-s 1x1
  // convolution it is the 1x1 convolution of dy stored at every s-th pixel of dx
  ConvParams p;
  memset(&p, 0, sizeof p);
  const int pad_t = kh - 1 - pad;
  const int oh = stride == 1 ? ho + 2 * pad_t - kh + 1 : ho, ow = stride == 1 ? wo + 2 * pad_t - kw + 1 : wo;
  RS_CHECK(stride == 1 ? (oh == hi && ow == wi) : ((ho - 1) * stride < hi && (wo - 1) * stride < wi), RS_ERR_ARG, "dgrad: geometry");
  void* zero_bias = nullptr;
  RS_HIP(hipMalloc(&zero_bias, (size_t)cin * 4 + 256));
  RS_HIP(hipMemsetAsync(zero_bias, 0, (size_t)cin * 4 + 256, (hipStream_t)stream));
  p.in = (const half_t*)dy; p.w = (const half_t*)w_t; p.bias = (const float*)zero_bias; p.out = dx;
  p.res = (const half_t*)res; p.res32 = res32; p.mask = (const half_t*)mask; p.down = (const half_t*)down;
  p.M = n * oh * ow; p.Ho = oh; p.Wo = ow;
  p.in_Hp = ho + 2 * halo; p.in_Wp = wo + 2 * halo; p.in_Cs = cout; p.in_off = halo - pad_t;
  p.stride = 1; p.KH = kh; p.KW = kw; p.Cin = cout; p.Kpad = kpad; p.Cout = cin;
  p.out_Hp = hi + 2 * halo; p.out_Wp = wi + 2 * halo; p.out_Cs = cin; p.out_pad = halo;
  p.out_stride = stride;
  if (down) { p.down_Hp = 2 * hi + 2 * halo; p.down_Wp = 2 * wi + 2 * halo; p.down_Cs = cin; p.down_pad = halo; }
  int rc = launch_conv(p, (hipStream_t)stream, variant, 1);
  (void)hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(zero_bias);
  return rc;
}

static int op_conv2d_wgrad(const void* dy, const void* x, float* grad, const float* scale, int n, int hi, int wi, int cin, int in_halo,
                           int kh, int kw, int stride, int pad, int cout, int kpad, int dy_halo, int splits, void* stream, int f32);
int rs_op_conv2d_wgrad(const void* dy, const void* x, float* grad, const float* scale, int n, int hi, int wi, int cin, int in_halo,
                       int kh, int kw, int stride, int pad, int cout, int kpad, int dy_halo, int splits, void* stream) {
  return op_conv2d_wgrad(dy, x, grad, scale, n, hi, wi, cin, in_halo, kh, kw, stride, pad, cout, kpad, dy_halo, splits, stream, 0);
}
// the same weight gradient from fp32 operands (reference-precision trainer: conv_wgrad_f32_kernel)
int rs_op_conv2d_wgrad_f32(const void* dy, const void* x, float* grad, const float* scale, int n, int hi, int wi, int cin, int in_halo,
                           int kh, int kw, int stride, int pad, int cout, int kpad, int dy_halo, int splits, void* stream) {
  return op_conv2d_wgrad(dy, x, grad, scale, n, hi, wi, cin, in_halo, kh, kw, stride, pad, cout, kpad, dy_halo, splits, stream, 1);
}
static int op_conv2d_wgrad(const void* dy, const void* x, float* grad, const float* scale, int n, int hi, int wi, int cin, int in_halo,
                           int kh, int kw, int stride, int pad, int cout, int kpad, int dy_halo, int splits, void* stream, int f32) {
  RS_CHECK(dy && x && grad, RS_ERR_ARG, "null argument");
  RS_CHECK(in_halo >= pad, RS_ERR_ARG, "input halo %d < pad %d", in_halo, pad);
  const int ho = (hi + 2 * pad - kh) / stride + 1, wo = (wi + 2 * pad - kw) / stride + 1;
  WgradParams p;
  memset(&p, 0, sizeof p);
  p.dy = (const half_t*)dy; p.x = (const half_t*)x; p.grad = grad; p.scale = scale;
  p.M = n * ho * wo; p.Ho = ho; p.Wo = wo;
  p.dy_Hp = ho + 2 * dy_halo; p.dy_Wp = wo + 2 * dy_halo; p.dy_Cs = cout; p.dy_pad = dy_halo;
  p.in_Hp = hi + 2 * in_halo; p.in_Wp = wi + 2 * in_halo; p.in_Cs = cin; p.in_off = in_halo - pad;
  p.stride = stride; p.KH = kh; p.KW = kw; p.Cin = cin; p.Cout = cout; p.Kpad = kpad;
  p.f32 = f32;
  p.splits = splits > 0 ? splits : wgrad_splits(p);
  hipStream_t s = (hipStream_t)stream;
  void *partial = nullptr, *zeros = nullptr;
  RS_HIP(hipMalloc(&partial, (size_t)p.splits * cout * kpad * 4));
  RS_HIP(hipMalloc(&zeros, (size_t)cout * 2 + 256));
  RS_HIP(hipMemsetAsync(zeros, 0, (size_t)cout * 2 + 256, s));
  RS_HIP(hipMemsetAsync(partial, 0, (size_t)p.splits * cout * kpad * 4, s));   // K padding columns stay zero
  p.partial = (float*)partial; p.zeros = (const half_t*)zeros;
  int rc = launch_conv_wgrad(p, s);
  (void)hipStreamSynchronize(s);
  (void)hipFree(partial);
  (void)hipFree(zeros);
  return rc;
}

int rs_op_nms(const float* boxes, const int32_t* counts, const uint8_t* valid, uint8_t* keep, int segments, int cap,
              float thresh, void* stream) {
  RS_CHECK(boxes && counts && keep && segments > 0, RS_ERR_ARG, "bad argument");
  RS_CHECK(cap >= 1 && cap <= 2048, RS_ERR_ARG, "cap %d outside [1,2048]", cap);
  NmsParams p = {};
  p.boxes = boxes; p.count = counts; p.valid = valid; p.keep = keep; p.cap = cap; p.thresh = thresh;
  void* scratch = nullptr;
  if (cap > 1024) {            // training capacity: the suppression mask lives in global memory
    RS_HIP(hipMalloc(&scratch, (size_t)segments * 2048 * 32 * 8));
    p.scratch = (unsigned long long*)scratch;
  }
  int rc = launch_nms(p, segments, (hipStream_t)stream);
  if (scratch) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(scratch); }
  return rc;
}

int rs_op_roi_align(const void* const feats[4], const int32_t heights[4], const int32_t widths[4], const float scales[4],
                    int nlevels, const float* rois, int n_rois, int rois_per_image, int P, int out_halo, void* out,
                    int32_t* levels_out, void* stream) {
  RS_CHECK(feats && rois && out && nlevels >= 1 && nlevels <= 4 && n_rois > 0 && rois_per_image > 0, RS_ERR_ARG, "bad argument");
  RoiAlignParams p;
  memset(&p, 0, sizeof p);
  for (int l = 0; l < nlevels; ++l) { p.feat[l] = (const half_t*)feats[l]; p.H[l] = heights[l]; p.W[l] = widths[l]; p.scale[l] = scales[l]; }
  p.nlevels = nlevels; p.C = 256; p.rois = rois; p.S = n_rois; p.slots_per_image = rois_per_image;
  p.out = (half_t*)out; p.P = P; p.out_pad = out_halo; p.out_level = levels_out;
  return launch_roi_align(p, (hipStream_t)stream);
}

int rs_op_roi_align_bwd(float* const dfeats[4], const int32_t heights[4], const int32_t widths[4], const float scales[4],
                        int nlevels, const float* rois, int n_rois, int rois_per_image, int P, int out_halo, const void* dout,
                        void* stream) {
  RS_CHECK(dfeats && rois && dout && nlevels >= 1 && nlevels <= 4 && n_rois > 0 && rois_per_image > 0, RS_ERR_ARG, "bad argument");
  RoiAlignParams p;
  memset(&p, 0, sizeof p);
  for (int l = 0; l < nlevels; ++l) { p.dfeat[l] = dfeats[l]; p.H[l] = heights[l]; p.W[l] = widths[l]; p.scale[l] = scales[l]; }
  p.nlevels = nlevels; p.C = 256; p.rois = rois; p.S = n_rois; p.slots_per_image = rois_per_image;
  p.out = (half_t*)dout; p.P = P; p.out_pad = out_halo;
  rs_debug_reload();                                         // RS_ROI_BWD_ATOMIC: the operator tests switch between the two forms
  // workspace of the owner-computes form (the trainer owns its own): per-entry tables + the overflow counter, for the time of this call
  p.n_images = (n_rois + rois_per_image - 1) / rois_per_image;
  void* ws = nullptr;
  RS_HIP(hipMalloc(&ws, (size_t)n_rois * RS_ROI_BWD_TABLE_BYTES + 64));
  p.bwd_overflow = (int*)ws;
  p.bwd_tables = (char*)ws + 64;
  const int rc = launch_roi_align_bwd(p, (hipStream_t)stream);
  hipError_t he = hipStreamSynchronize((hipStream_t)stream);
  hipFree(ws);
  RS_HIP(he);
  return rc;
}

int rs_op_rpn_loss(const float* head, void* dhead, const int32_t* labels, const float* anchors, const float* matched_gt,
                   float* loss_out, int n, int hw, int num_anchors, int cs, int level_off, int total_anchors, float normalizer,
                   float loss_scale, void* stream) {
  RpnLossParams p;
  memset(&p, 0, sizeof p);
  p.head = head; p.dhead = (half_t*)dhead; p.labels = labels; p.anchors = anchors; p.matched_gt = matched_gt; p.loss_out = loss_out;
  p.A = num_anchors; p.cs = cs; p.HW = hw; p.n_anchors = hw * num_anchors; p.level_off = level_off; p.total_anchors = total_anchors;
  p.normalizer = normalizer; p.loss_scale = loss_scale;
  return launch_rpn_loss(p, n, (hipStream_t)stream);
}

int rs_op_box_loss(const float* pred, void* dpred, const int32_t* gt_classes, const float* proposals, const float* gt_boxes,
                   float* loss_out, int n_rois, int num_classes, int cs, float n_valid, const float reg_weights[4], float loss_scale,
                   void* stream) {
  RS_CHECK(reg_weights, RS_ERR_ARG, "null argument");
  BoxLossParams p;
  memset(&p, 0, sizeof p);
  p.pred = pred; p.dpred = (half_t*)dpred; p.gt_classes = gt_classes; p.proposals = proposals; p.gt_boxes = gt_boxes; p.loss_out = loss_out;
  p.n_rois = n_rois; p.K = num_classes; p.cs = cs; p.n_valid = n_valid;
  p.wx = reg_weights[0]; p.wy = reg_weights[1]; p.ww = reg_weights[2]; p.wh = reg_weights[3]; p.loss_scale = loss_scale;
  return launch_box_loss(p, (hipStream_t)stream);
}

int rs_op_mask_loss(const float* logits, void* dlogits, const uint8_t* targets, const int32_t* gt_classes, float* loss_out, int n_masks,
                    int side, int cs, float loss_scale, void* stream) {
  MaskLossParams p;
  memset(&p, 0, sizeof p);
  p.logits = logits; p.dlogits = (half_t*)dlogits; p.targets = targets; p.gt_classes = gt_classes; p.loss_out = loss_out;
  p.n_masks = n_masks; p.S = side; p.cs = cs; p.loss_scale = loss_scale;
  return launch_mask_loss(p, (hipStream_t)stream);
}

int rs_op_match(const float* boxes, int per_image_boxes, const int32_t* box_count, const float* gt, const int32_t* gt_count,
                int32_t* matched, int32_t* labels, float* best_iou, int n_images, int n_boxes, int gt_cap, float t_lo, float t_hi,
                int lbl_lo, int lbl_mid, int lbl_hi, int allow_low_quality, void* stream) {
  MatchParams p;
  memset(&p, 0, sizeof p);
  p.boxes = boxes; p.per_image_boxes = per_image_boxes; p.box_count = box_count; p.gt = gt; p.gt_count = gt_count;
  p.matched = matched; p.labels = labels; p.best_iou = best_iou; p.n_boxes = n_boxes; p.gt_cap = gt_cap;
  p.t_lo = t_lo; p.t_hi = t_hi; p.lbl_lo = lbl_lo; p.lbl_mid = lbl_mid; p.lbl_hi = lbl_hi;
  void* scratch = nullptr;
  if (allow_low_quality) {
    RS_HIP(hipMalloc(&scratch, (size_t)n_images * gt_cap * 4));
    p.gt_best = (unsigned int*)scratch;
  }
  int rc = launch_match(p, n_images, (hipStream_t)stream);
  if (scratch) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(scratch); }
  return rc;
}

int rs_op_subsample(int32_t* labels, int32_t* sampled, int32_t* sampled_count, int n_images, int n, int num_samples,
                    float positive_fraction, int bg_label, int rpn_mode, uint32_t seed, void* stream) {
  SubsampleParams p;
  memset(&p, 0, sizeof p);
  p.labels = labels; p.sampled = sampled; p.sampled_count = sampled_count; p.n = n; p.num_samples = num_samples;
  p.positive_fraction = positive_fraction; p.bg_label = bg_label; p.rpn_mode = rpn_mode; p.seed = seed;
  return launch_subsample(p, n_images, (hipStream_t)stream);
}

int rs_op_sgd_momentum(float* w, float* momentum_buf, const float* grad, int64_t n, float lr, float momentum, float weight_decay,
                       float inv_loss_scale, int first_step, void* stream) {
  return launch_sgd_momentum(w, momentum_buf, grad, n, lr, momentum, weight_decay, inv_loss_scale, first_step, (hipStream_t)stream);
}

int rs_op_fold_weights(const float* w32, const float* scale, void* w_fwd, void* w_bwd, int cout, int cin, int kh, int kw, int kpad,
                       int kpad_t, void* stream) {
  return launch_fold_weights(w32, scale, (half_t*)w_fwd, (half_t*)w_bwd, cout, cin, kh, kw, kpad, cout, kpad_t, (hipStream_t)stream);
}

}  // extern "C"

#include "train_engine.inc"
