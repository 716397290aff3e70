"""ctypes binding of ``librs_engine.so`` (C ABI: ``include/rs_engine.h``).

This is the host-side mirror of the call the reference's detector makes per tile,
``DefaultPredictor(cfg)(im)`` ([EXT d2: engine/defaults.py], invoked by the object-detector's
``make_detections.py``, R:README.md:78).  There is no CPU fallback: if the shared library or a HIP
device is missing the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import weakref
import math
import os
import time
from typing import Any, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from .spec import EngineSpec
from .weights import pack_weights

_LIB: Optional[C.CDLL] = None
# RS_ENGINE_LIB: another build of the library (diagnostic builds of tools/ubench: ablations, clock probes); default = the in-tree one
LIB_PATH = os.environ.get("RS_ENGINE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "librs_engine.so")
MAX_LEVELS, MAX_ANCHORS, MASK_SIDE = 5, 8, 28
DT_NP = {1: np.float16, 2: np.float32, 3: np.int32, 4: np.uint8, 5: np.float16}       # 5: two fp16 planes (split-operand mode), value = hi + lo
DT_SPLIT16 = 5


class RsError(RuntimeError):
    pass


class RsSpec(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("in_channels", C.c_int32), ("flip_channels", C.c_int32),
        ("pixel_mean", C.c_float * 4), ("pixel_std", C.c_float * 4),
        ("min_size_test", C.c_int32), ("max_size_test", C.c_int32), ("size_divisibility", C.c_int32),
        ("res_blocks", C.c_int32 * 4), ("stem_out_channels", C.c_int32), ("res2_out_channels", C.c_int32),
        ("stride_in_1x1", C.c_int32), ("fpn_out_channels", C.c_int32),
        ("num_levels", C.c_int32), ("num_anchors", C.c_int32),
        ("cell_anchors", ((C.c_float * 4) * MAX_ANCHORS) * MAX_LEVELS), ("anchor_offset", C.c_float),
        ("rpn_bbox_reg_weights", C.c_float * 4), ("rpn_pre_nms_topk", C.c_int32), ("rpn_post_nms_topk", C.c_int32),
        ("rpn_nms_thresh", C.c_float), ("rpn_min_size", C.c_float),
        ("num_classes", C.c_int32), ("box_reg_weights", C.c_float * 4),
        ("score_thresh_test", C.c_float), ("nms_thresh_test", C.c_float), ("detections_per_image", C.c_int32),
        ("box_fc_dim", C.c_int32), ("box_pooler_resolution", C.c_int32),
        ("mask_on", C.c_int32), ("mask_pooler_resolution", C.c_int32), ("mask_num_conv", C.c_int32),
        ("mask_conv_dim", C.c_int32), ("mask_threshold", C.c_float), ("scale_clamp", C.c_float),
        ("precision", C.c_int32),
    ]


class RsDets(C.Structure):
    _fields_ = [("count", C.POINTER(C.c_int32)), ("boxes", C.POINTER(C.c_float)), ("scores", C.POINTER(C.c_float)),
                ("classes", C.POINTER(C.c_int32)), ("masks", C.POINTER(C.c_uint8)), ("mask_probs", C.POINTER(C.c_float))]


class RsMaskCrops(C.Structure):
    _fields_ = [("rects", C.POINTER(C.c_int32)), ("offsets", C.POINTER(C.c_uint32)), ("data", C.POINTER(C.c_uint8)),
                ("capacity", C.c_uint64), ("used", C.c_uint64)]


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load librs_engine.so (built in-tree by ``__graft_entry__.build()`` / csrc/Makefile)."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    try:
        # PyTorch-ROCm wheels bundle their own libamdhip64; librs_engine.so links the system one (/opt/rocm).  If ours is loaded
        # first and torch later initialises the GPU through its copy, the second HIP runtime of the process reports "no
        # ROCm-capable device".  Importing torch first (when it is installed -- tests, bench.py and the data-parallel trainer use it)
        # lets both bind to ONE runtime; the engine itself never calls into torch.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise RsError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc, gfx950). There is no CPU fallback for the detection path.")
    lib = C.CDLL(p)
    vp, i32, f32p, i32p, u8p = C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    lib.rs_last_error.restype = C.c_char_p
    lib.rs_abi_version.restype = i32
    lib.rs_engine_create.argtypes = [C.POINTER(RsSpec), vp, C.c_size_t, i32, i32, i32, i32, i32, vp, C.POINTER(vp)]
    lib.rs_engine_destroy.argtypes = [vp]
    lib.rs_engine_destroy.restype = None
    lib.rs_engine_infer.argtypes = [vp, vp, i32, C.POINTER(RsDets)]
    lib.rs_engine_infer_device.argtypes = [vp, vp, i32]
    lib.rs_engine_infer_phase.argtypes = [vp, vp, i32, i32]
    lib.rs_engine_sync.argtypes = [vp]
    lib.rs_host_alloc.argtypes = [C.c_size_t]
    lib.rs_host_alloc.restype = vp
    lib.rs_host_free.argtypes = [vp]
    lib.rs_host_free.restype = None
    lib.rs_engine_upload_async.argtypes = [vp, vp, i32]
    lib.rs_engine_fetch_async.argtypes = [vp, i32, C.POINTER(RsDets)]
    lib.rs_engine_fetch_wait.argtypes = [vp]
    lib.rs_engine_fetch_crops_async.argtypes = [vp, i32, C.POINTER(RsDets), C.POINTER(RsMaskCrops)]
    lib.rs_engine_fetch_crops_wait.argtypes = [vp, C.POINTER(RsMaskCrops)]
    lib.rs_engine_fetch.argtypes = [vp, i32, C.POINTER(RsDets)]
    lib.rs_engine_stream.argtypes = [vp]
    lib.rs_engine_stream.restype = vp
    lib.rs_engine_set_profiling.argtypes = [vp, i32]
    lib.rs_debug_run_stages_matching.argtypes = [vp, C.c_char_p, i32]
    lib.rs_engine_stage_count.argtypes = [vp]
    lib.rs_engine_stage_info.argtypes = [vp, i32, C.c_char_p, C.POINTER(C.c_double), i32p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.rs_engine_stage_kernel.argtypes = [vp, i32, C.c_char_p]
    lib.rs_engine_stage_variant.argtypes = [vp, i32]
    lib.rs_op_conv_variant.argtypes = [i32] * 7 + [i32p]
    lib.rs_engine_tensor.argtypes = [vp, C.c_char_p, C.POINTER(vp), i32p, i32p, C.POINTER(C.c_int64), i32p]
    lib.rs_engine_tensor_count.argtypes = [vp]
    lib.rs_engine_tensor_name.argtypes = [vp, i32, C.c_char_p]
    lib.rs_engine_net_shape.argtypes = [vp, i32p, i32p, i32p, i32p]
    lib.rs_op_conv2d.argtypes = [vp, vp, vp, vp, vp, vp] + [i32] * 17 + [vp]
    lib.rs_op_bneck_tail.argtypes = [vp] * 12 + [i32, i32, i32, i32, vp]
    lib.rs_op_conv2d_dual.argtypes = [vp, vp, vp, vp, vp] + [i32] * 19 + [vp]
    i64 = C.c_int64
    lib.rs_op_conv2d_split.argtypes = [vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, vp, i64] + [i32] * 16 + [vp]
    lib.rs_op_bneck_tail_split.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp]
    lib.rs_op_conv2d_dgrad.argtypes = [vp] * 7 + [i32] * 14 + [vp]
    lib.rs_op_conv2d_wgrad.argtypes = [vp, vp, vp, vp] + [i32] * 13 + [vp]
    lib.rs_op_nms.argtypes = [vp, vp, vp, vp, i32, i32, C.c_float, vp]
    lib.rs_op_roi_align.argtypes = [C.POINTER(vp), i32p, i32p, f32p, i32, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.rs_op_roi_align_bwd.argtypes = [C.POINTER(vp), i32p, i32p, f32p, i32, vp, i32, i32, i32, i32, vp, vp]
    f32 = C.c_float
    lib.rs_op_match.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, i32, i32, i32, i32, vp]
    lib.rs_op_subsample.argtypes = [vp, vp, vp, i32, i32, i32, f32, i32, i32, C.c_uint32, vp]
    lib.rs_op_rpn_loss.argtypes = [vp] * 6 + [i32] * 6 + [f32, f32, vp]
    lib.rs_op_box_loss.argtypes = [vp] * 6 + [i32, i32, i32, f32, f32p, f32, vp]
    lib.rs_op_mask_loss.argtypes = [vp] * 5 + [i32, i32, i32, f32, vp]
    lib.rs_op_sgd_momentum.argtypes = [vp, vp, vp, C.c_int64, f32, f32, f32, f32, i32, vp]
    lib.rs_op_fold_weights.argtypes = [vp] * 4 + [i32] * 6 + [vp]
    lib.rs_resize_shape.argtypes = [i32, i32, i32, i32, i32p, i32p]
    lib.rs_resize_shape.restype = None
    lib.rs_resize_coeffs.argtypes = [i32, i32, i32p, i32p]
    i64p, f64p = C.POINTER(C.c_int64), C.POINTER(C.c_double)
    lib.rs_vectorize_masks.argtypes = [vp, i32, i32, i32, C.c_double, i32]
    lib.rs_vectorize_masks.restype = vp
    lib.rs_vec_counts.argtypes = [vp, i64p, i64p, i64p, i64p]
    lib.rs_vec_counts.restype = None
    lib.rs_vec_copy.argtypes = [vp, i32p, i32p, i32p, f64p]
    lib.rs_vec_free.argtypes = [vp]
    lib.rs_vec_free.restype = None
    lib.rs_rasterize_polygons_within_box.argtypes = [f64p, i32p, i32, f64p, i32, u8p]
    lib.rs_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
    lib.rs_op_fdiv.argtypes = [vp, vp, vp, C.c_int64, vp]
    lib.rs_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
    if path is None:
        _LIB = lib
    return lib


def _check(lib: C.CDLL, rc: int, what: str) -> None:
    if rc != 0:
        raise RsError(f"{what} failed ({rc}): {lib.rs_last_error().decode(errors='replace')}")


def cell_anchor_table(spec: EngineSpec) -> np.ndarray:
    """``DefaultAnchorGenerator.generate_cell_anchors`` [EXT d2: modeling/anchor_generator.py]
    (R:45-56): float64 arithmetic, stored as fp32.  Shape (levels, anchors, 4)."""
    out = np.zeros((len(spec.anchor_sizes), spec.num_anchors, 4), np.float32)
    for l, sizes in enumerate(spec.anchor_sizes):
        a = 0
        for size in sizes:
            area = float(size) ** 2.0
            for ar in spec.anchor_aspect_ratios:
                w = math.sqrt(area / ar)
                h = ar * w
                out[l, a] = (-w / 2.0, -h / 2.0, w / 2.0, h / 2.0)
                a += 1
    return out


def make_rs_spec(spec: EngineSpec) -> RsSpec:
    spec.check_supported()
    s = RsSpec()
    s.struct_size = C.sizeof(RsSpec)
    s.in_channels = spec.in_channels
    s.flip_channels = 1 if spec.input_format == "RGB" else 0
    for i in range(spec.in_channels):
        s.pixel_mean[i] = spec.pixel_mean[i]
        s.pixel_std[i] = spec.pixel_std[i]
    s.min_size_test, s.max_size_test, s.size_divisibility = spec.min_size_test, spec.max_size_test, spec.size_divisibility
    for i, b in enumerate(spec.res_blocks):
        s.res_blocks[i] = b
    s.stem_out_channels, s.res2_out_channels = spec.stem_out_channels, spec.res2_out_channels
    s.stride_in_1x1 = int(spec.stride_in_1x1)
    s.fpn_out_channels = spec.fpn_out_channels
    s.num_levels, s.num_anchors = len(spec.rpn_in_features), spec.num_anchors
    if s.num_levels > MAX_LEVELS or s.num_anchors > MAX_ANCHORS:
        raise RsError("too many RPN levels / anchors")
    ca = cell_anchor_table(spec)
    for l in range(ca.shape[0]):
        for a in range(ca.shape[1]):
            for d in range(4):
                s.cell_anchors[l][a][d] = float(ca[l, a, d])
    s.anchor_offset = spec.anchor_offset
    for i in range(4):
        s.rpn_bbox_reg_weights[i] = spec.rpn_bbox_reg_weights[i]
        s.box_reg_weights[i] = spec.box_reg_weights[i]
    s.rpn_pre_nms_topk, s.rpn_post_nms_topk = spec.rpn_pre_nms_topk_test, spec.rpn_post_nms_topk_test
    s.rpn_nms_thresh, s.rpn_min_size = spec.rpn_nms_thresh, spec.rpn_min_size
    s.num_classes = spec.num_classes
    s.score_thresh_test, s.nms_thresh_test = spec.score_thresh_test, spec.nms_thresh_test
    s.detections_per_image = spec.detections_per_image
    s.box_fc_dim, s.box_pooler_resolution = spec.box_fc_dim, spec.box_pooler_resolution
    s.mask_on, s.mask_pooler_resolution = int(spec.mask_on), spec.mask_pooler_resolution
    s.mask_num_conv, s.mask_conv_dim = spec.mask_num_conv, spec.mask_conv_dim
    s.mask_threshold, s.scale_clamp = spec.mask_threshold, spec.scale_clamp
    s.precision = {"fp16": 0, "fp32": 1, "split": 2}[spec.precision]
    return s


class Instances:
    """Numpy-backed stand-in for detectron2 ``Instances`` ([EXT d2: structures/instances.py]) with the
    four fields the object-detector reads: ``pred_boxes`` (n,4 XYXY abs, tile pixels), ``scores``,
    ``pred_classes``, ``pred_masks`` (n,H,W bool).  Sorted by score, descending."""

    def __init__(self, image_size: Tuple[int, int], pred_boxes: np.ndarray, scores: np.ndarray, pred_classes: np.ndarray,
                 packed_masks: Optional[np.ndarray], mask_probs: Optional[np.ndarray],
                 crops: Optional[Tuple[np.ndarray, np.ndarray, np.ndarray]] = None):
        self.image_size = image_size
        self.pred_boxes = pred_boxes
        self.scores = scores
        self.pred_classes = pred_classes
        self._packed_full = packed_masks
        self.mask_probs = mask_probs
        # masks as crops of their boxes (rs_mask_crops): (rects (n,4) int32 [first byte column, first row, bytes per row, rows],
        # offsets (n,) uint32 into data, data uint8); the full canvases are rebuilt on demand
        self._crops = crops

    @property
    def _packed(self) -> Optional[np.ndarray]:
        """Bit-packed full-canvas masks (n, h, ceil(w/8)); rebuilt from the crops when the masks arrived cropped."""
        if self._packed_full is None and self._crops is not None:
            h, w = self.image_size
            rects, offs, data = self._crops
            full = np.zeros((len(self), h, (w + 7) // 8), np.uint8)
            for i in range(len(self)):
                x0b, y0, wb, rows = (int(v) for v in rects[i])
                if wb > 0 and rows > 0:
                    o = int(offs[i])
                    full[i, y0:y0 + rows, x0b:x0b + wb] = data[o:o + wb * rows].reshape(rows, wb)
            self._packed_full = full
        return self._packed_full

    def __len__(self) -> int:
        return int(self.scores.shape[0])

    def has(self, name: str) -> bool:
        return name in ("pred_boxes", "scores", "pred_classes") or (name == "pred_masks" and (self._packed_full is not None or self._crops is not None))

    @property
    def pred_masks(self) -> np.ndarray:
        if self._packed is None:
            raise AttributeError("pred_masks not computed (MASK_ON false)")
        h, w = self.image_size
        if len(self) == 0:
            return np.zeros((0, h, w), bool)
        bits = np.unpackbits(self._packed, axis=-1, bitorder="little")[:, :, :w]
        return bits.astype(bool)

    def to(self, *_a: Any, **_k: Any) -> "Instances":   # ``instances.to("cpu")`` in caller code
        return self

    def get_fields(self) -> Dict[str, Any]:
        d = {"pred_boxes": self.pred_boxes, "scores": self.scores, "pred_classes": self.pred_classes}
        if self.has("pred_masks"):
            d["pred_masks"] = self.pred_masks
        return d


_REGISTERED_HOST: List[Tuple[int, int, Any]] = []        # ([lo, hi) address range pinned with rs_host_register, weak reference to the array)


def _registered_view(a: np.ndarray) -> bool:
    """Is ``a`` (a view of) an array pinned by ``register_host_buffer``?  Address AND object identity: an unrelated array that
    happens to sit at the address of a slab that has gone away must not pass for pinned."""
    if not a.flags.c_contiguous:
        return False
    addr = a.ctypes.data
    for lo, hi, buf in _REGISTERED_HOST:
        if lo <= addr and addr + a.nbytes <= hi and buf() is not None and (a is buf() or a.base is buf()):
            return True
    return False


def register_host_buffer(buf: np.ndarray, lib_path: Optional[str] = None) -> bool:
    """Pin the memory behind ``buf`` (e.g. ``decode_pool.DecodePool.slab``, a shared-memory array that decoder processes fill) so
    that ``Engine.upload_async`` / ``LanePipeline.run`` copy batches that are views of it straight to the device instead of
    staging them through the engine's own pinned buffer (a 12.6 MB memcpy per batch of 16 512x512 tiles on the thread that feeds
    the GPU).  The caller must leave a batch untouched until the forward that consumed it has delivered its results.  Returns
    False (and changes nothing) if the runtime refuses to pin the range."""
    lib = load_library(lib_path)
    lib.rs_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lo = buf.ctypes.data
    if lib.rs_host_register(C.c_void_p(lo), buf.nbytes) != 0:
        return False
    _REGISTERED_HOST.append((lo, lo + buf.nbytes, weakref.ref(buf)))
    return True


def unregister_host_buffer(buf: np.ndarray, lib_path: Optional[str] = None) -> None:
    lib = load_library(lib_path)
    lib.rs_host_unregister.argtypes = [C.c_void_p]
    lo = buf.ctypes.data
    for r in list(_REGISTERED_HOST):
        if r[0] == lo:
            _REGISTERED_HOST.remove(r)
            lib.rs_host_unregister(C.c_void_p(lo))


class Engine:
    """One engine = one process, one GPU, one tile shape, batches up to ``max_batch``."""

    def __init__(self, spec: EngineSpec, weights: Dict[str, np.ndarray], tile_shape: Tuple[int, int, int], max_batch: int = 16,
                 device: int = 0, stream: Optional[int] = None, lib_path: Optional[str] = None, blob: Optional[bytes] = None):
        """``blob``: the result of ``pack_weights(spec, weights)`` when the caller already has it (LanePipeline packs once for all lanes)."""
        self.lib = load_library(lib_path)
        if self.lib.rs_abi_version() != 1:
            raise RsError("librs_engine.so ABI version mismatch")
        self.spec = spec
        self.tile_h, self.tile_w, self.tile_c = (int(x) for x in tile_shape)
        self.max_batch = int(max_batch)
        self.D = spec.detections_per_image
        if blob is None:
            blob = pack_weights(spec, weights)
        rs = make_rs_spec(spec)
        h = C.c_void_p()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        rc = self.lib.rs_engine_create(C.byref(rs), buf, len(blob), device, self.max_batch, self.tile_h, self.tile_w, self.tile_c,
                                       C.c_void_p(stream) if stream else None, C.byref(h))
        _check(self.lib, rc, "rs_engine_create")
        self._h = h
        self._alloc_out(self.max_batch)

    @classmethod
    def from_handle(cls, lib: C.CDLL, handle: int, spec: EngineSpec, tile_shape: Tuple[int, int, int], max_batch: int) -> "Engine":
        """Wrap an engine owned by someone else (the trainer's forward engine: ``rs_trainer_engine``) for inference calls;
        ``close`` frees only this wrapper's pinned host buffers."""
        self = cls.__new__(cls)
        self.lib, self.spec = lib, spec
        self.tile_h, self.tile_w, self.tile_c = (int(x) for x in tile_shape)
        self.max_batch, self.D = int(max_batch), spec.detections_per_image
        self._h = C.c_void_p(handle)
        self._borrowed = True
        self._alloc_out(self.max_batch)
        return self

    # ------------------------------------------------------------------ plumbing
    def _pinned(self, shape: Tuple[int, ...], dtype) -> np.ndarray:
        """numpy array over pinned host memory (rs_host_alloc): device<->host copies run at PCIe speed and asynchronously."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = self.lib.rs_host_alloc(max(nbytes, 16))
        if not ptr:
            raise RsError(f"rs_host_alloc({nbytes}) failed")
        self._pinned_ptrs.append(ptr)
        a = np.ctypeslib.as_array((C.c_uint8 * max(nbytes, 1)).from_address(ptr))[:nbytes].view(dtype).reshape(shape)
        a[...] = 0
        return a

    def _alloc_out(self, n: int) -> None:
        D, h, wb = self.D, self.tile_h, (self.tile_w + 7) // 8
        self._pinned_ptrs: List[int] = []
        self._count = self._pinned((n,), np.int32)
        self._boxes = self._pinned((n, D, 4), np.float32)
        self._scores = self._pinned((n, D), np.float32)
        self._classes = self._pinned((n, D), np.int32)
        self._masks = self._pinned((n, D, h, wb), np.uint8) if self.spec.mask_on else None
        self._probs = np.zeros((n, D, MASK_SIDE, MASK_SIDE), np.float32) if self.spec.mask_on else None
        self._stage_tiles = self._pinned((n, self.tile_h, self.tile_w, self.tile_c), np.uint8)
        # masks as crops (rs_engine_fetch_crops_*): table + data; the data buffer is pinned lazily on first use
        self._crop_rects = self._pinned((n, D, 4), np.int32) if self.spec.mask_on else None
        self._crop_offsets = self._pinned((n, D), np.uint32) if self.spec.mask_on else None
        self._crop_data: Optional[np.ndarray] = None
        self._crops_struct: Optional[RsMaskCrops] = None
        self._crops_pending = False

    def _dets_struct(self, want_probs: bool) -> RsDets:
        d = RsDets()
        d.count = self._count.ctypes.data_as(C.POINTER(C.c_int32))
        d.boxes = self._boxes.ctypes.data_as(C.POINTER(C.c_float))
        d.scores = self._scores.ctypes.data_as(C.POINTER(C.c_float))
        d.classes = self._classes.ctypes.data_as(C.POINTER(C.c_int32))
        if self._masks is not None:
            d.masks = self._masks.ctypes.data_as(C.POINTER(C.c_uint8))
            if want_probs:
                d.mask_probs = self._probs.ctypes.data_as(C.POINTER(C.c_float))
        return d

    def _collect(self, n: int, want_probs: bool, crops: bool = False) -> List[Instances]:
        out = []
        if crops:
            for i in range(n):
                c = int(self._count[i])
                rects = self._crop_rects[i, :c].copy()
                offs = self._crop_offsets[i, :c].astype(np.int64)
                if c:
                    lo = int(offs[0])
                    hi = int(offs[c - 1]) + int(rects[c - 1, 2]) * int(rects[c - 1, 3])
                    data = self._crop_data[lo:hi].copy()          # the tile's crops are contiguous, in slot order
                    offs = (offs - lo).astype(np.uint32)
                else:
                    data, offs = np.zeros(0, np.uint8), np.zeros(0, np.uint32)
                out.append(Instances((self.tile_h, self.tile_w), self._boxes[i, :c].copy(), self._scores[i, :c].copy(),
                                     self._classes[i, :c].astype(np.int64), None, None, crops=(rects, offs, data)))
            return out
        for i in range(n):
            c = int(self._count[i])
            out.append(Instances((self.tile_h, self.tile_w), self._boxes[i, :c].copy(), self._scores[i, :c].copy(),
                                 self._classes[i, :c].astype(np.int64),
                                 self._masks[i, :c].copy() if self._masks is not None else None,
                                 self._probs[i, :c].copy() if (self._probs is not None and want_probs) else None))
        return out

    # ------------------------------------------------------------------ inference
    def infer(self, tiles: np.ndarray, want_probs: bool = False) -> List[Instances]:
        """tiles: (n, h, w, c) uint8, BGR as ``cv2.imread`` returns them.  Returns one ``Instances`` per tile."""
        tiles = np.ascontiguousarray(tiles)
        if tiles.dtype != np.uint8 or tiles.ndim != 4 or tiles.shape[1:] != (self.tile_h, self.tile_w, self.tile_c):
            raise ValueError(f"tiles must be uint8 (n,{self.tile_h},{self.tile_w},{self.tile_c}), got {tiles.dtype} {tiles.shape}")
        n = tiles.shape[0]
        if not 1 <= n <= self.max_batch:
            raise ValueError(f"batch {n} outside [1, {self.max_batch}]")
        d = self._dets_struct(want_probs)
        rc = self.lib.rs_engine_infer(self._h, tiles.ctypes.data_as(C.c_void_p), n, C.byref(d))
        _check(self.lib, rc, "rs_engine_infer")
        return self._collect(n, want_probs)

    # ------------------------------------------------------------------ asynchronous host interface
    def upload_async(self, tiles: np.ndarray) -> int:
        """Stage ``tiles`` in pinned memory and enqueue their upload on the engine's stream; returns the device pointer."""
        n = tiles.shape[0]
        if tiles.dtype != np.uint8 or tiles.shape[1:] != (self.tile_h, self.tile_w, self.tile_c) or not 1 <= n <= self.max_batch:
            raise ValueError(f"tiles must be uint8 (<= {self.max_batch},{self.tile_h},{self.tile_w},{self.tile_c}), got {tiles.dtype} {tiles.shape}")
        addr = tiles.ctypes.data
        if _registered_view(tiles):
            # the batch already sits in pinned memory (a registered slab, see register_host_buffer): copy straight out of it
            _check(self.lib, self.lib.rs_engine_upload_async(self._h, C.c_void_p(addr), n), "rs_engine_upload_async")
            return self.tensor_ptr("tiles")[0]
        self._stage_tiles[:n] = tiles
        _check(self.lib, self.lib.rs_engine_upload_async(self._h, self._stage_tiles.ctypes.data_as(C.c_void_p), n), "rs_engine_upload_async")
        return self.tensor_ptr("tiles")[0]

    def fetch_async(self, n: int, crops: bool = True) -> None:
        """Enqueue the copy of the last forward's results to the host behind it.  ``crops`` (default, with MASK_ON): the masks
        travel as crops of their boxes (``rs_mask_crops``) instead of full canvases -- 100 x h x w/8 bytes per tile shrink to
        the boxes' area, on the PCIe link and in the host-side copies."""
        d = self._dets_struct(False)
        if crops and self._masks is not None:
            if self._crop_data is None:
                self._crop_data = self._pinned((self.max_batch * self.D * self.tile_h * ((self.tile_w + 7) // 8),), np.uint8)
                c = RsMaskCrops()
                c.rects = self._crop_rects.ctypes.data_as(C.POINTER(C.c_int32))
                c.offsets = self._crop_offsets.ctypes.data_as(C.POINTER(C.c_uint32))
                c.data = self._crop_data.ctypes.data_as(C.POINTER(C.c_uint8))
                c.capacity = self._crop_data.nbytes
                self._crops_struct = c
            d.masks = None
            _check(self.lib, self.lib.rs_engine_fetch_crops_async(self._h, n, C.byref(d), C.byref(self._crops_struct)), "rs_engine_fetch_crops_async")
            self._crops_pending = True
            return
        _check(self.lib, self.lib.rs_engine_fetch_async(self._h, n, C.byref(d)), "rs_engine_fetch_async")

    def wait_results(self) -> None:
        """Block until the copies of the last ``fetch_async`` have landed (for crops: wait for the crop table, copy exactly the
        bytes in use, wait for them)."""
        if self._crops_pending:
            _check(self.lib, self.lib.rs_engine_fetch_crops_wait(self._h, C.byref(self._crops_struct)), "rs_engine_fetch_crops_wait")
            self._crops_pending, self._crops_landed = False, True
        else:
            _check(self.lib, self.lib.rs_engine_fetch_wait(self._h), "rs_engine_fetch_wait")
            self._crops_landed = False

    def collect_results(self, n: int) -> List[Instances]:
        """``Instances`` of the results ``wait_results`` waited for (copies out of the pinned buffers)."""
        return self._collect(n, False, crops=getattr(self, "_crops_landed", False))

    def fetch_wait(self, n: int) -> List[Instances]:
        self.wait_results()
        return self.collect_results(n)

    def infer_device(self, tiles_dev_ptr: int, n: int) -> None:
        """Enqueue a forward on tiles already in device memory (no wait)."""
        _check(self.lib, self.lib.rs_engine_infer_device(self._h, C.c_void_p(tiles_dev_ptr), n), "rs_engine_infer_device")

    def infer_phase(self, tiles_dev_ptr: int, n: int, phase: int) -> None:
        """Enqueue one phase (0, 1, 2) of a forward; see ``LanePipeline`` and include/rs_engine.h."""
        _check(self.lib, self.lib.rs_engine_infer_phase(self._h, C.c_void_p(tiles_dev_ptr), n, phase), "rs_engine_infer_phase")

    def sync(self) -> None:
        _check(self.lib, self.lib.rs_engine_sync(self._h), "rs_engine_sync")

    def fetch(self, n: int, want_probs: bool = False) -> List[Instances]:
        d = self._dets_struct(want_probs)
        _check(self.lib, self.lib.rs_engine_fetch(self._h, n, C.byref(d)), "rs_engine_fetch")
        return self._collect(n, want_probs)

    @property
    def stream(self) -> int:
        return int(self.lib.rs_engine_stream(self._h) or 0)

    def net_shape(self) -> Tuple[int, int, int, int]:
        a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _check(self.lib, self.lib.rs_engine_net_shape(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "net_shape")
        return a.value, b.value, c.value, d.value

    # ------------------------------------------------------------------ introspection
    def tensor_names(self) -> List[str]:
        buf = C.create_string_buffer(96)
        names = []
        for i in range(self.lib.rs_engine_tensor_count(self._h)):
            self.lib.rs_engine_tensor_name(self._h, i, buf)
            names.append(buf.value.decode())
        return names

    def tensor_ptr(self, name: str) -> Tuple[int, np.dtype, Tuple[int, ...], int]:
        p, dt, nd, halo = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32()
        dims = (C.c_int64 * 5)()
        _check(self.lib, self.lib.rs_engine_tensor(self._h, name.encode(), C.byref(p), C.byref(dt), C.byref(nd), dims, C.byref(halo)),
               f"rs_engine_tensor({name})")
        return int(p.value), np.dtype(DT_NP[dt.value]), tuple(int(dims[i]) for i in range(nd.value)), halo.value

    def tensor(self, name: str, n: Optional[int] = None, strip_halo: bool = True) -> np.ndarray:
        """Copy an intermediate tensor to the host.  NHWC activations lose their halo; ``n`` limits the
        leading (batch) dimension."""
        ptr, dt, shape, halo = self.tensor_ptr(name)
        if self.tensor_is_split(name):
            # split-operand mode: two fp16 planes of the full shape back to back; fp32(hi) + fp32(lo) is exact
            planes = np.empty((2,) + shape, np.float16)
            _check(self.lib, self.lib.rs_memcpy_d2h(planes.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), planes.nbytes), "rs_memcpy_d2h")
            a = planes[0].astype(np.float32) + planes[1].astype(np.float32)
            if n is not None:
                a = a[: min(n, shape[0])]
        else:
            if n is not None:
                shape = (min(n, shape[0]),) + shape[1:]
            a = np.empty(shape, dt)
            _check(self.lib, self.lib.rs_memcpy_d2h(a.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), a.nbytes), "rs_memcpy_d2h")
        if strip_halo and halo:
            a = a[:, halo:-halo, halo:-halo]
        return a

    def tensor_is_split(self, name: str) -> bool:
        dt = C.c_int32()
        _check(self.lib, self.lib.rs_engine_tensor(self._h, name.encode(), None, C.byref(dt), None, None, None), f"rs_engine_tensor({name})")
        return dt.value == DT_SPLIT16

    def set_profiling(self, mode: int) -> None:
        """0 off, 1 per-stage events + host wait, 2 events only (read back by ``stage_times``)."""
        _check(self.lib, self.lib.rs_engine_set_profiling(self._h, int(mode)), "rs_engine_set_profiling")

    def upload_tiles(self, tiles: np.ndarray) -> int:
        """Copy tiles into the engine's own device input buffer; returns its device pointer so that
        ``infer_device`` can run on tiles already resident in HBM (no staging copy)."""
        tiles = np.ascontiguousarray(tiles)
        assert tiles.dtype == np.uint8 and tiles.shape[1:] == (self.tile_h, self.tile_w, self.tile_c) and tiles.shape[0] <= self.max_batch
        ptr, _, _, _ = self.tensor_ptr("tiles")
        _check(self.lib, self.lib.rs_memcpy_h2d(C.c_void_p(ptr), tiles.ctypes.data_as(C.c_void_p), tiles.nbytes), "rs_memcpy_h2d")
        return ptr

    def stage_times(self) -> List[Dict[str, Any]]:
        out = []
        name = C.create_string_buffer(96)
        ms, fl, by, calls = C.c_double(), C.c_double(), C.c_double(), C.c_int32()
        for i in range(self.lib.rs_engine_stage_count(self._h)):
            self.lib.rs_engine_stage_info(self._h, i, name, C.byref(ms), C.byref(calls), C.byref(fl), C.byref(by))
            kn = C.create_string_buffer(96)
            self.lib.rs_engine_stage_kernel(self._h, i, kn)
            out.append({"name": name.value.decode(), "ms_total": ms.value, "calls": calls.value, "flops": fl.value, "bytes": by.value,
                        "kernel": kn.value.decode()})
        return out

    def stage_variants(self) -> Dict[str, int]:
        """Conv tile variant each GEMM stage launched in its last call (numbering: include/rs_engine.h rs_op_conv_variant)."""
        name = C.create_string_buffer(96)
        out = {}
        for i in range(self.lib.rs_engine_stage_count(self._h)):
            v = self.lib.rs_engine_stage_variant(self._h, i)
            if v != -2:
                self.lib.rs_engine_stage_info(self._h, i, name, None, None, None, None)
                out[name.value.decode()] = v
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self.lib.rs_engine_destroy(self._h)
            self._h = None
            self._crops_struct = None
            for name in ("_count", "_boxes", "_scores", "_classes", "_masks", "_stage_tiles", "_crop_rects", "_crop_offsets", "_crop_data"):
                setattr(self, name, None)             # views over the pinned memory freed below
            for p in getattr(self, "_pinned_ptrs", []):
                self.lib.rs_host_free(p)
            self._pinned_ptrs = []

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass


class LanePipeline:
    """Several engines ("lanes") fed alternately so that consecutive batches overlap on one GPU.  The reference processes one tile at a time
    and has no counterpart ([EXT d2: engine/defaults.py]).  Two forms:

    * independent lanes (default since the end of round 4): every lane is a complete engine on its OWN stream and a batch is one ``infer_device``
      on its lane; the hardware interleaves the workgroups of the lanes' kernels.  On one stream every CU reaches the HBM-bound prologue / epilogue of
      a ``conv_deep`` tile together and the matrix pipe idles meanwhile; kernels of two queues run out of phase, and one lane's
      latency-bound detection glue runs beside the other's convolutions.  Against the phased form (``tools/ubench/two_pipes.py``, one box): split mode
      905 -> 954 tiles/s at batch 16, 821 -> 891 at batch 8; fp16 2 189 -> 2 241 and 1 796 -> 2 068.  Bit-identical to a single engine -- once the
      device code stopped using the compiler's fp32 division sequence, which returns wrong quotients beside another kernel's MFMA waves
      (csrc/common.h ``rs_fdiv``; ``tools/parity/lanes_stress.py``: 8 400 tile results, none differing).
    * ``shared_stream=True`` (rounds 2-4): the lanes' convolutions serialised on ONE wide stream, each lane's glue on its own side stream
      behind the other lane's convolutions.  Enqueue order for batch k on lane k mod 2:

          phase0(k)   backbone, FPN, RPN heads  -> glue G1(k) on the side stream
          phase2(k-1) mask head of the previous batch (hides G1(k))
          phase1(k)   box head                  -> glue G2(k) on the side stream, hidden by phase0(k+1)

      Results of batch k are complete once phase2(k) has run, i.e. after ``submit`` of batch k+1 or ``flush``."""

    def __init__(self, spec: EngineSpec, weights: Dict[str, np.ndarray], tile_shape: Tuple[int, int, int], max_batch: int = 16,
                 device: int = 0, lanes: int = 2, shared_stream: bool = False):
        if lanes not in (1, 2, 3, 4):
            raise ValueError("lanes must be 1..4")
        blob = pack_weights(spec, weights)           # folding + fragment orders once, not once per lane (0.3 s of host time each)
        self.shared = bool(shared_stream) and lanes > 1
        first = Engine(spec, weights, tile_shape, max_batch, device, blob=blob)
        self.engines = [first] + [Engine(spec, weights, tile_shape, max_batch, device, stream=first.stream if self.shared else None, blob=blob)
                                  for _ in range(lanes - 1)]
        self.k = 0
        self._pending: Optional[Tuple[int, int, int]] = None     # shared form: (lane, tiles ptr, n) whose phase 2 is still to be enqueued

    def lane_of_next(self) -> Engine:
        return self.engines[self.k % len(self.engines)]

    def submit(self, tiles_dev_ptr: int, n: int) -> int:
        """Enqueue one batch (tiles resident in the next lane's device memory); returns the lane index used."""
        lane = self.k % len(self.engines)
        e = self.engines[lane]
        self.k += 1
        if not self.shared:
            e.infer_device(tiles_dev_ptr, n)
            return lane
        e.infer_phase(tiles_dev_ptr, n, 0)
        self._finish_pending()
        e.infer_phase(tiles_dev_ptr, n, 1)
        self._pending = (lane, tiles_dev_ptr, n)
        return lane

    def _finish_pending(self) -> None:
        if self._pending is not None:
            lane, ptr, n = self._pending
            self.engines[lane].infer_phase(ptr, n, 2)
            self._pending = None

    def flush(self) -> None:
        """Enqueue the outstanding mask-head phase; after this every submitted batch is fully enqueued."""
        self._finish_pending()

    def run(self, batches) -> "Iterator[List[Instances]]":
        """Stream host batches ((n,h,w,c) uint8 arrays) through the lanes and yield their detections in order.  Uploads go
        through pinned staging on the lane's stream, every batch's results come back on its lane's copy stream behind an event,
        and the generator runs ``lanes`` batches ahead of what it yields, so host-side collection overlaps the GPU.  Independent lanes:
            submit(k): upload(k), forward(k), fetch_async(k);   then collect + yield batch k - lanes
        shared stream:
            submit(k): upload(k), phase0(k), phase2(k-1) + fetch_async(k-1), phase1(k);   then collect + yield batch k-2."""
        L = len(self.engines)
        inflight = []                               # (lane, n) of submitted batches not yet yielded
        T = self.timing = {"pull": 0.0, "wait_results": 0.0, "upload": 0.0, "enqueue": 0.0, "collect": 0.0, "batches": 0}
        clock = time.perf_counter
        it = iter(batches)
        while True:
            t0 = clock()
            tiles = next(it, None)                  # the source's time (decode wait) is not the pipeline's
            T["pull"] += clock() - t0
            if tiles is None:
                break
            T["batches"] += 1
            n = int(tiles.shape[0])
            lane = self.k % L
            e = self.engines[lane]
            self.k += 1
            if L == 1:                              # one lane has one set of host buffers: no look-ahead
                e.infer_device(e.upload_async(np.ascontiguousarray(tiles)), n)
                e.fetch_async(n)
                yield e.fetch_wait(n)
                continue
            done = None
            if len(inflight) == L:
                # this lane's previous batch: its result copy was enqueued one submit ago; once it has landed the lane's pinned
                # staging buffer (upload) is free again.  The other lane's batch keeps the GPU busy meanwhile.
                ol, on = inflight.pop(0)
                t0 = clock()
                self.engines[ol].wait_results()
                T["wait_results"] += clock() - t0
                done = (ol, on)
            t0 = clock()
            ptr = e.upload_async(np.ascontiguousarray(tiles))
            t1 = clock()
            T["upload"] += t1 - t0
            if not self.shared:
                e.infer_device(ptr, n)
                T["enqueue"] += clock() - t1
                res = None
                if done is not None:                # the lane's previous results leave its pinned buffers BEFORE the next copy into them is enqueued
                    t0 = clock()
                    res = self.engines[done[0]].collect_results(done[1])
                    T["collect"] += clock() - t0
                t1 = clock()
                e.fetch_async(n)
                T["enqueue"] += clock() - t1
                inflight.append((lane, n))
                if res is not None:
                    yield res
                continue
            e.infer_phase(ptr, n, 0)
            if self._pending is not None:
                pl, pp, pn = self._pending
                self.engines[pl].infer_phase(pp, pn, 2)
                self.engines[pl].fetch_async(pn)
                self._pending = None
            e.infer_phase(ptr, n, 1)
            T["enqueue"] += clock() - t1
            self._pending = (lane, ptr, n)
            inflight.append((lane, n))
            if done is not None:                    # host-side collection overlaps the batches just enqueued; this lane's next
                t0 = clock()
                res = self.engines[done[0]].collect_results(done[1])      # fetch_async comes one submit later
                T["collect"] += clock() - t0
                yield res
        if self._pending is not None:
            pl, pp, pn = self._pending
            self.engines[pl].infer_phase(pp, pn, 2)
            self.engines[pl].fetch_async(pn)
            self._pending = None
        for ol, on in inflight:
            yield self.engines[ol].fetch_wait(on)

    def sync(self) -> None:
        self.flush()
        for e in self.engines:
            e.sync()

    def close(self) -> None:
        for e in reversed(self.engines):     # lane 0 owns the shared stream: destroy it last
            e.close()


class Predictor:
    """Drop-in for ``detectron2.engine.DefaultPredictor``: ``predictor(im_bgr) -> {"instances": Instances}``.
    One two-lane pipeline is built per tile shape on first use (tilesets are single-shape: 256^2/512^2 z18 tiles,
    R:config/config_obj_detec.yaml:45).  ``predict_batch`` is the batched, streaming form the CLI shim uses."""

    def __init__(self, spec: EngineSpec, weights: Dict[str, np.ndarray], max_batch: int = 16, device: int = 0, lanes: int = 2):
        self.spec, self.weights, self.max_batch, self.device, self.lanes = spec, weights, max_batch, device, lanes
        self._pipes: Dict[Tuple[int, int, int], LanePipeline] = {}

    def _pipe(self, shape: Tuple[int, int, int]) -> LanePipeline:
        if shape not in self._pipes:
            self._pipes[shape] = LanePipeline(self.spec, self.weights, shape, self.max_batch, self.device, self.lanes)
        return self._pipes[shape]

    def prepare(self, shape: Tuple[int, int, int], warm: bool = True) -> None:
        """Build the lane pipeline for tiles of ``shape`` now instead of at the first batch, and (``warm``) run one blank tile through
        every lane so that the code objects are loaded and the per-kernel attributes set -- the CLI calls this while its decoder
        processes are still starting, the way the reference builds its ``DefaultPredictor`` before it loops over the tiles."""
        pipe = self._pipe(tuple(int(x) for x in shape))
        if warm:
            blank = np.zeros((1,) + tuple(int(x) for x in shape), np.uint8)
            for e in pipe.engines:
                e.infer(blank)

    def __call__(self, original_image: np.ndarray) -> Dict[str, Instances]:
        if original_image.ndim != 3:
            raise ValueError("expected an HWC image")
        eng = self._pipe(tuple(original_image.shape)).engines[0]
        return {"instances": eng.infer(original_image[None])[0]}

    def predict_batch(self, images: Sequence[np.ndarray]) -> List[Dict[str, Instances]]:
        """Any number of images; runs of equal shape are cut into batches of ``max_batch`` and streamed through the lanes
        (``LanePipeline.run``: uploads, forwards and result copies of consecutive batches overlap)."""
        out: List[Dict[str, Instances]] = []
        i = 0
        while i < len(images):
            shape = tuple(images[i].shape)
            j = i
            while j < len(images) and tuple(images[j].shape) == shape:
                j += 1
            chunks = (np.stack(images[k:min(k + self.max_batch, j)]) for k in range(i, j, self.max_batch))
            for res in self._pipe(shape).run(chunks):
                out.extend({"instances": r} for r in res)
            i = j
        return out

    def predict_stream(self, batches) -> "Iterator[List[Dict[str, Instances]]]":
        """Streaming form for a whole tileset: ``batches`` is an iterator of image lists (each at most ``max_batch`` images);
        yields one result list per batch, in order.  Consecutive batches of one tile shape flow through ONE
        ``LanePipeline.run`` generator, so the upload, forward and result copy of batch k+1 overlap batch k (calling
        ``predict_batch`` once per batch would drain the pipeline after every batch).  The iterator is pulled lazily, two
        batches ahead of what is yielded."""
        it = iter(batches)
        carry: List[Any] = []            # a batch read ahead that does not fit the running pipeline

        def next_batch():
            return carry.pop() if carry else next(it, None)

        while True:
            first = next_batch()
            if first is None:
                return
            shapes = {tuple(im.shape) for im in first}
            if len(shapes) != 1 or len(first) > self.max_batch:
                yield self.predict_batch(first)          # mixed shapes inside one batch: no streaming for it
                continue
            shape = shapes.pop()

            def run_of_shape(b=first):
                while b is not None:
                    if len(b) > self.max_batch or any(tuple(im.shape) != shape for im in b):
                        carry.append(b)
                        return
                    yield b if isinstance(b, np.ndarray) else np.stack(b)      # a decoded batch in shared memory is used in place
                    b = next_batch()
            for res in self._pipe(shape).run(run_of_shape()):
                yield [{"instances": r} for r in res]

    def close(self) -> None:
        for p in self._pipes.values():
            p.close()
        self._pipes.clear()


class Trainer:
    """Training engine (include/rs_engine.h ``rs_trainer_*``): forward engine + flat fp32
    master / gradient / momentum buffers + backward stage list.  ``train_step`` = ``SimpleTrainer.run_step`` up to
    ``losses.backward()`` (forward, label assignment + sampling, the five losses, backward of every trainable layer);
    ``apply_sgd`` = the optimiser step; ``allreduce_gradients`` = DDP's gradient averaging over RCCL."""

    def __init__(self, spec: EngineSpec, weights: Dict[str, np.ndarray], tile_shape: Tuple[int, int, int], batch: int = 2,
                 device: int = 0, loss_scale: float = 1.0, lib_path: Optional[str] = None):
        self.lib = lib = load_library(lib_path)
        vp, i32 = C.c_void_p, C.c_int
        lib.rs_trainer_create.argtypes = [C.POINTER(RsSpec), vp, C.c_size_t, i32, i32, i32, i32, i32, C.c_float, C.POINTER(vp)]
        lib.rs_trainer_destroy.argtypes = [vp]
        lib.rs_trainer_destroy.restype = None
        lib.rs_trainer_engine.argtypes = [vp]
        lib.rs_trainer_engine.restype = vp
        lib.rs_trainer_forward_trunk.argtypes = [vp, vp, i32]
        lib.rs_trainer_backward_trunk.argtypes = [vp, i32]
        lib.rs_trainer_apply_sgd.argtypes = [vp, C.c_float, C.c_float, C.c_float]
        lib.rs_trainer_set_targets.argtypes = [vp, vp, vp, vp, i32, i32]
        lib.rs_trainer_set_image_sizes.argtypes = [vp, vp, vp, i32]
        lib.rs_trainer_rpn_step.argtypes = [vp, i32, C.c_uint32, i32]
        lib.rs_trainer_rpn_forward.argtypes = [vp, i32]
        lib.rs_trainer_rpn_targets_async.argtypes = [vp, i32, C.c_uint32]
        lib.rs_trainer_roi_step.argtypes = [vp, i32, C.c_uint32]
        lib.rs_trainer_set_sampling.argtypes = [vp, i32, C.c_float, i32, C.c_float]
        lib.rs_trainer_set_rpn_topk.argtypes = [vp, i32, i32]
        lib.rs_trainer_set_grad_divisor.argtypes = [vp, C.c_float]
        lib.rs_trainer_copy_state.argtypes = [vp, vp]
        lib.rs_trainer_grad_buffer.argtypes = [vp]
        lib.rs_trainer_set_loss_scale.argtypes = [vp, C.c_float]
        lib.rs_trainer_fetch_rois.argtypes = [vp, C.c_int, vp, vp, vp]
        lib.rs_trainer_grad_buffer.restype = vp
        lib.rs_trainer_master_buffer.argtypes = [vp]
        lib.rs_trainer_master_buffer.restype = vp
        lib.rs_trainer_bucket_count.argtypes = [vp]
        lib.rs_trainer_bucket_info.argtypes = [vp, i32, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.rs_trainer_bucket_wait.argtypes = [vp, i32, vp]
        lib.rs_trainer_bucket_sync.argtypes = [vp, i32]
        lib.rs_trainer_wait_stream.argtypes = [vp, vp]
        lib.rs_trainer_mask_forward.argtypes = [vp, i32]
        lib.rs_trainer_mask_backward.argtypes = [vp, i32, vp, i32]
        lib.rs_trainer_sync.argtypes = [vp]
        lib.rs_trainer_set_profiling.argtypes = [vp, i32]
        lib.rs_trainer_stage_count.argtypes = [vp]
        lib.rs_trainer_stage_info.argtypes = [vp, i32, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        lib.rs_trainer_tensor.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        lib.rs_trainer_tensor_count.argtypes = [vp]
        lib.rs_trainer_tensor_name.argtypes = [vp, i32, C.c_char_p]
        lib.rs_trainer_param_count.argtypes = [vp]
        lib.rs_trainer_param_count.restype = C.c_int64
        self.spec = spec
        self.tile_h, self.tile_w, self.tile_c = (int(x) for x in tile_shape)
        self.batch = int(batch)
        blob = pack_weights(spec, weights, train=True)
        rs = make_rs_spec(spec)
        h = C.c_void_p()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        _check(lib, lib.rs_trainer_create(C.byref(rs), buf, len(blob), device, self.batch, self.tile_h, self.tile_w, self.tile_c,
                                          float(loss_scale), C.byref(h)), "rs_trainer_create")
        self._h = h
        self._eng = C.c_void_p(lib.rs_trainer_engine(h))

    def _tensor_ptr(self, name: str, engine: bool = False):
        p, dt, nd, halo = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32()
        dims = (C.c_int64 * 5)()
        if engine:
            _check(self.lib, self.lib.rs_engine_tensor(self._eng, name.encode(), C.byref(p), C.byref(dt), C.byref(nd), dims, C.byref(halo)), f"tensor {name}")
        else:
            _check(self.lib, self.lib.rs_trainer_tensor(self._h, name.encode(), C.byref(p), C.byref(dt), C.byref(nd), dims, C.byref(halo)), f"tensor {name}")
        return int(p.value), np.dtype(DT_NP[dt.value]), tuple(int(dims[i]) for i in range(nd.value)), halo.value

    def tensor(self, name: str, engine: bool = False, strip_halo: bool = True) -> np.ndarray:
        """Copy a trainer tensor ("d:<act>", "g:<layer>.w", "m:<layer>.w") or, with ``engine=True``, a forward tensor to the host."""
        ptr, dt, shape, halo = self._tensor_ptr(name, engine)
        a = np.empty(shape, dt)
        self.sync()                     # the trainer's stream is non-blocking: a plain memcpy would not wait for it
        _check(self.lib, self.lib.rs_memcpy_d2h(a.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), a.nbytes), "rs_memcpy_d2h")
        if strip_halo and halo:
            a = a[:, halo:-halo, halo:-halo]
        return a

    def set_tensor(self, name: str, interior: np.ndarray) -> None:
        """Write the interior of an activation-gradient tensor (the halo stays zero)."""
        ptr, dt, shape, halo = self._tensor_ptr(name)
        full = np.zeros(shape, dt)
        n = interior.shape[0]
        if halo:
            full[:n, halo:-halo, halo:-halo] = interior.astype(dt)
        else:
            full[:n] = interior.astype(dt)
        _check(self.lib, self.lib.rs_memcpy_h2d(C.c_void_p(ptr), full.ctypes.data_as(C.c_void_p), full.nbytes), "rs_memcpy_h2d")

    def tensor_names(self) -> List[str]:
        buf = C.create_string_buffer(96)
        out = []
        for i in range(self.lib.rs_trainer_tensor_count(self._h)):
            self.lib.rs_trainer_tensor_name(self._h, i, buf)
            out.append(buf.value.decode())
        return out

    def upload_tiles(self, tiles: np.ndarray) -> int:
        tiles = np.ascontiguousarray(tiles)
        assert tiles.dtype == np.uint8 and tiles.shape[1:] == (self.tile_h, self.tile_w, self.tile_c) and tiles.shape[0] <= self.batch
        ptr, _, _, _ = self._tensor_ptr("tiles", engine=True)
        _check(self.lib, self.lib.rs_memcpy_h2d(C.c_void_p(ptr), tiles.ctypes.data_as(C.c_void_p), tiles.nbytes), "rs_memcpy_h2d")
        return ptr

    def forward_trunk(self, tiles_ptr: int, n: int) -> None:
        _check(self.lib, self.lib.rs_trainer_forward_trunk(self._h, C.c_void_p(tiles_ptr), n), "rs_trainer_forward_trunk")

    def set_targets(self, gt_boxes: Sequence[np.ndarray], gt_classes: Sequence[np.ndarray]) -> None:
        """Ground truth per image: boxes (k_i, 4) in NETWORK-INPUT pixels, classes (k_i,)."""
        n = len(gt_boxes)
        cap = max(1, max(int(b.shape[0]) for b in gt_boxes))
        bx = np.zeros((n, cap, 4), np.float32)
        cl = np.zeros((n, cap), np.int32)
        cnt = np.zeros(n, np.int32)
        for i, (b, c) in enumerate(zip(gt_boxes, gt_classes)):
            k = int(b.shape[0])
            bx[i, :k], cl[i, :k], cnt[i] = b, c, k
        _check(self.lib, self.lib.rs_trainer_set_targets(self._h, bx.ctypes.data_as(C.c_void_p), cl.ctypes.data_as(C.c_void_p),
                                                         cnt.ctypes.data_as(C.c_void_p), n, cap), "rs_trainer_set_targets")

    def set_image_sizes(self, sizes: Optional[Sequence[int]]) -> List[Tuple[int, int]]:
        """``INPUT.MIN_SIZE_TRAIN`` drawn PER IMAGE (R:config/detectron2_config_3bands.yaml:31-38; [EXT d2: DatasetMapper ->
        ResizeShortestEdge per record, ImageList.from_tensors pads the batch to its largest image]): image i of the coming
        batches is resized to shortest edge ``sizes[i]`` inside this trainer's canvas (whose own size must be the largest of the
        batch), zeros beyond it, proposals clipped to it.  ``None`` / empty: one size for all again.  Returns the (h, w) per
        image -- ground truth of image i is in THOSE pixels."""
        from .spec import resize_shortest_edge_shape
        if not sizes:
            _check(self.lib, self.lib.rs_trainer_set_image_sizes(self._h, None, None, 0), "rs_trainer_set_image_sizes")
            self._image_sizes = None
            return []
        hw = [resize_shortest_edge_shape(self.tile_h, self.tile_w, int(s), self.spec.max_size_test) for s in sizes]
        if list(sizes) == getattr(self, "_image_sizes", None):
            return hw
        nh = np.asarray([h for h, _ in hw], np.int32)
        nw = np.asarray([w for _, w in hw], np.int32)
        _check(self.lib, self.lib.rs_trainer_set_image_sizes(self._h, nh.ctypes.data_as(C.c_void_p), nw.ctypes.data_as(C.c_void_p), len(hw)),
               "rs_trainer_set_image_sizes")
        self._image_sizes = list(sizes)
        return hw

    def rpn_step(self, n: int, seed: int = 1, external_labels: bool = False) -> None:
        _check(self.lib, self.lib.rs_trainer_rpn_step(self._h, n, seed & 0xFFFFFFFF, int(external_labels)), "rs_trainer_rpn_step")

    def rpn_forward(self, n: int) -> None:
        _check(self.lib, self.lib.rs_trainer_rpn_forward(self._h, n), "rs_trainer_rpn_forward")

    def roi_step(self, n: int, seed: int = 1) -> None:
        _check(self.lib, self.lib.rs_trainer_roi_step(self._h, n, seed & 0xFFFFFFFF), "rs_trainer_roi_step")

    def mask_forward(self, n: int) -> None:
        _check(self.lib, self.lib.rs_trainer_mask_forward(self._h, n), "rs_trainer_mask_forward")

    def mask_backward(self, n: int, targets: np.ndarray) -> None:
        """targets: (n_entries, 28, 28) bool/uint8 gt masks of the mask-head entries, in entry order."""
        t = np.ascontiguousarray(targets.astype(np.uint8))
        _check(self.lib, self.lib.rs_trainer_mask_backward(self._h, n, t.ctypes.data_as(C.c_void_p) if t.size else None, int(t.shape[0])),
               "rs_trainer_mask_backward")

    def mask_entries(self, gt_polygons: Sequence[Sequence[Sequence[np.ndarray]]], n: int) -> Tuple[np.ndarray, List[Tuple[int, int]]]:
        """Host part of the mask branch: gt masks of the sampled foreground RoIs (``PolygonMasks.crop_and_resize``).
        gt_polygons[image][gt index] = list of polygons ([x0,y0,...], network-input pixels).  Returns (targets, [(image, slot)])."""
        from .train_targets import rasterize_entries
        boxes = np.empty((n, 1024, 4), np.float32)
        gti = np.empty((n, 1024), np.int32)
        cnt = np.empty((n, 2), np.int32)
        _check(self.lib, self.lib.rs_trainer_fetch_rois(self._h, n, boxes.ctypes.data_as(C.c_void_p), gti.ctypes.data_as(C.c_void_p),
                                                        cnt.ctypes.data_as(C.c_void_p)), "rs_trainer_fetch_rois")
        side = 2 * self.spec.mask_pooler_resolution
        first = np.zeros(n + 1, np.int64)                      # instance index of image i's gt j = first[i] + j
        first[1:] = np.cumsum([len(gt_polygons[i]) for i in range(n)])
        instances = [polys for i in range(n) for polys in gt_polygons[i]]
        where = [(i, j) for i in range(n) for j in range(min(int(cnt[i, 0]), 256))]
        if not where:
            return np.zeros((0, side, side), bool), where
        ii = np.array([w[0] for w in where]); jj = np.array([w[1] for w in where])
        return rasterize_entries(instances, first[ii] + gti[ii, jj], boxes[ii, jj], side), where

    # ------------------------------------------------------------------ a whole step
    def train_step(self, tiles: np.ndarray, gt_boxes: Sequence[np.ndarray], gt_classes: Sequence[np.ndarray],
                   gt_polygons: Optional[Sequence[Sequence[Sequence[np.ndarray]]]], seed: int, allreduce: bool = False,
                   sizes: Optional[Sequence[int]] = None) -> Dict[str, float]:
        """Forward + losses + backward of one batch (``SimpleTrainer.run_step`` up to ``losses.backward()``): gradients end
        up in the flat gradient buffer; returns the five losses.  ``gt_*`` in NETWORK-INPUT pixels.  ``allreduce``: enqueue
        the data-parallel gradient all-reduce (``allreduce_gradients``) behind the backward pass before the losses are read
        back, so that the collectives of the early buckets overlap the rest of the backward.  ``sizes``: shortest-edge size
        per image (``set_image_sizes``); ``None`` keeps whatever was set last."""
        n = int(tiles.shape[0])
        if sizes is not None:
            self.set_image_sizes(sizes if any(int(s) != self.spec.min_size_test for s in sizes) else None)
        self.set_targets(gt_boxes, gt_classes)
        # the RPN's targets depend on the ground truth and the seed only: on the side stream, ahead of the forward pass
        _check(self.lib, self.lib.rs_trainer_rpn_targets_async(self._h, n, seed & 0xFFFFFFFF), "rs_trainer_rpn_targets_async")
        self.forward_trunk(self.upload_tiles(tiles), n)
        self.rpn_forward(n)
        self.roi_step(n, seed)
        if self.spec.mask_on:
            self.mask_forward(n)
            targets, _ = self.mask_entries(gt_polygons, n)
            self.mask_backward(n, targets)
        self.rpn_step(n, seed)
        self.backward_trunk(n)
        if allreduce:
            self.allreduce_gradients()
        l = self.tensor("losses")
        names = ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg", "loss_mask")
        return {k: float(l[i]) for i, k in enumerate(names)}

    def copy_state_from(self, other: "Trainer") -> None:
        """Carry master weights + momentum over from ``other`` (a trainer of another input size) and refold."""
        _check(self.lib, self.lib.rs_trainer_copy_state(self._h, other._h), "rs_trainer_copy_state")

    def inference_engine(self) -> Engine:
        """The trainer's forward engine as an inference ``Engine`` (same device buffers, current weights): validation detections
        without a second copy of the network.  Valid until the trainer is closed; do not interleave with a training step."""
        if getattr(self, "_infer", None) is None:
            self._infer = Engine.from_handle(self.lib, int(self._eng.value), self.spec, (self.tile_h, self.tile_w, self.tile_c), self.batch)
        return self._infer

    def set_profiling(self, on: bool) -> None:
        """HIP events around every stage of the following training steps (``stage_times`` reads them)."""
        _check(self.lib, self.lib.rs_trainer_set_profiling(self._h, 1 if on else 0), "rs_trainer_set_profiling")

    def stage_times(self) -> List[Dict[str, Any]]:
        """Per stage of the training step since ``set_profiling(True)``: name, ms_total, calls, algorithmic flops of one execution,
        side (True: runs on the weight-gradient side stream, next to the chain)."""
        out = []
        name = C.create_string_buffer(96)
        ms, fl, calls, sd = C.c_double(), C.c_double(), C.c_int32(), C.c_int32()
        for i in range(self.lib.rs_trainer_stage_count(self._h)):
            _check(self.lib, self.lib.rs_trainer_stage_info(self._h, i, name, C.byref(ms), C.byref(calls), C.byref(fl), C.byref(sd)), "rs_trainer_stage_info")
            out.append({"name": name.value.decode(), "ms_total": ms.value, "calls": calls.value, "flops": fl.value, "side": bool(sd.value)})
        return out

    def buckets(self) -> List[Tuple[str, int, int]]:
        """Gradient buckets (name, offset, count in floats of the flat gradient buffer) in the order a step completes them."""
        out = []
        name = C.create_string_buffer(96)
        off, cnt = C.c_int64(), C.c_int64()
        for i in range(self.lib.rs_trainer_bucket_count(self._h)):
            _check(self.lib, self.lib.rs_trainer_bucket_info(self._h, i, name, C.byref(off), C.byref(cnt)), "rs_trainer_bucket_info")
            out.append((name.value.decode(), int(off.value), int(cnt.value)))
        return out

    def flat(self, which: str = "grad") -> np.ndarray:
        """Host copy of the whole flat fp32 gradient ("grad") or master-weight ("master") buffer (waits for the step)."""
        ptr = int(self.lib.rs_trainer_grad_buffer(self._h) if which == "grad" else self.lib.rs_trainer_master_buffer(self._h))
        a = np.empty(self.param_count, np.float32)
        self.sync()
        _check(self.lib, self.lib.rs_memcpy_d2h(a.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), a.nbytes), "rs_memcpy_d2h")
        return a

    def write_flat_grad(self, g: np.ndarray) -> None:
        g = np.ascontiguousarray(g, np.float32)
        assert g.shape == (self.param_count,)
        self.sync()
        _check(self.lib, self.lib.rs_memcpy_h2d(C.c_void_p(int(self.lib.rs_trainer_grad_buffer(self._h))), g.ctypes.data_as(C.c_void_p), g.nbytes), "rs_memcpy_h2d")

    def allreduce_gradients(self, force: bool = False) -> None:
        """DistributedDataParallel's gradient averaging: SUM every gradient bucket over the ranks of the default process group
        and set the divisor the SGD step applies.  Buckets are reduced in the order the step completes them (heads, FPN,
        res5, res4, res3 -- ``buckets()``), each behind its own completion events:

        * RCCL (backend "nccl"): the bucket is handed to torch.distributed in place (``__cuda_array_interface__`` view of the
          device buffer) as an ASYNCHRONOUS all-reduce; torch's current stream is first made to wait (device side) for the
          bucket's events, so the call returns at once and the collective runs over xGMI while the trainer's streams are still
          computing the later buckets' gradients.  Afterwards the trainer's stream waits for the collectives (one event).
          One process per GPU; the host never blocks here.
        * gloo (CPU tests, several ranks on one card): per bucket, wait for its events on the host, reduce through a host copy.

        Call it after the backward pass has been ENQUEUED (``train_step(..., allreduce=True)`` does, before it reads the
        losses back) and before ``apply_sgd``."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
            return                      # ``force``: run the collectives even in a one-rank group (exercises the RCCL path on one GPU)
        ptr = int(self.lib.rs_trainer_grad_buffer(self._h))
        if dist.get_backend() == "nccl":
            cur = torch.cuda.current_stream()
            works = []
            for i, (_, off, cnt) in enumerate(self.buckets()):
                _check(self.lib, self.lib.rs_trainer_bucket_wait(self._h, i, C.c_void_p(cur.cuda_stream)), "rs_trainer_bucket_wait")

                class _Buf:                      # zero-copy view of the bucket inside the device buffer
                    __cuda_array_interface__ = {"shape": (cnt,), "typestr": "<f4", "data": (ptr + 4 * off, False), "version": 2}
                t = torch.as_tensor(_Buf(), device="cuda")
                works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))
            for w in works:
                w.wait()                         # torch's current stream waits for the collective (no host block)
            _check(self.lib, self.lib.rs_trainer_wait_stream(self._h, C.c_void_p(cur.cuda_stream)), "rs_trainer_wait_stream")
        else:
            for i, (_, off, cnt) in enumerate(self.buckets()):
                _check(self.lib, self.lib.rs_trainer_bucket_sync(self._h, i), "rs_trainer_bucket_sync")
                host = np.empty(cnt, np.float32)
                _check(self.lib, self.lib.rs_memcpy_d2h(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr + 4 * off), host.nbytes), "rs_memcpy_d2h")
                t = torch.from_numpy(host)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                _check(self.lib, self.lib.rs_memcpy_h2d(C.c_void_p(ptr + 4 * off), host.ctypes.data_as(C.c_void_p), host.nbytes), "rs_memcpy_h2d")
        _check(self.lib, self.lib.rs_trainer_set_grad_divisor(self._h, float(dist.get_world_size())), "rs_trainer_set_grad_divisor")

    def export_weights(self, base: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
        """Current weights under detectron2's checkpoint key names: ``base`` (the weights the trainer was created from) with
        every trainable tensor replaced by its fp32 master copy, converted back from the engine's GEMM layouts."""
        from .weights import master_to_d2
        return master_to_d2(self.spec, base, lambda name: self.tensor(name))

    def set_sampling(self, rpn_batch: int = 256, rpn_positive_fraction: float = 0.5, roi_batch: int = 1024, roi_positive_fraction: float = 0.25) -> None:
        _check(self.lib, self.lib.rs_trainer_set_sampling(self._h, rpn_batch, rpn_positive_fraction, roi_batch, roi_positive_fraction), "rs_trainer_set_sampling")

    def set_rpn_topk(self, pre_nms_topk_train: int = 2000, post_nms_topk_train: int = 1000) -> None:
        _check(self.lib, self.lib.rs_trainer_set_rpn_topk(self._h, pre_nms_topk_train, post_nms_topk_train), "rs_trainer_set_rpn_topk")

    def write_tensor(self, name: str, data: np.ndarray) -> None:
        """Overwrite a whole trainer tensor (no halo handling)."""
        ptr, dt, shape, _ = self._tensor_ptr(name)
        a = np.ascontiguousarray(data.astype(dt))
        assert a.shape == shape, (a.shape, shape)
        _check(self.lib, self.lib.rs_memcpy_h2d(C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes), "rs_memcpy_h2d")

    def backward_trunk(self, n: int) -> None:
        _check(self.lib, self.lib.rs_trainer_backward_trunk(self._h, n), "rs_trainer_backward_trunk")

    def apply_sgd(self, lr: float, momentum: float = 0.9, weight_decay: float = 1e-4) -> None:
        _check(self.lib, self.lib.rs_trainer_apply_sgd(self._h, lr, momentum, weight_decay), "rs_trainer_apply_sgd")

    def overflowed(self) -> bool:
        """True if the last ``apply_sgd`` found inf / nan in the gradient and therefore skipped the step (fp16 loss scale too
        large for that batch).  Synchronises the trainer's stream."""
        return bool(int(self.tensor("grad_overflow")[0]))

    def set_loss_scale(self, loss_scale: float) -> None:
        """fp16 loss scale of the following steps; call it after ``apply_sgd`` (the gradient buffer carries the scale it was
        computed with)."""
        _check(self.lib, self.lib.rs_trainer_set_loss_scale(self._h, float(loss_scale)), "rs_trainer_set_loss_scale")
        self.loss_scale = float(loss_scale)

    def sync(self) -> None:
        _check(self.lib, self.lib.rs_trainer_sync(self._h), "rs_trainer_sync")

    @property
    def param_count(self) -> int:
        return int(self.lib.rs_trainer_param_count(self._h))

    def close(self) -> None:
        if getattr(self, "_infer", None) is not None:
            self._infer.close()
            self._infer = None
        if getattr(self, "_h", None):
            self.lib.rs_trainer_destroy(self._h)
            self._h = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass


class MultiScaleTrainer:
    """``INPUT.MIN_SIZE_TRAIN`` with ``MIN_SIZE_TRAIN_SAMPLING: choice`` (R:31-38): one ``Trainer`` per shortest-edge size,
    built lazily.  detectron2 draws the size per image and pads the batch to its largest image: ``select_batch(sizes)`` picks
    the trainer of the largest drawn size (the canvas) and hands it the per-image sizes (``Trainer.set_image_sizes``); with one
    image per GPU that is one size per batch.  The optimiser state follows the batch from trainer to trainer
    (``rs_trainer_copy_state``)."""

    def __init__(self, spec: EngineSpec, weights: Dict[str, np.ndarray], tile_shape: Tuple[int, int, int], sizes: Sequence[int], batch: int = 1,
                 device: int = 0, loss_scale: float = 1024.0):
        self.spec, self.weights, self.tile_shape, self.batch, self.device, self.loss_scale = spec, weights, tile_shape, batch, device, loss_scale
        self.sizes = [int(s) for s in sizes]
        self._t: Dict[int, Trainer] = {}
        self.current: Optional[Trainer] = None
        self._sampling: Optional[Tuple[int, float, int, float]] = None

    def set_sampling(self, *a) -> None:
        self._sampling = tuple(a)
        for t in self._t.values():
            t.set_sampling(*a)

    def set_loss_scale(self, loss_scale: float) -> None:
        self.loss_scale = float(loss_scale)
        for t in self._t.values():
            t.set_loss_scale(loss_scale)

    def set_rpn_topk(self, pre: int, post: int) -> None:
        """RPN.PRE_NMS_TOPK_TRAIN / POST_NMS_TOPK_TRAIN (R:config/detectron2_config_3bands.yaml:248-250) of every trainer."""
        self._rpn_topk = (int(pre), int(post))
        for t in self._t.values():
            t.set_rpn_topk(*self._rpn_topk)

    def select(self, size: int) -> Trainer:
        """The trainer for shortest-edge ``size``, holding the up-to-date optimiser state."""
        if size not in self._t:
            t = Trainer(self.spec.replace(min_size_test=int(size)), self.weights, self.tile_shape, self.batch, self.device, self.loss_scale)
            if self._sampling:
                t.set_sampling(*self._sampling)
            if getattr(self, "_rpn_topk", None):
                t.set_rpn_topk(*self._rpn_topk)
            self._t[size] = t
        t = self._t[size]
        if self.current is not None and self.current is not t:
            t.copy_state_from(self.current)
        if getattr(t, "_image_sizes", None):
            t.set_image_sizes(None)              # one size for the whole batch unless select_batch says otherwise
        self.current = t
        return t

    def select_batch(self, sizes: Sequence[int]) -> Trainer:
        """The trainer for a batch whose image i was drawn at shortest edge ``sizes[i]``."""
        t = self.select(max(int(s) for s in sizes))
        t.set_image_sizes(list(sizes) if len(set(int(s) for s in sizes)) > 1 else None)
        return t

    def net_shape(self, size: int) -> Tuple[int, int]:
        from .spec import resize_shortest_edge_shape
        return resize_shortest_edge_shape(self.tile_shape[0], self.tile_shape[1], int(size), self.spec.max_size_test)

    def close(self) -> None:
        for t in self._t.values():
            t.close()
        self._t.clear()
