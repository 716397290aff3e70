"""Host-side training targets: ground-truth masks of the sampled foreground RoIs, as
``PolygonMasks.crop_and_resize`` produces them for ``mask_rcnn_loss`` ([EXT d2: structures/masks.py,
modeling/roi_heads/mask_head.py]; INPUT.MASK_FORMAT polygon, R:config/detectron2_config_3bands.yaml:27).  detectron2 does this
step on the host as well (pycocotools); the rasteriser here is the C++ restatement in librs_engine.so."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np


def rasterize_entries(instances: Sequence[Sequence[np.ndarray]], entry_instance: np.ndarray, boxes: np.ndarray, mask_size: int,
                      threads: int = 0) -> np.ndarray:
    """All mask targets of a step in one native call: instances[g] = polygons of gt instance g, entry e = instance
    entry_instance[e] cropped to boxes[e] (float32 x1,y1,x2,y2).  Returns (n_entries, mask_size, mask_size) bool."""
    from .engine import load_library, RsError
    lib = load_library()
    ne = int(len(entry_instance))
    out = np.zeros((ne, mask_size, mask_size), np.uint8)
    if ne == 0:
        return out.astype(bool)
    arrs = [np.asarray(p, np.float64).reshape(-1) for polys in instances for p in polys]
    lens = np.array([a.size for a in arrs], np.int32)
    off = np.zeros(len(arrs), np.int64)
    if len(arrs) > 1:
        off[1:] = np.cumsum(lens[:-1], dtype=np.int64)
    flat = np.ascontiguousarray(np.concatenate(arrs)) if arrs else np.zeros(1)
    first = np.zeros(len(instances) + 1, np.int32)
    first[1:] = np.cumsum([len(polys) for polys in instances])
    ei = np.ascontiguousarray(np.asarray(entry_instance, np.int32))
    bx = np.ascontiguousarray(np.asarray(boxes, np.float32).reshape(ne, 4))
    lib.rs_rasterize_entries.restype = C.c_int
    lib.rs_rasterize_entries.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    rc = lib.rs_rasterize_entries(flat.ctypes.data, off.ctypes.data, lens.ctypes.data, first.ctypes.data, len(instances), ei.ctypes.data,
                                  bx.ctypes.data, ne, mask_size, out.ctypes.data, threads)
    if rc != 0:
        raise RsError(f"rs_rasterize_entries failed ({rc})")
    return out.astype(bool)


def rasterize_polygons_within_box(polygons: Sequence[np.ndarray], box: np.ndarray, mask_size: int) -> np.ndarray:
    """Polygons ([x0,y0,x1,y1,...] each, image coordinates) of one instance -> (mask_size, mask_size) bool target inside ``box``."""
    from .engine import load_library, RsError
    lib = load_library()
    flat = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float64).reshape(-1) for p in polygons]) if len(polygons) else np.zeros(0))
    lens = np.ascontiguousarray(np.array([np.asarray(p).size for p in polygons], np.int32))
    b = np.ascontiguousarray(np.asarray(box, np.float64).reshape(4))
    out = np.zeros((mask_size, mask_size), np.uint8)
    rc = lib.rs_rasterize_polygons_within_box(flat.ctypes.data_as(C.POINTER(C.c_double)), lens.ctypes.data_as(C.POINTER(C.c_int32)), len(polygons),
                                              b.ctypes.data_as(C.POINTER(C.c_double)), mask_size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc != 0:
        raise RsError(f"rs_rasterize_polygons_within_box failed ({rc})")
    return out.astype(bool)
