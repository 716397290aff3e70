"""Host-side training targets: ground-truth masks of the sampled foreground RoIs, as
``PolygonMasks.crop_and_resize`` produces them for ``mask_rcnn_loss`` ([EXT d2: structures/masks.py,
modeling/roi_heads/mask_head.py]; INPUT.MASK_FORMAT polygon, R:config/detectron2_config_3bands.yaml:27).  detectron2 does this
step on the host as well (pycocotools); the rasteriser here is the C++ restatement in librs_engine.so."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np


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
