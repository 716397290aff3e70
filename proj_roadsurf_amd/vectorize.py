"""Detections -> vector features (host side).

What the object-detector's ``detectron2dets_to_features`` does after the predictor returns
([EXT od: helpers/detectron2.py], driven by R:config/config_obj_detec.yaml:87-89): each instance mask is
polygonised with ``rasterio.features.shapes`` (keep value 1), simplified with Ramer-Douglas-Peucker
(``rdp`` package, epsilon 0.75) and emitted as a feature ``{score, det_class, geometry}``.

rasterio, rdp and geopandas are not available offline; both algorithms are implemented in C++ in librs_engine.so
(csrc/vectorize.cpp, ``rs_vectorize_masks``): GDAL-polygonize semantics -- 4-connected regions of a binary mask, rings along
pixel edges (vertices only where the direction changes), exterior ring first, holes after it -- and the classic
Douglas-Peucker with perpendicular distance to the chord.  Multi-threaded over instances, straight on the bit-packed masks
the engine returns.  This module is the ctypes side only; the pure-Python restatement the C++ is checked against vertex for
vertex lives in oracle/host_tail_oracle.py (tests only).

PARITY UNPINNED against rasterio/rdp (absent); the unit tests pin areas, hole counts, documented conventions and invariants.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

Ring = List[Tuple[float, float]]
Polygon = List[Ring]                     # [exterior, hole, hole, ...], each ring closed (first == last)


def vectorize_masks_native(packed: np.ndarray, h: int, w: int, rdp_epsilon: float = 0.0, threads: int = 0) -> List[List[Polygon]]:
    """Polygons of n bit-packed masks ``packed`` (n, h, ceil(w/8)) uint8 (LSB first, the engine's layout) through
    ``rs_vectorize_masks``: per instance a list of polygons, each a list of closed rings of (x, y) tuples --
    exactly ``[[rdp(r) ...] for poly in mask_to_polygons(mask)]`` of oracle/host_tail_oracle.py.  Raises if librs_engine.so is missing."""
    import ctypes as C
    from .engine import load_library, RsError
    lib = load_library()
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    n = packed.shape[0]
    if packed.shape[1:] != (h, (w + 7) // 8):
        raise ValueError(f"packed masks must be (n,{h},{(w + 7) // 8}), got {packed.shape}")
    if n == 0:
        return []
    r = lib.rs_vectorize_masks(packed.ctypes.data_as(C.c_void_p), n, h, w, float(rdp_epsilon), int(threads))
    if not r:
        raise RsError("rs_vectorize_masks failed")
    try:
        c = [C.c_int64() for _ in range(4)]
        lib.rs_vec_counts(r, *[C.byref(x) for x in c])
        ni, npoly, nr, nv = (int(x.value) for x in c)
        ipc = np.zeros(ni, np.int32); prc = np.zeros(npoly, np.int32); rl = np.zeros(nr, np.int32); xy = np.zeros((nv, 2), np.float64)
        lib.rs_vec_copy(r, ipc.ctypes.data_as(C.POINTER(C.c_int32)), prc.ctypes.data_as(C.POINTER(C.c_int32)),
                        rl.ctypes.data_as(C.POINTER(C.c_int32)), xy.ctypes.data_as(C.POINTER(C.c_double)))
    finally:
        lib.rs_vec_free(r)
    out: List[List[Polygon]] = []
    pts = xy.tolist()
    pi = ri = vi = 0
    for i in range(ni):
        polys: List[Polygon] = []
        for _ in range(int(ipc[i])):
            rings: Polygon = []
            for _ in range(int(prc[pi])):
                k = int(rl[ri])
                rings.append([(p[0], p[1]) for p in pts[vi:vi + k]])
                vi += k
                ri += 1
            polys.append(rings)
            pi += 1
        out.append(polys)
    return out


def instances_to_gpkg_rows(instances, image_name: str, extent: Optional[Sequence[float]] = None, rdp_enabled: bool = True,
                           rdp_epsilon: float = 0.75, srs_id: int = -1, threads: int = 0):
    """The fast path of the CLI: masks -> polygons -> RDP -> georeferenced GeoPackage geometry blobs, all in C++
    (``rs_vectorize_masks`` + ``rs_vec_gpkg_blobs``); Python only pairs each blob with its instance's score and class.
    Returns (rows [(blob, score, det_class, image)], bbox [minx, miny, maxx, maxy] or None).  Same polygons, vertices and bytes as
    ``gpkg.gpkg_geom`` applied to ``instances_to_features`` (tests/test_vector_cli.py)."""
    import ctypes as C
    from .engine import load_library, RsError
    n = len(instances)
    if n == 0 or not instances.has("pred_masks"):
        return [], None
    lib = load_library()
    lib.rs_vec_gpkg_blobs.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.rs_vec_gpkg_blobs.restype = C.c_int64
    h, w = instances.image_size
    crops = getattr(instances, "_crops", None)
    if crops is not None:
        # masks arrived as crops of their boxes (engine.Engine.fetch_wait, rs_mask_crops): trace inside the crops, same vertices
        rects, offs_c, data = crops
        lib.rs_vectorize_mask_crops.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        lib.rs_vectorize_mask_crops.restype = C.c_void_p
        rects = np.ascontiguousarray(rects, np.int32); offs_c = np.ascontiguousarray(offs_c, np.uint32); data = np.ascontiguousarray(data, np.uint8)
        r = lib.rs_vectorize_mask_crops(data.ctypes.data_as(C.c_void_p), rects.ctypes.data_as(C.c_void_p), offs_c.ctypes.data_as(C.c_void_p),
                                        n, h, w, float(rdp_epsilon if rdp_enabled else 0.0), int(threads))
    else:
        packed = getattr(instances, "_packed", None)
        if packed is None:
            packed = np.packbits(np.asarray(instances.pred_masks, dtype=bool), axis=2, bitorder="little")
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        r = lib.rs_vectorize_masks(packed.ctypes.data_as(C.c_void_p), n, h, w, float(rdp_epsilon if rdp_enabled else 0.0), int(threads))
    if not r:
        raise RsError("rs_vectorize_masks failed")
    try:
        c = [C.c_int64() for _ in range(4)]
        lib.rs_vec_counts(r, *[C.byref(x) for x in c])
        ni, npoly = int(c[0].value), int(c[1].value)
        if npoly == 0:
            return [], None
        ipc = np.zeros(ni, np.int32)
        lib.rs_vec_copy(r, ipc.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None)
        xform = None
        if extent is not None:
            xmin, ymin, xmax, ymax = (float(v) for v in extent)
            xform = np.ascontiguousarray(np.tile(np.array([xmin, ymax, (xmax - xmin) / w, (ymax - ymin) / h], np.float64), (ni, 1)))
        xp = xform.ctypes.data_as(C.c_void_p) if xform is not None else None
        need = int(lib.rs_vec_gpkg_blobs(r, xp, int(srs_id), None, 0, None, None))
        buf = np.empty(need, np.uint8)
        offs = np.zeros(npoly + 1, np.int64)
        bbox = np.zeros(4, np.float64)
        got = int(lib.rs_vec_gpkg_blobs(r, xp, int(srs_id), buf.ctypes.data_as(C.c_void_p), need, offs.ctypes.data_as(C.c_void_p), bbox.ctypes.data_as(C.c_void_p)))
        if got != need:
            raise RsError("rs_vec_gpkg_blobs failed")
    finally:
        lib.rs_vec_free(r)
    raw = buf.tobytes()
    o = offs.tolist()
    scores = [float(s) for s in instances.scores]
    classes = [int(k) for k in instances.pred_classes]
    inst_of = np.repeat(np.arange(ni), ipc).tolist()
    rows = [(raw[o[i]:o[i + 1]], scores[inst_of[i]], classes[inst_of[i]], image_name) for i in range(npoly)]
    return rows, bbox.tolist()


def instances_to_features(instances, image_name: str, extent: Optional[Sequence[float]] = None,
                          rdp_enabled: bool = True, rdp_epsilon: float = 0.75, threads: int = 0) -> List[dict]:
    """GeoJSON-like features, one per polygon, with the columns the reference's post-stage reads
    (``score``, ``det_class``, ``geometry`` -- R:scripts/road_segmentation/determine_class.py:22-25,113).
    ``extent`` = (xmin, ymin, xmax, ymax) of the tile in its CRS; None keeps pixel coordinates.
    The RDP tolerance is applied in pixel units, before georeferencing (documented assumption).
    Polygonisation and RDP run in C++ (``rs_vectorize_masks``); instances without masks give their boxes."""
    h, w = instances.image_size
    feats: List[dict] = []
    has_masks = instances.has("pred_masks")
    native_polys = None
    if has_masks and len(instances):
        packed = getattr(instances, "_packed", None)
        if packed is None:          # a detectron2-style Instances with bool masks: pack them the way the engine does
            packed = np.packbits(np.asarray(instances.pred_masks, dtype=bool), axis=2, bitorder="little")
        native_polys = vectorize_masks_native(packed, h, w, rdp_epsilon if rdp_enabled else 0.0, threads)
    for i in range(len(instances)):
        if native_polys is None:
            x1, y1, x2, y2 = [float(v) for v in instances.pred_boxes[i]]
            polys: List[Polygon] = [[[(x1, y1), (x2, y1), (x2, y2), (x1, y2), (x1, y1)]]]
        else:
            polys = native_polys[i]
        for poly in polys:
            rings = []
            for r in poly:
                rr = list(r)
                if extent is not None:
                    xmin, ymin, xmax, ymax = extent
                    sx, sy = (xmax - xmin) / w, (ymax - ymin) / h
                    rr = [(xmin + x * sx, ymax - y * sy) for x, y in rr]
                rings.append([[float(x), float(y)] for x, y in rr])
            feats.append({"type": "Feature",
                          "geometry": {"type": "Polygon", "coordinates": rings},
                          "properties": {"score": float(instances.scores[i]), "det_class": int(instances.pred_classes[i]),
                                         "image": image_name}})
    return feats
