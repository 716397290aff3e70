"""Detections -> vector features (host side).

What the object-detector's ``detectron2dets_to_features`` does after the predictor returns
([EXT od: helpers/detectron2.py], driven by R:config/config_obj_detec.yaml:87-89): each instance mask is
polygonised with ``rasterio.features.shapes`` (keep value 1), simplified with Ramer-Douglas-Peucker
(``rdp`` package, epsilon 0.75) and emitted as a feature ``{score, det_class, geometry}``.

rasterio, rdp and geopandas are not available offline, so this module restates the two algorithms:

* ``mask_to_polygons``: GDAL-polygonize semantics -- 4-connected regions of a binary mask, rings along pixel
  edges (vertices only where the direction changes), exterior ring first, holes after it;
* ``rdp``: the classic recursive Douglas-Peucker with perpendicular distance to the chord (distance to the
  start point when the chord is degenerate, as the ``rdp`` package does for closed rings).

PARITY UNPINNED against rasterio/rdp (absent); the unit tests pin areas, hole counts and invariants.

The CLI does not run the Python loops (0.7 s per 100-instance tile): ``vectorize_masks_native`` calls the C++
form in librs_engine.so (csrc/vectorize.cpp, ``rs_vectorize_masks``), multi-threaded over instances, straight on
the bit-packed masks the engine returns; tests/test_vector_cli.py checks it against the functions below vertex for
vertex.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

Ring = List[Tuple[float, float]]
Polygon = List[Ring]                     # [exterior, hole, hole, ...], each ring closed (first == last)


def _trace_rings(mask: np.ndarray) -> List[Ring]:
    """All boundary rings of the foreground of a binary mask, along pixel edges, in pixel-corner coordinates
    (x = column, y = row).  Edges are directed with the foreground on their RIGHT (clockwise in image
    coordinates, y down, for exteriors; counter-clockwise for holes).  At a corner where two foreground pixels
    touch only diagonally the ring turns RIGHT, which keeps 4-connected regions separate."""
    h, w = mask.shape
    m = np.zeros((h + 2, w + 2), bool)
    m[1:-1, 1:-1] = mask.astype(bool)
    # directed edges keyed by start vertex: direction index 0:E 1:S 2:W 3:N
    DX = (1, 0, -1, 0)
    DY = (0, 1, 0, -1)
    out: Dict[Tuple[int, int], List[int]] = {}

    def add(x: int, y: int, d: int) -> None:
        out.setdefault((x, y), []).append(d)

    ys, xs = np.nonzero(m)
    for y, x in zip(ys.tolist(), xs.tolist()):
        px, py = x - 1, y - 1              # pixel coordinates in the original mask
        if not m[y - 1, x]:                # top side: go east along the top edge (foreground below = right)
            add(px, py, 0)
        if not m[y, x + 1]:                # right side: go south
            add(px + 1, py, 1)
        if not m[y + 1, x]:                # bottom side: go west
            add(px + 1, py + 1, 2)
        if not m[y, x - 1]:                # left side: go north
            add(px, py + 1, 3)
    rings: List[Ring] = []
    while out:
        v0 = next(iter(out))
        d0 = out[v0][0]
        x, y = v0
        cur_d = d0
        ring: Ring = [(float(x), float(y))]
        while True:
            x, y = x + DX[cur_d], y + DY[cur_d]
            nxt = out[(x, y)]
            # prefer the right turn, then straight, then left (the right turn separates diagonal neighbours)
            choice = next(c for c in ((cur_d + 1) % 4, cur_d, (cur_d + 3) % 4) if c in nxt)
            closing = (x, y) == v0 and choice == d0
            nxt.remove(choice)
            if not nxt:
                del out[(x, y)]
            if closing:
                if choice == cur_d:          # the walk started in the middle of a straight run
                    ring = ring[1:]
                break
            if choice != cur_d:
                ring.append((float(x), float(y)))
            cur_d = choice
        ring.append(ring[0])
        rings.append(ring)
    return rings


def ring_area(ring: Sequence[Tuple[float, float]]) -> float:
    """Signed shoelace area (positive = clockwise in image coordinates with y down)."""
    a = 0.0
    for (x0, y0), (x1, y1) in zip(ring[:-1], ring[1:]):
        a += x0 * y1 - x1 * y0
    return a / 2.0


def _point_in_ring(pt: Tuple[float, float], ring: Sequence[Tuple[float, float]]) -> bool:
    x, y = pt
    inside = False
    for (x0, y0), (x1, y1) in zip(ring[:-1], ring[1:]):
        if (y0 > y) != (y1 > y):
            xi = x0 + (y - y0) * (x1 - x0) / (y1 - y0)
            if xi > x:
                inside = not inside
    return inside


def mask_to_polygons(mask: np.ndarray) -> List[Polygon]:
    """Polygons (pixel-corner coordinates) of the 4-connected foreground regions of ``mask``."""
    rings = _trace_rings(np.asarray(mask))
    ext = [r for r in rings if ring_area(r) > 0]
    holes = [r for r in rings if ring_area(r) < 0]
    polys: List[Polygon] = [[r] for r in ext]
    for hr in holes:
        # a point strictly inside the hole next to its first edge: take the edge midpoint shifted into the hole
        (x0, y0), (x1, y1) = hr[0], hr[1]
        mx, my = (x0 + x1) / 2.0, (y0 + y1) / 2.0
        dx, dy = x1 - x0, y1 - y0
        n = max(abs(dx), abs(dy))
        # hole rings run counter-clockwise with the foreground on the right => the hole interior is on the left
        px, py = mx + 0.5 * (dy / n), my - 0.5 * (dx / n)
        best, best_area = None, None
        for i, p in enumerate(polys):
            if _point_in_ring((px, py), p[0]):
                a = ring_area(p[0])
                if best is None or a < best_area:
                    best, best_area = i, a
        if best is not None:
            polys[best].append(hr)
    return polys


def rdp(points: Sequence[Tuple[float, float]], epsilon: float) -> List[Tuple[float, float]]:
    """Ramer-Douglas-Peucker (iterative form of the recursion used by the ``rdp`` package)."""
    pts = np.asarray(points, dtype=np.float64)
    n = len(pts)
    if n < 3 or epsilon <= 0:
        return [tuple(p) for p in pts.tolist()]
    keep = np.zeros(n, bool)
    keep[0] = keep[-1] = True
    stack = [(0, n - 1)]
    while stack:
        i0, i1 = stack.pop()
        if i1 <= i0 + 1:
            continue
        a, b = pts[i0], pts[i1]
        seg = b - a
        mid = pts[i0 + 1:i1]
        if np.allclose(seg, 0):
            d = np.linalg.norm(mid - a, axis=1)
        else:
            d = np.abs(seg[0] * (mid[:, 1] - a[1]) - seg[1] * (mid[:, 0] - a[0])) / np.linalg.norm(seg)
        k = int(np.argmax(d))
        if d[k] > epsilon:
            idx = i0 + 1 + k
            keep[idx] = True
            stack.append((i0, idx))
            stack.append((idx, i1))
    return [tuple(p) for p in pts[keep].tolist()]


def vectorize_masks_native(packed: np.ndarray, h: int, w: int, rdp_epsilon: float = 0.0, threads: int = 0) -> List[List[Polygon]]:
    """Polygons of n bit-packed masks ``packed`` (n, h, ceil(w/8)) uint8 (LSB first, the engine's layout) through
    ``rs_vectorize_masks``: per instance a list of polygons, each a list of closed rings of (x, y) tuples --
    exactly ``[[rdp(r) ...] for poly in mask_to_polygons(mask)]``.  Raises if librs_engine.so is missing."""
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
                          rdp_enabled: bool = True, rdp_epsilon: float = 0.75, native: bool = True, threads: int = 0) -> List[dict]:
    """GeoJSON-like features, one per polygon, with the columns the reference's post-stage reads
    (``score``, ``det_class``, ``geometry`` -- R:scripts/road_segmentation/determine_class.py:22-25,113).
    ``extent`` = (xmin, ymin, xmax, ymax) of the tile in its CRS; None keeps pixel coordinates.
    The RDP tolerance is applied in pixel units, before georeferencing (documented assumption)."""
    h, w = instances.image_size
    feats: List[dict] = []
    has_masks = instances.has("pred_masks")
    native_polys = None
    if has_masks and native and len(instances):
        # C++ path on the bit-packed masks (RDP included); ``native=False`` runs the Python restatement below
        packed = getattr(instances, "_packed", None)
        if packed is None:          # a detectron2-style Instances with bool masks: pack them the way the engine does
            packed = np.packbits(np.asarray(instances.pred_masks, dtype=bool), axis=2, bitorder="little")
        native_polys = vectorize_masks_native(packed, h, w, rdp_epsilon if rdp_enabled else 0.0, threads)
    masks = instances.pred_masks if (has_masks and native_polys is None) else None
    for i in range(len(instances)):
        if not has_masks:
            x1, y1, x2, y2 = [float(v) for v in instances.pred_boxes[i]]
            polys: List[Polygon] = [[[(x1, y1), (x2, y1), (x2, y2), (x1, y2), (x1, y1)]]]
        elif native_polys is not None:
            polys = native_polys[i]
        else:
            polys = mask_to_polygons(masks[i])
        for poly in polys:
            rings = []
            for r in poly:
                if native_polys is not None or not has_masks:
                    rr = list(r)
                else:
                    rr = rdp(r, rdp_epsilon) if rdp_enabled else list(r)
                    if len(rr) < 4:
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
