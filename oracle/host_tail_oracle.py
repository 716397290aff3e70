"""ORACLE (test infrastructure only) -- pure-Python / numpy restatements of the host tail of the detector step, the checkers of
``csrc/vectorize.cpp`` (``rs_vectorize_masks`` / ``rs_vectorize_mask_crops``) and ``csrc/raster_vote.hip``
(``rs_op_mask_overlap``).  Moved out of the product package in round 3: only ``tests/`` may import this module.

What it restates: the STDL object-detector's ``detectron2dets_to_features``
([EXT od: helpers/detectron2.py], driven by R:config/config_obj_detec.yaml:87-89) polygonises each instance mask with
``rasterio.features.shapes`` (value 1 kept) and simplifies the rings with Ramer-Douglas-Peucker (``rdp`` package, epsilon 0.75).

* ``mask_to_polygons``: GDAL-polygonize semantics -- 4-connected regions of a binary mask (rasterio's default ``connectivity=4``),
  rings along pixel edges (vertices only where the direction changes), exterior ring first, holes after it, each hole assigned to
  the smallest exterior that contains it;
* ``rdp``: the classic recursive Douglas-Peucker with perpendicular distance to the chord (distance to the start point when the
  chord is degenerate, as the ``rdp`` package does for closed rings);
* ``overlap_counts``: |label raster AND detection mask| per pair, the numpy statement of the popcount kernel.

PARITY UNPINNED against rasterio / GDAL / rdp (absent from /root/reference and from this image): the tests pin areas,
component and hole counts, the documented conventions listed in tests/test_vector_cli.py, and invariants.
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
    """Signed shoelace area (positive = clockwise on the screen, y down: the tracer's exteriors; ``mask_to_polygons`` RETURNS its
    rings reversed, rasterio's direction, so its exteriors read negative and its holes positive)."""
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
    # Ring direction as rasterio.features.shapes emits it (its documentation, topics/features: the single pixel at column 71, row 6
    # comes out as [(71,6), (71,7), (72,7), (72,6), (71,6)]): from the start vertex DOWN first, counter-clockwise on the screen for
    # exteriors, holes the other way round.  ``_trace_rings`` walks the other way; reversing a closed ring keeps its start vertex.
    return [[list(reversed(r)) for r in poly] for poly in polys]


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


def instances_to_features(instances, image_name: str, extent: Optional[Sequence[float]] = None,
                          rdp_enabled: bool = True, rdp_epsilon: float = 0.75) -> List[dict]:
    """GeoJSON-like features, one per polygon (``score``, ``det_class``, ``geometry`` --
    R:scripts/road_segmentation/determine_class.py:22-25,113), through the Python restatements above: what
    ``proj_roadsurf_amd.vectorize.instances_to_features`` (C++ path) must reproduce vertex for vertex."""
    h, w = instances.image_size
    feats: List[dict] = []
    masks = instances.pred_masks
    for i in range(len(instances)):
        for poly in mask_to_polygons(masks[i]):
            rings = []
            for r in poly:
                rr = rdp(r, rdp_epsilon) if rdp_enabled else list(r)
                if len(rr) < 4:
                    rr = list(r)
                if extent is not None:
                    xmin, ymin, xmax, ymax = extent
                    sx, sy = (xmax - xmin) / w, (ymax - ymin) / h
                    rr = [(xmin + x * sx, ymax - y * sy) for x, y in rr]
                rings.append([[float(x), float(y)] for x, y in rr])
            feats.append({"type": "Feature", "geometry": {"type": "Polygon", "coordinates": rings},
                          "properties": {"score": float(instances.scores[i]), "det_class": int(instances.pred_classes[i]),
                                         "image": image_name}})
    return feats


def overlap_counts(det_packed: np.ndarray, lab_packed: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """numpy statement of ``rs_op_mask_overlap`` (csrc/raster_vote.hip): (inter [n_lab][n_det], label_area [n_lab]) in pixels --
    the intersection areas ``determine_class.get_weighted_scores`` takes from a polygon overlay
    (R:scripts/road_segmentation/determine_class.py:97-120), counted on the tile grid."""
    d = np.unpackbits(det_packed.reshape(det_packed.shape[0], -1), axis=1).astype(np.int64)
    l = np.unpackbits(lab_packed.reshape(lab_packed.shape[0], -1), axis=1).astype(np.int64)
    return (l @ d.T).astype(np.int32), l.sum(1).astype(np.int32)
