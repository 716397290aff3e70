"""Raster voting on the tile grid: the class vote of the reference's post-stage computed from the detection
MASKS instead of from vectorised polygons.

Reference (vector form, geopandas/shapely): ``determine_class.get_weighted_scores`` overlays the tile-clipped road labels with the
detection polygons, weights every (road, detection) intersection by ``round(area(intersection) / area(label), 2)``, keeps pairs
above 0.05, and ``determine_detected_class`` sums, per road and class, weighted scores over weights
(R:scripts/road_segmentation/determine_class.py:97-120, :122-190).  Here the intersection areas are pixel counts:
``|label_raster AND detection_mask|`` on the device (``rs_op_mask_overlap`` / ``rs_engine_label_overlap``: popcounts of bit-packed
rows, the detection masks never leave HBM), everything after the counts is restated line by line on the host.

Differences from the vector form, by construction: areas are counted in whole pixels of the tile grid (0.4 m at z18,
R:config/config_obj_detec.yaml:20,45), and RDP simplification (ε 0.75 px) does not enter.  The numpy statement of the device kernel (the
oracle of its test) is ``oracle/host_tail_oracle.py::overlap_counts``."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np

CLASS_NAMES = {0: "artificial", 1: "natural"}          # R:scripts/road_segmentation/determine_class.py:19-28 (det_class -> name)


def label_rasters(label_polygons: Sequence[Sequence[np.ndarray]], h: int, w: int) -> np.ndarray:
    """Bit-packed (n_labels, h, ceil(w/8)) rasters of road labels given as polygons in TILE PIXEL coordinates (one list of
    [x0, y0, x1, y1, ...] rings per label), pixel centres inside = 1 (the rasteriser of the training targets, square tiles)."""
    from .train_targets import rasterize_polygons_within_box
    if h != w:
        raise ValueError("label rasters are built for square tiles")
    out = np.zeros((len(label_polygons), h, (w + 7) // 8), np.uint8)
    for i, polys in enumerate(label_polygons):
        m = rasterize_polygons_within_box(polys, np.array([0.0, 0.0, float(w), float(h)]), h)
        out[i] = np.packbits(m, axis=1, bitorder="little")
    return out


def overlap_counts_device(lib, det_dev_ptr: int, n_det: int, lab_packed: np.ndarray, h: int, w: int) -> Tuple[np.ndarray, np.ndarray]:
    """Counts for detection masks ALREADY on the device (``Engine.tensor_ptr("masks")`` + tile offset): uploads the label rasters,
    runs the kernel, reads the two small count arrays back."""
    import torch
    lab = torch.from_numpy(np.ascontiguousarray(lab_packed)).cuda()
    inter = torch.zeros((lab_packed.shape[0], n_det), dtype=torch.int32, device="cuda")
    area = torch.zeros((lab_packed.shape[0],), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    lib.rs_op_mask_overlap.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib.rs_op_mask_overlap(C.c_void_p(det_dev_ptr), n_det, C.c_void_p(lab.data_ptr()), lab_packed.shape[0], h, w,
                                C.c_void_p(inter.data_ptr()), C.c_void_p(area.data_ptr()), None)
    if rc != 0:
        raise RuntimeError(f"rs_op_mask_overlap failed ({rc}): {lib.rs_last_error().decode(errors='replace')}")
    torch.cuda.synchronize()
    return inter.cpu().numpy(), area.cpu().numpy()


def weighted_scores(inter: np.ndarray, label_area: np.ndarray, scores: np.ndarray, classes: np.ndarray, road_ids: Sequence) -> List[Dict]:
    """``get_weighted_scores`` (R:determine_class.py:97-120) from pixel counts: one row per (label, detection) pair with
    ``area_pred_in_label = round(intersection / label area, 2) > 0.05``; ``weighted_score = area_pred_in_label * score``."""
    rows = []
    for li in range(inter.shape[0]):
        if label_area[li] <= 0:
            continue
        for di in range(inter.shape[1]):
            if inter[li, di] <= 0:
                continue
            frac = float(np.round(inter[li, di] / label_area[li], 2))
            if frac > 0.05:
                rows.append({"OBJECTID": road_ids[li], "det": di, "score": float(scores[di]), "det_class": int(classes[di]),
                             "det_class_name": CLASS_NAMES.get(int(classes[di]), str(int(classes[di]))),
                             "area_pred_in_label": frac, "weighted_score": frac * float(scores[di])})
    return rows


def determine_detected_class(rows: Sequence[Dict], road_ids: Sequence, threshold: float = 0.0) -> List[Dict]:
    """``determine_detected_class`` (R:determine_class.py:122-190): per road, the class whose score-weighted mean
    (sum of weighted scores / sum of weights over the road's pairs with score >= threshold) is larger; "undetected" without
    pairs, "undetermined" on a tie; scores rounded to 3 decimals, ``diff_score`` their absolute difference."""
    valid = [r for r in rows if r["score"] >= threshold]
    out = []
    seen = []
    for rid in road_ids:
        if rid in seen:
            continue
        seen.append(rid)
        mine = [r for r in valid if r["OBJECTID"] == rid]
        if not mine:
            out.append({"road_id": rid, "cover_type": "undetected", "nat_score": 0, "art_score": 0, "diff_score": 0})
            continue
        idx = {}
        for name in ("natural", "artificial"):
            ws = sum(r["weighted_score"] for r in mine if r["det_class_name"] == name)
            wa = sum(r["area_pred_in_label"] for r in mine if r["det_class_name"] == name)
            idx[name] = 0 if ws == 0 else ws / wa
        nat, art = idx["natural"], idx["artificial"]
        cover = "undetermined" if art == nat else ("artificial" if art > nat else "natural")
        out.append({"road_id": rid, "cover_type": cover, "nat_score": round(nat, 3), "art_score": round(art, 3),
                    "diff_score": 0 if art == nat else abs(art - nat)})
    return out
