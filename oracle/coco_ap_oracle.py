"""TEST INFRASTRUCTURE (only tests/ may import this): an independent brute-force statement of COCO average precision, the checker of
proj_roadsurf_amd/coco_eval.py (the reference evaluates bbox / segm AP every 200 iterations,
R:config/detectron2_config_3bands.yaml:13-16,322, through detectron2's COCOEvaluator -> pycocotools COCOeval).

PARITY UNPINNED: pycocotools is absent offline and the reference holds no AP fixtures, so neither file can be checked against COCOeval itself.  What
this file buys is independence: it states the published definitions directly instead of following COCOeval's evaluateImg / accumulate code path --

  * matching as a declarative choice per detection (best available ground truth by a total preference order) instead of the sequential scan with its
    early `break`;
  * interpolated precision by its definition, p(r) = max { precision_i : recall_i >= r } over the ranked list (0 when no recall reaches r), instead
    of the right-to-left running maximum + searchsorted;
  * plain Python loops over explicit sorted lists of (score, image, rank) records, no cumulative-sum arrays.

Definitions restated (COCO detection evaluation, cocodataset.org "Evaluate" + [EXT coco: PythonAPI/pycocotools/cocoeval.py], numpy conventions):
  - IoU thresholds 0.50:0.05:0.95, recall thresholds 0:0.01:1, area ranges all [0, 1e10], small [0, 32^2], medium [32^2, 96^2], large [96^2, 1e10];
  - per (image, category): detections ranked by score (descending, ties by input order), the first `max_dets` kept;
  - a ground truth is IGNORED in an area range when it is a crowd region or its area lies outside the range; IoU against a crowd region uses the
    detection's area as the union;
  - a detection, in rank order, takes one ground truth of IoU >= min(t, 1 - 1e-10): non-crowd ground truths taken by an earlier detection are
    unavailable (crowd regions stay available); a non-ignored candidate beats every ignored one; otherwise higher IoU wins and, at equal IoU, the
    candidate later in the (non-ignored first, input order) list;
  - a detection matched to an ignored ground truth is ignored; an unmatched detection whose own area lies outside the range is ignored;
  - over all images, the non-ignored detections of a (category, range) ranked by score (ties: image order, then rank within the image): TP if
    matched, FP otherwise; recall_i = TP_i / (number of non-ignored ground truths), precision_i = TP_i / (TP_i + FP_i);
  - AP = mean of p(r) over the recall thresholds, IoU thresholds and categories that have at least one non-ignored ground truth."""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np

# The threshold VALUES are COCOeval's Params: np.linspace(.5, .95, 10) and np.linspace(0, 1, 101).  Several of them differ from 0.5 + 0.05 i and i / 100 in
# the last bit (linspace(0, 1, 101)[29] = 0.29000000000000004), which decides `recall >= r` whenever a recall TP / n lands on a multiple of 0.01 -- the
# first version of this file used i / 100 and disagreed with coco_eval.py on exactly those sets (n = 10, 20, 25, 50 ground truths).
_T = [float(v) for v in np.linspace(0.5, 0.95, 10)]
_R = [float(v) for v in np.linspace(0.0, 1.0, 101)]
_AREAS = {"all": (0.0, 1e10), "small": (0.0, 1024.0), "medium": (1024.0, 9216.0), "large": (9216.0, 1e10)}


def _box_area(b) -> float:
    return float((b[2] - b[0]) * (b[3] - b[1]))


def _iou_boxes(d, g, crowd: bool) -> float:
    iw = min(d[2], g[2]) - max(d[0], g[0])
    ih = min(d[3], g[3]) - max(d[1], g[1])
    if iw <= 0 or ih <= 0:
        return 0.0
    inter = iw * ih
    union = _box_area(d) if crowd else _box_area(d) + _box_area(g) - inter
    return inter / union if union > 0 else 0.0


def _iou_masks(d, g, crowd: bool) -> float:
    inter = float(np.logical_and(d, g).sum())
    ad, ag = float(d.sum()), float(g.sum())
    union = ad if crowd else ad + ag - inter
    return inter / union if union > 0 else 0.0


def average_precision(gts: Sequence[Dict], dets: Sequence[Dict], num_classes: int, iou_type: str = "bbox", max_dets: int = 100) -> Dict[str, float]:
    """Same inputs and result keys as ``coco_eval.evaluate``: AP, AP50, AP75, APs, APm, APl, AP-class<k> in percent (NaN where undefined)."""
    assert iou_type in ("bbox", "segm")
    # p[area][t][category] = list of 101 interpolated precisions, or None when the category has no non-ignored ground truth in that range
    table = {a: [[None] * num_classes for _ in _T] for a in _AREAS}
    for cat in range(num_classes):
        per_image = []
        for img, (g, d) in enumerate(zip(gts, dets)):
            gi = [k for k in range(len(g["classes"])) if int(g["classes"][k]) == cat]
            di = [k for k in range(len(d["classes"])) if int(d["classes"][k]) == cat]
            di.sort(key=lambda k: (-float(d["scores"][k]), k))
            di = di[:max_dets]
            crowd = [bool(g["crowd"][k]) if "crowd" in g else False for k in gi]
            if iou_type == "segm":
                g_area = [float(np.asarray(g["masks"][k]).sum()) for k in gi]
                d_area = [float(np.asarray(d["masks"][k]).sum()) for k in di]
                iou = [[_iou_masks(np.asarray(d["masks"][a], bool), np.asarray(g["masks"][b], bool), crowd[j]) for j, b in enumerate(gi)] for a in di]
            else:
                g_area = [_box_area(g["boxes"][k]) for k in gi]
                d_area = [_box_area(d["boxes"][k]) for k in di]
                iou = [[_iou_boxes(d["boxes"][a], g["boxes"][b], crowd[j]) for j, b in enumerate(gi)] for a in di]
            if "area" in g:
                g_area = [float(g["area"][k]) for k in gi]
            per_image.append({"img": img, "scores": [float(d["scores"][k]) for k in di], "d_area": d_area, "g_area": g_area, "crowd": crowd, "iou": iou})
        for aname, (lo, hi) in _AREAS.items():
            for ti, thr in enumerate(_T):
                ranked = []            # (-score, image, rank in image, is true positive) of the non-ignored detections
                n_gt = 0
                for rec in per_image:
                    ignored_gt = [c or a < lo or a > hi for c, a in zip(rec["crowd"], rec["g_area"])]
                    n_gt += sum(1 for x in ignored_gt if not x)
                    # position of every ground truth in the (non-ignored first, input order) list: the tie rule prefers the later position
                    pos = {j: p for p, j in enumerate(sorted(range(len(ignored_gt)), key=lambda j: (ignored_gt[j], j)))}
                    taken = set()
                    for rank in range(len(rec["scores"])):
                        need = min(thr, 1 - 1e-10)
                        cand = [j for j in range(len(ignored_gt)) if (rec["crowd"][j] or j not in taken) and rec["iou"][rank][j] >= need]
                        regular = [j for j in cand if not ignored_gt[j]]
                        pool = regular if regular else cand
                        match = max(pool, key=lambda j: (rec["iou"][rank][j], pos[j])) if pool else None
                        if match is not None:
                            taken.add(match)
                            if ignored_gt[match]:
                                continue                                   # matched to an ignored ground truth: the detection is ignored
                        elif rec["d_area"][rank] < lo or rec["d_area"][rank] > hi:
                            continue                                       # unmatched and outside the area range: ignored
                        ranked.append((-rec["scores"][rank], rec["img"], rank, match is not None))
                if n_gt == 0:
                    continue
                ranked.sort(key=lambda r: r[:3])
                tp = fp = 0
                curve = []             # (recall, precision) after each ranked detection
                for _, _, _, hit in ranked:
                    tp, fp = tp + (1 if hit else 0), fp + (0 if hit else 1)
                    curve.append((tp / n_gt, tp / (tp + fp)))
                table[aname][ti][cat] = [max((p for r, p in curve if r >= rt), default=0.0) for rt in _R]

    def mean_over(area: str, t_idx: Sequence[int], cats: Sequence[int]) -> float:
        vals = [v for ti in t_idx for c in cats if table[area][ti][c] is not None for v in table[area][ti][c]]
        return 100.0 * sum(vals) / len(vals) if vals else math.nan
    allc, allt = list(range(num_classes)), list(range(len(_T)))
    out = {"AP": mean_over("all", allt, allc), "AP50": mean_over("all", [0], allc), "AP75": mean_over("all", [5], allc),
           "APs": mean_over("small", allt, allc), "APm": mean_over("medium", allt, allc), "APl": mean_over("large", allt, allc)}
    for c in allc:
        out[f"AP-class{c}"] = mean_over("all", allt, [c])
    return out
