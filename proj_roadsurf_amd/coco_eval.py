"""COCO-style average precision for the detector's validation runs: what detectron2's ``COCOEvaluator``
reports for ``bbox`` and ``segm`` through pycocotools' ``COCOeval`` ([EXT d2: evaluation/coco_evaluation.py]; [EXT coco:
PythonAPI/pycocotools/cocoeval.py]).  pycocotools is not available offline, so the published algorithm is restated here:

  * per image and category, detections sorted by score (stable) and capped at ``max_dets``; IoU against the ground truths
    (crowd regions count as "ignore": IoU uses the detection's area as the union);
  * greedy matching per IoU threshold: a detection takes the still-unmatched ground truth of highest IoU >= threshold
    (non-ignored ground truths first); detections matched to ignored ground truths, and unmatched detections outside the area
    range, are ignored;
  * precision at 101 recall thresholds with the monotone (right-to-left maximum) envelope; AP = mean over recall
    thresholds, IoU thresholds 0.50:0.05:0.95, categories; area ranges all / small / medium / large (32^2, 96^2).

PARITY UNPINNED against pycocotools (absent); tests/test_coco_eval.py pins closed-form cases (perfect detections AP = 1, a
known precision/recall staircase, score ordering, area ranges, crowd handling) and checks this module on 220 random detection /
ground-truth sets (crowd regions, empty images, tied scores, max_dets caps, annotated areas; bbox and segm) against an independent
brute-force statement of the definitions (oracle/coco_ap_oracle.py: declarative matching, interpolated precision by its definition;
test infrastructure, never imported here) to 1e-9.  Masks are numpy bool arrays here (the engine's
``Instances.pred_masks`` / rasterised ground-truth polygons), boxes XYXY."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, 10)
REC_THRS = np.linspace(0.0, 1.0, 101)
AREA_RNG = {"all": (0.0, 1e10), "small": (0.0, 32.0 ** 2), "medium": (32.0 ** 2, 96.0 ** 2), "large": (96.0 ** 2, 1e10)}


def box_iou(d: np.ndarray, g: np.ndarray, crowd: np.ndarray) -> np.ndarray:
    """(D,4) x (G,4) XYXY -> (D,G); for crowd ground truths the union is the detection's area (maskApi bbIou)."""
    if d.shape[0] == 0 or g.shape[0] == 0:
        return np.zeros((d.shape[0], g.shape[0]))
    ad = (d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1])
    ag = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
    w = np.clip(np.minimum(d[:, None, 2], g[None, :, 2]) - np.maximum(d[:, None, 0], g[None, :, 0]), 0, None)
    h = np.clip(np.minimum(d[:, None, 3], g[None, :, 3]) - np.maximum(d[:, None, 1], g[None, :, 1]), 0, None)
    inter = w * h
    union = np.where(crowd[None, :], ad[:, None], ad[:, None] + ag[None, :] - inter)
    return np.where(union > 0, inter / np.maximum(union, 1e-300), 0.0)


def mask_iou(d: np.ndarray, g: np.ndarray, crowd: np.ndarray) -> np.ndarray:
    """(D,H,W) x (G,H,W) bool -> (D,G)."""
    if d.shape[0] == 0 or g.shape[0] == 0:
        return np.zeros((d.shape[0], g.shape[0]))
    df = d.reshape(d.shape[0], int(np.prod(d.shape[1:]))).astype(np.float64)
    gf = g.reshape(g.shape[0], int(np.prod(g.shape[1:]))).astype(np.float64)
    inter = df @ gf.T
    ad, ag = df.sum(1), gf.sum(1)
    union = np.where(crowd[None, :], ad[:, None], ad[:, None] + ag[None, :] - inter)
    return np.where(union > 0, inter / np.maximum(union, 1e-300), 0.0)


def _evaluate_img(ious: np.ndarray, d_scores: np.ndarray, d_area: np.ndarray, g_area: np.ndarray, g_crowd: np.ndarray,
                  rng: Tuple[float, float], max_det: int):
    """COCOeval.evaluateImg for one (image, category, area range): detection match flags / ignore flags per IoU threshold."""
    g_ignore = g_crowd | (g_area < rng[0]) | (g_area > rng[1])
    gorder = np.argsort(g_ignore, kind="mergesort")                 # non-ignored ground truths first
    g_ignore, g_crowd_o = g_ignore[gorder], g_crowd[gorder]
    dorder = np.argsort(-d_scores, kind="mergesort")[:max_det]
    iou = ious[dorder][:, gorder] if ious.size else np.zeros((len(dorder), len(gorder)))
    T, D, G = len(IOU_THRS), len(dorder), len(gorder)
    dtm = np.zeros((T, D), bool)
    dt_ig = np.zeros((T, D), bool)
    for ti, t in enumerate(IOU_THRS):
        gtm = np.zeros(G, bool)
        for di in range(D):
            best, m = min(t, 1 - 1e-10), -1
            for gi in range(G):
                if gtm[gi] and not g_crowd_o[gi]:
                    continue
                if m > -1 and not g_ignore[m] and g_ignore[gi]:
                    break                                           # already matched a regular gt: ignored ones cannot replace it
                if iou[di, gi] < best:
                    continue
                best, m = iou[di, gi], gi
            if m == -1:
                continue
            dt_ig[ti, di] = g_ignore[m]
            dtm[ti, di] = True
            gtm[m] = True
    out_of_range = (d_area[dorder] < rng[0]) | (d_area[dorder] > rng[1])
    dt_ig = dt_ig | (~dtm & out_of_range[None, :])
    return d_scores[dorder], dtm, dt_ig, int((~g_ignore).sum())


def match_images(gts: Sequence[Dict], dets: Sequence[Dict], num_classes: int, iou_type: str = "bbox", max_dets: int = 100) -> List[Dict]:
    """First half of COCOeval (``evaluateImg`` for every image, category and area range).  Returns one dict per image,
    ``{(category, area name): (scores, matched (T,D), ignored (T,D), n non-ignored gt)}`` -- a few hundred bytes per image, so
    ranks of a data-parallel job evaluate their own share of the validation set (mask IoUs included) and only these
    records travel to rank 0 for ``accumulate``."""
    assert iou_type in ("bbox", "segm") and len(gts) == len(dets)
    out: List[Dict] = []
    for g, d in zip(gts, dets):
        rec: Dict = {}
        for c in range(num_classes):
            gm, dm = np.asarray(g["classes"]) == c, np.asarray(d["classes"]) == c
            if not gm.any() and not dm.any():
                continue
            gb, db = np.asarray(g["boxes"], np.float64).reshape(-1, 4)[gm], np.asarray(d["boxes"], np.float64).reshape(-1, 4)[dm]
            crowd = np.asarray(g.get("crowd", np.zeros(len(g["classes"]), bool)), bool)[gm]
            sc = np.asarray(d["scores"], np.float64)[dm]
            if iou_type == "segm":
                gmk, dmk = np.asarray(g["masks"], bool)[gm], np.asarray(d["masks"], bool)[dm]
                ious = mask_iou(dmk, gmk, crowd)
                g_area = gmk.sum(axis=(1, 2)).astype(np.float64) if gmk.ndim == 3 else np.zeros(gmk.shape[0])
                d_area = dmk.sum(axis=(1, 2)).astype(np.float64) if dmk.ndim == 3 else np.zeros(dmk.shape[0])
            else:
                ious = box_iou(db, gb, crowd)
                g_area = (gb[:, 2] - gb[:, 0]) * (gb[:, 3] - gb[:, 1])
                d_area = (db[:, 2] - db[:, 0]) * (db[:, 3] - db[:, 1])
            if "area" in g:
                g_area = np.asarray(g["area"], np.float64)[gm]
            for a, rng in AREA_RNG.items():
                rec[(c, a)] = _evaluate_img(ious, sc, d_area, g_area, crowd, rng, max_dets)
        out.append(rec)
    return out


def accumulate(per_image: Sequence[Dict], num_classes: int) -> Dict[str, float]:
    """Second half of COCOeval (``accumulate`` + ``summarize``) over the per-image records of ``match_images``, in image
    order.  Returns COCO's AP, AP50, AP75, APs, APm, APl (in percent, NaN where undefined) and the per-category AP."""
    res: Dict[str, float] = {}
    prec = {a: np.full((len(IOU_THRS), len(REC_THRS), num_classes), -1.0) for a in AREA_RNG}
    for c in range(num_classes):
        for a in AREA_RNG:
            ents = [r[(c, a)] for r in per_image if (c, a) in r]
            if not ents:
                continue
            scores = np.concatenate([e[0] for e in ents])
            order = np.argsort(-scores, kind="mergesort")
            dtm = np.concatenate([e[1] for e in ents], axis=1)[:, order]
            dt_ig = np.concatenate([e[2] for e in ents], axis=1)[:, order]
            npig = sum(e[3] for e in ents)
            if npig == 0:
                continue
            tps = np.cumsum(dtm & ~dt_ig, axis=1, dtype=np.float64)
            fps = np.cumsum(~dtm & ~dt_ig, axis=1, dtype=np.float64)
            for ti in range(len(IOU_THRS)):
                tp, fp = tps[ti], fps[ti]
                rc = tp / npig
                pr = tp / np.maximum(tp + fp, np.spacing(1))
                q = np.zeros(len(REC_THRS))
                pr = pr.tolist()
                for i in range(len(pr) - 1, 0, -1):
                    if pr[i] > pr[i - 1]:
                        pr[i - 1] = pr[i]
                inds = np.searchsorted(rc, REC_THRS, side="left")
                for ri, pi in enumerate(inds):
                    if pi < len(pr):
                        q[ri] = pr[pi]
                prec[a][ti, :, c] = q

    def mean_valid(x: np.ndarray) -> float:
        v = x[x > -1]
        return float(v.mean() * 100) if v.size else float("nan")
    res["AP"] = mean_valid(prec["all"])
    res["AP50"] = mean_valid(prec["all"][0])
    res["AP75"] = mean_valid(prec["all"][5])
    res["APs"], res["APm"], res["APl"] = mean_valid(prec["small"]), mean_valid(prec["medium"]), mean_valid(prec["large"])
    for c in range(num_classes):
        res[f"AP-class{c}"] = mean_valid(prec["all"][:, :, c])
    return res


def evaluate(gts: Sequence[Dict], dets: Sequence[Dict], num_classes: int, iou_type: str = "bbox", max_dets: int = 100) -> Dict[str, float]:
    """gts[i] / dets[i] describe image i: {"boxes" (k,4) XYXY, "classes" (k,), ["masks" (k,H,W) bool], ["crowd" (k,) bool]} and
    {"boxes", "classes", "scores", ["masks"]}.  ``accumulate(match_images(...))``."""
    return accumulate(match_images(gts, dets, num_classes, iou_type, max_dets), num_classes)
