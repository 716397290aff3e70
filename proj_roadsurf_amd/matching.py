"""Detection-set comparison: greedy one-to-one matching of two detection sets by class and box IoU, the
agreement metric of the parity tests, of ``bench.py``'s ``parity`` object (fp16 engine against the reference-precision engine,
GPU vs GPU) and of ``tools/parity``.  Pure numpy; no arithmetic of the hot path lives here."""
import math

import numpy as np


def wilson_lower(matched: int, n: int, z: float = 1.96) -> float:
    """Lower end of the Wilson score interval (95 % at z = 1.96) of a matched fraction ``matched / n``."""
    if n <= 0:
        return 0.0
    p = matched / n
    return (p + z * z / (2 * n) - z * math.sqrt(p * (1 - p) / n + z * z / (4 * n * n))) / (1 + z * z / n)


def box_iou(a, b):
    """IoU matrix (len(a), len(b)) of XYXY boxes."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)))
    ix = np.maximum(0, np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]))
    iy = np.maximum(0, np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]))
    inter = ix * iy
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(aa[:, None] + ab[None, :] - inter, 1e-12)


def match_detections(ref, got, min_score=0.1, iou_thr=0.95, dscore_tol=0.02, mask_iou_thr=0.95):
    """Greedy one-to-one matching (same class, box IoU >= thr) of reference detections with score >= min_score.
    ref/got: dicts with boxes (n,4), scores (n,), classes (n,), optional masks (n,H,W) bool.
    Returns dict(frac_matched, max_dscore, min_mask_iou (masks >= 100 px), agg_mask_iou (sum inter / sum union), n_ref,
    n_matched, n_full): ``n_full`` counts the matched pairs that ALSO meet SURVEY section 8d's other two conditions -- |dscore| <=
    ``dscore_tol`` and, when both sides carry masks, mask IoU >= ``mask_iou_thr`` (every mask, no size exemption)."""
    rb, rs, rc = np.asarray(ref["boxes"]), np.asarray(ref["scores"]), np.asarray(ref["classes"])
    gb, gs, gc = np.asarray(got["boxes"]), np.asarray(got["scores"]), np.asarray(got["classes"])
    sel = np.where(rs >= min_score)[0]
    iou = box_iou(rb, gb)
    used = set()
    matched, full, dscore, miou = 0, 0, 0.0, 1.0
    inter_sum, union_sum = 0, 0
    dbox = 0.0
    for i in sel:
        cand = [(iou[i, j], j) for j in range(len(gb)) if j not in used and gc[j] == rc[i] and iou[i, j] >= iou_thr]
        if not cand:
            continue
        _, j = max(cand)
        used.add(j)
        matched += 1
        ds = abs(float(rs[i]) - float(gs[j]))
        dscore = max(dscore, ds)
        ok = ds <= dscore_tol
        dbox = max(dbox, float(np.abs(rb[i] - gb[j]).max()))
        if "masks" in ref and "masks" in got:
            a, b = np.asarray(ref["masks"][i], bool), np.asarray(got["masks"][j], bool)
            u = np.logical_or(a, b).sum()
            it = np.logical_and(a, b).sum()
            inter_sum += it
            union_sum += u
            if u >= 100:     # a 1-pixel flip on a 3-pixel mask is not a meaningful IoU
                miou = min(miou, it / u)
            ok = ok and (it >= mask_iou_thr * u)
        full += int(ok)
    n = len(sel)
    return {"frac_matched": matched / n if n else 1.0, "max_dscore": dscore, "min_mask_iou": miou, "n_ref": n, "n_matched": matched, "n_full": full,
            "agg_mask_iou": (inter_sum / union_sum) if union_sum else 1.0, "max_dbox": dbox}
