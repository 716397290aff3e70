"""Shared helpers for the parity tests (synthetic tiles, detection matching)."""
import numpy as np


def synthetic_tiles(n, h, w, c=3, seed=1234, kind="S"):
    """SURVEY.md §8d synthetic inputs: (U) iid uniform noise, (S) smooth multi-octave noise + random
    filled rectangles ("aerial-like")."""
    out = np.zeros((n, h, w, c), np.uint8)
    for i in range(n):
        rng = np.random.default_rng(seed + i)
        if kind == "U":
            out[i] = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
            continue
        img = np.zeros((h, w, c), np.float32)
        for o in range(4):
            g = 4 * (2 ** o)
            coarse = rng.uniform(0, 1, (g + 1, g + 1, c)).astype(np.float32)
            ys = np.linspace(0, g, h, endpoint=False)
            xs = np.linspace(0, g, w, endpoint=False)
            y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
            fy = (ys - y0)[:, None, None]; fx = (xs - x0)[None, :, None]
            a = coarse[y0][:, x0]; b = coarse[y0][:, x0 + 1]; cc = coarse[y0 + 1][:, x0]; d = coarse[y0 + 1][:, x0 + 1]
            img += ((a * (1 - fx) + b * fx) * (1 - fy) + (cc * (1 - fx) + d * fx) * fy) / (2 ** o)
        img = img / img.max() * 200.0
        for _ in range(20):
            cx, cy = rng.uniform(0, w), rng.uniform(0, h)
            rw, rh = rng.uniform(4, w / 3), rng.uniform(4, h / 3)
            x0, x1 = int(max(0, cx - rw / 2)), int(min(w, cx + rw / 2))
            y0, y1 = int(max(0, cy - rh / 2)), int(min(h, cy + rh / 2))
            img[y0:y1, x0:x1] = rng.uniform(0, 255, c)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


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


def match_detections(ref, got, min_score=0.1, iou_thr=0.95):
    """Greedy one-to-one matching (same class, box IoU >= thr) of reference detections with score >= min_score.
    ref/got: dicts with boxes (n,4), scores (n,), classes (n,), optional masks (n,H,W) bool.
    Returns dict(frac_matched, max_dscore, min_mask_iou (masks >= 100 px), agg_mask_iou (sum inter / sum union), n_ref)."""
    rb, rs, rc = np.asarray(ref["boxes"]), np.asarray(ref["scores"]), np.asarray(ref["classes"])
    gb, gs, gc = np.asarray(got["boxes"]), np.asarray(got["scores"]), np.asarray(got["classes"])
    sel = np.where(rs >= min_score)[0]
    iou = box_iou(rb, gb)
    used = set()
    matched, dscore, miou = 0, 0.0, 1.0
    inter_sum, union_sum = 0, 0
    dbox = 0.0
    for i in sel:
        cand = [(iou[i, j], j) for j in range(len(gb)) if j not in used and gc[j] == rc[i] and iou[i, j] >= iou_thr]
        if not cand:
            continue
        _, j = max(cand)
        used.add(j)
        matched += 1
        dscore = max(dscore, abs(float(rs[i]) - float(gs[j])))
        dbox = max(dbox, float(np.abs(rb[i] - gb[j]).max()))
        if "masks" in ref and "masks" in got:
            a, b = np.asarray(ref["masks"][i], bool), np.asarray(got["masks"][j], bool)
            u = np.logical_or(a, b).sum()
            it = np.logical_and(a, b).sum()
            inter_sum += it
            union_sum += u
            if u >= 100:     # a 1-pixel flip on a 3-pixel mask is not a meaningful IoU
                miou = min(miou, it / u)
    n = len(sel)
    return {"frac_matched": matched / n if n else 1.0, "max_dscore": dscore, "min_mask_iou": miou, "n_ref": n,
            "agg_mask_iou": (inter_sum / union_sum) if union_sum else 1.0, "max_dbox": dbox}
