"""Synthetic workloads (no imagery, weights or labels exist offline: SURVEY.md §8c/§8d).

* ``synthetic_tiles``  -- the SURVEY §8d tile distributions: (U) iid uniform noise, (S) smooth multi-octave noise
  + random filled rectangles ("aerial-like").  Used by bench.py and the parity tests.
* ``synthetic_scenes`` -- tiles WITH ground truth: a smooth background and a few objects of two classes, the shape of
  the reference's task (R:scripts/road_segmentation/determine_class.py:22-25: det_class 0 = artificial, 1 = natural):
  class 0 = uniformly coloured axis-aligned rectangles ("sealed surface"), class 1 = textured ellipses.
* ``train_trained_like`` -- a few hundred SGD steps of the repo's own training engine on a stream of such scenes,
  starting from ``weights.synthetic_weights``: gives weights whose scores separate and whose duplicate proposals
  regress to the same object, as a trained detector's do (the random-weight workload has neither property, so a single
  flipped NMS decision changes its detection set).  Needs a HIP device.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def _smooth_background(rng: np.random.Generator, h: int, w: int, c: int) -> np.ndarray:
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
    return img / img.max() * 200.0


def synthetic_tiles(n: int, h: int, w: int, c: int = 3, seed: int = 1234, kind: str = "S") -> np.ndarray:
    """SURVEY.md §8d synthetic inputs, uint8 (n, h, w, c): (U) iid uniform noise, (S) smooth multi-octave noise + 20
    random filled rectangles per tile."""
    out = np.zeros((n, h, w, c), np.uint8)
    for i in range(n):
        rng = np.random.default_rng(seed + i)
        if kind == "U":
            out[i] = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
            continue
        img = _smooth_background(rng, h, w, c)
        for _ in range(20):
            cx, cy = rng.uniform(0, w), rng.uniform(0, h)
            rw, rh = rng.uniform(4, w / 3), rng.uniform(4, h / 3)
            x0, x1 = int(max(0, cx - rw / 2)), int(min(w, cx + rw / 2))
            y0, y1 = int(max(0, cy - rh / 2)), int(min(h, cy + rh / 2))
            img[y0:y1, x0:x1] = rng.uniform(0, 255, c)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def synthetic_scenes(n: int, h: int, w: int, c: int = 3, seed: int = 0, objects: Tuple[int, int] = (3, 8)
                     ) -> Tuple[np.ndarray, List[np.ndarray], List[np.ndarray], List[List[List[np.ndarray]]]]:
    """Tiles with ground truth.  Returns (tiles uint8 (n,h,w,c), boxes [(k,4) XYXY tile px], classes [(k,)],
    polygons [[ [flat xy array] per object ]]).  Objects do not overlap by more than a corner (rejection sampling)."""
    tiles = np.zeros((n, h, w, c), np.uint8)
    boxes: List[np.ndarray] = []
    classes: List[np.ndarray] = []
    polys: List[List[List[np.ndarray]]] = []
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32) + 0.5
    for i in range(n):
        rng = np.random.default_rng(seed * 1000003 + i)
        img = _smooth_background(rng, h, w, c) * 0.6 + 30.0
        k = int(rng.integers(objects[0], objects[1] + 1))
        bx, cl, pl = [], [], []
        tries = 0
        while len(bx) < k and tries < 200:
            tries += 1
            bw, bh = rng.uniform(0.07, 0.3) * w, rng.uniform(0.07, 0.3) * h
            x0, y0 = rng.uniform(2, w - bw - 2), rng.uniform(2, h - bh - 2)
            b = np.array([x0, y0, x0 + bw, y0 + bh])
            if any(min(b[2], o[2]) - max(b[0], o[0]) > 0.2 * min(bw, o[2] - o[0]) and
                   min(b[3], o[3]) - max(b[1], o[1]) > 0.2 * min(bh, o[3] - o[1]) for o in bx):
                continue
            cls = int(rng.integers(0, 2))
            if cls == 0:      # sealed surface: flat bright grey-ish rectangle
                inside = (xx >= b[0]) & (xx < b[2]) & (yy >= b[1]) & (yy < b[3])
                col = rng.uniform(150, 250) + rng.uniform(-15, 15, c)
                img[inside] = col
                poly = np.array([b[0], b[1], b[2], b[1], b[2], b[3], b[0], b[3]])
            else:             # natural: dark textured ellipse
                cx, cy, rx, ry = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, bw / 2, bh / 2
                inside = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
                base = np.array([rng.uniform(40, 90), rng.uniform(70, 130), rng.uniform(20, 70)][:c] + [60.0] * max(0, c - 3))
                tex = rng.uniform(-25, 25, (h, w, 1)).astype(np.float32)
                img[inside] = (base[None, :] + tex[inside])
                ang = np.linspace(0, 2 * np.pi, 24, endpoint=False)
                poly = np.stack([cx + rx * np.cos(ang), cy + ry * np.sin(ang)], 1).reshape(-1)
            bx.append(b); cl.append(cls); pl.append([poly])
        tiles[i] = np.clip(img, 0, 255).astype(np.uint8)
        boxes.append(np.array(bx, np.float32).reshape(-1, 4))
        classes.append(np.array(cl, np.int64))
        polys.append(pl)
    return tiles, boxes, classes, polys


def detectron2_head_init(spec, W: Dict[str, np.ndarray], seed: int = 0) -> Dict[str, np.ndarray]:
    """Copy of ``W`` with the PREDICTION layers re-drawn at detectron2's own initial scales (RPN objectness / deltas std
    0.01, cls_score 0.01, bbox_pred 0.001, mask predictor 0.001, zero biases  [EXT d2: modeling/proposal_generator/rpn.py,
    roi_heads/fast_rcnn.py, roi_heads/mask_head.py]).  ``weights.synthetic_weights`` draws them large on purpose (spread
    logits for the random-weight workload); SGD from there diverges, as it would for detectron2."""
    rng = np.random.default_rng(seed + 4242)
    out = dict(W)
    for key, std in (("proposal_generator.rpn_head.objectness_logits", 0.01), ("proposal_generator.rpn_head.anchor_deltas", 0.01),
                     ("roi_heads.box_predictor.cls_score", 0.01), ("roi_heads.box_predictor.bbox_pred", 0.001),
                     ("roi_heads.mask_head.predictor", 0.001)):
        if key + ".weight" in out:
            out[key + ".weight"] = (rng.standard_normal(out[key + ".weight"].shape) * std).astype(np.float32)
            out[key + ".bias"] = np.zeros_like(out[key + ".bias"], dtype=np.float32)
    return out


def train_trained_like(spec, tile: int = 512, steps: int = 300, batch: int = 4, seed: int = 0, lr: Optional[float] = None,
                       warmup: int = 100, loss_scale: float = 1024.0, log=None, W0: Optional[Dict[str, np.ndarray]] = None,
                       pool: int = 48) -> Tuple[Dict[str, np.ndarray], List[float]]:
    """``steps`` SGD iterations (reference solver: momentum 0.9, weight decay 1e-4, linear warm-up from 0.001 x lr,
    R:config/detectron2_config_3bands.yaml:268-305) of the training engine on a pool of ``pool`` ``synthetic_scenes``
    (scene seeds ``seed * 7919 + 1 ...``), starting from ``synthetic_weights`` with detectron2's head initialisation.
    ``lr`` defaults to the reference's BASE_LR 0.01 at IMS_PER_BATCH 8 scaled linearly to ``batch`` (0.005 at batch 4): at 0.01 with
    a 50-step warm-up one training seed in three diverged right after the warm-up (round 3, seed 2: loss_cls 8e4 at iteration 73).
    Returns (weights under detectron2 key names, total-loss curve)."""
    from .engine import Trainer
    from .spec import resize_shortest_edge_shape
    from .weights import synthetic_weights

    W = W0 if W0 is not None else detectron2_head_init(spec, synthetic_weights(spec, seed=0), seed)
    tr = Trainer(spec, W, (tile, tile, spec.in_channels), batch=batch, loss_scale=loss_scale)
    nh, nw = resize_shortest_edge_shape(tile, tile, spec.min_size_test, spec.max_size_test)
    sx, sy = nw / tile, nh / tile
    tiles, boxes, classes, polys = synthetic_scenes(pool, tile, tile, spec.in_channels, seed=seed * 7919 + 1)
    nb = [b * np.array([sx, sy, sx, sy], np.float32) for b in boxes]
    npoly = [[[p * np.tile([sx, sy], p.size // 2) for p in inst] for inst in img] for img in polys]
    order = np.random.default_rng(seed)
    curve: List[float] = []
    if lr is None:
        lr = 0.01 * batch / 8.0
    try:
        scale = loss_scale
        for it in range(steps):
            idx = order.choice(pool, size=batch, replace=False)
            losses = tr.train_step(tiles[idx], [nb[i] for i in idx], [classes[i] for i in idx], [npoly[i] for i in idx],
                                   seed=seed * 1000003 + it)
            alpha = min(1.0, it / max(1, warmup))
            tr.apply_sgd(lr * (0.001 * (1 - alpha) + alpha), 0.9, 1e-4)
            if tr.overflowed():
                scale = max(1.0, scale / 2.0)
                tr.set_loss_scale(scale)
            total = float(sum(losses.values()))
            curve.append(total)
            if log and (it % 25 == 0 or it == steps - 1):
                log(f"[train_trained_like] iter {it:4d} total {total:.4f} " + " ".join(f"{k[5:]} {v:.3f}" for k, v in losses.items()))
            if not np.isfinite(total) or total > 1e4:
                raise RuntimeError(f"train_trained_like diverged at iteration {it}: {losses}")
        out = tr.export_weights(W)
    finally:
        tr.close()
    return out, curve
