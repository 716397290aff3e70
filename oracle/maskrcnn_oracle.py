"""ORACLE (test infrastructure only) -- CPU fp32 restatement of the detectron2 0.6 Mask R-CNN
R50-FPN *inference* path that the reference drives through the STDL object-detector's
``make_detections.py`` (R:README.md:78, R:config/config_obj_detec.yaml:74-90,
R:config/detectron2_config_3bands.yaml).

PARITY UNPINNED: the arithmetic of this path lives in third-party packages that are pinned by the
reference (R:requirements.txt: detectron2 0.6 :63, torchvision 0.11.3 :281, torch 1.10.2 :277,
pillow 9.2.0 :164) but are *absent* from /root/reference and from this image, and
the reference ships no tests, golden vectors or weights.  This file restates the published
algorithms of those versions; what pins it is (i) every constant of the reference YAML,
(ii) closed-form known-answer tests (tests/test_oracle_kat.py), (iii) PIL itself for the resize
(the one dependency that *is* importable here).  Restatement-vs-detectron2 parity is unverified.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``proj_roadsurf_amd``) never does.

Each function cites the third-party file it follows as ``[EXT d2: <path>]`` (detectron2 0.6) or
``[EXT tv: <path>]`` (torchvision 0.11.3) and the reference YAML line that fixes its parameters
as ``R:<line>`` (= R:config/detectron2_config_3bands.yaml:<line>).
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# The oracle takes the same frozen spec object as the product: it is a plain dataclass of YAML
# values (no arithmetic), so sharing it does not route the product through the oracle.
from proj_roadsurf_amd.spec import EngineSpec, resize_shortest_edge_shape

Tensor = torch.Tensor


# =====================================================================================
# 1. DefaultPredictor.__call__  [EXT d2: engine/defaults.py]  R:26,28,30
# =====================================================================================
def pil_resize(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """``ResizeTransform.apply_image`` for uint8 input [EXT d2: data/transforms/transform.py]:
    ``Image.fromarray(img).resize((new_w, new_h), Image.BILINEAR)``.  Uses PIL itself."""
    from PIL import Image

    assert img.dtype == np.uint8
    if img.shape[2] == 3 or img.shape[2] == 1:
        im = Image.fromarray(img if img.shape[2] == 3 else img[:, :, 0])
        out = np.asarray(im.resize((new_w, new_h), Image.BILINEAR))
        return out if out.ndim == 3 else out[:, :, None]
    # C == 4: PIL would treat the array as RGBA and pre-multiply alpha.  The reference has no
    # 4-band detectron2 YAML; this oracle defines 4-band resize as per-channel (documented
    # deviation, DESIGN.md "4-band").
    chans = [np.asarray(Image.fromarray(np.ascontiguousarray(img[:, :, c])).resize((new_w, new_h), Image.BILINEAR))
             for c in range(img.shape[2])]
    return np.stack(chans, axis=2)


def _pil_bilinear_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Restatement of Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc``
    (src/libImaging/Resample.c, unchanged between 9.2.0 and 12.x) for the bilinear filter.
    Returns (bounds[out,2] = (xmin, n), kk[out,ksize] int32 fixed-point (22 bit), ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                       # bilinear support = 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0 else a
            v = 1.0 - a if a < 1.0 else 0.0
            w[x] = v
            ww += v
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        # normalize_coeffs_8bpc: PRECISION_BITS = 32 - 8 - 2 = 22, round half away from zero
        for x in range(ksize):
            v = w[x] * (1 << 22)
            kk[xx, x] = int(v - 0.5) if v < 0 else int(v + 0.5)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def pil_resize_restated(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """Pure-numpy restatement of Pillow's 2-pass uint8 bilinear resize (horizontal pass first,
    uint8 rounding between passes).  Pinned bit-exact against PIL in tests/test_resize.py; it is
    the algorithm the HIP preprocess kernel implements."""
    h, w, c = img.shape
    cur = img.astype(np.int64)
    if new_w != w:
        b, kk, ks = _pil_bilinear_coeffs(w, new_w)
        out = np.full((h, new_w, c), 1 << 21, np.int64)
        for t in range(ks):
            idx = np.minimum(b[:, 0] + t, w - 1)
            valid = (t < b[:, 1]).astype(np.int64)
            out += cur[:, idx, :] * (kk[:, t].astype(np.int64) * valid)[None, :, None]
        cur = np.clip(out >> 22, 0, 255)
    if new_h != h:
        b, kk, ks = _pil_bilinear_coeffs(h, new_h)
        out = np.full((new_h, cur.shape[1], c), 1 << 21, np.int64)
        for t in range(ks):
            idx = np.minimum(b[:, 0] + t, h - 1)
            valid = (t < b[:, 1]).astype(np.int64)
            out += cur[idx, :, :] * (kk[:, t].astype(np.int64) * valid)[:, None, None]
        cur = np.clip(out >> 22, 0, 255)
    return cur.astype(np.uint8)


def predictor_preprocess(spec: EngineSpec, img_bgr: np.ndarray) -> Tuple[Tensor, Tuple[int, int]]:
    """``DefaultPredictor.__call__`` up to the model call: BGR->RGB flip (R:26), ResizeShortestEdge
    (R:28,30), HWC uint8 -> CHW float32.  Returns (image float32 (C,h,w), (h,w))."""
    assert img_bgr.dtype == np.uint8 and img_bgr.ndim == 3
    img = img_bgr[:, :, ::-1] if spec.input_format == "RGB" else img_bgr
    h, w = img.shape[:2]
    nh, nw = resize_shortest_edge_shape(h, w, spec.min_size_test, spec.max_size_test)
    img = pil_resize(np.ascontiguousarray(img), nh, nw)
    t = torch.as_tensor(img.astype("float32").transpose(2, 0, 1).copy())
    return t, (nh, nw)


def normalize_and_pad(spec: EngineSpec, images: Sequence[Tensor]) -> Tuple[Tensor, List[Tuple[int, int]]]:
    """``GeneralizedRCNN.preprocess_image`` + ``ImageList.from_tensors`` [EXT d2: meta_arch/rcnn.py,
    structures/image_list.py]; mean/std R:81-88, divisibility 32."""
    mean = torch.tensor(spec.pixel_mean, dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(spec.pixel_std, dtype=torch.float32).view(-1, 1, 1)
    normed = [(x - mean) / std for x in images]
    sizes = [(int(x.shape[1]), int(x.shape[2])) for x in normed]
    d = spec.size_divisibility
    mh = max(s[0] for s in sizes)
    mw = max(s[1] for s in sizes)
    mh = (mh + d - 1) // d * d
    mw = (mw + d - 1) // d * d
    out = torch.zeros((len(normed), normed[0].shape[0], mh, mw), dtype=torch.float32)
    for i, x in enumerate(normed):
        out[i, :, : x.shape[1], : x.shape[2]] = x
    return out, sizes


# =====================================================================================
# 3-5. Backbone  [EXT d2: modeling/backbone/resnet.py, fpn.py, layers/batch_norm.py]
# =====================================================================================
def frozen_bn(x: Tensor, W: Dict[str, Tensor], prefix: str, eps: float) -> Tensor:
    """``FrozenBatchNorm2d.forward``: scale = weight * (var + eps).rsqrt(); bias = bias - mean*scale."""
    scale = W[prefix + ".weight"] * (W[prefix + ".running_var"] + eps).rsqrt()
    bias = W[prefix + ".bias"] - W[prefix + ".running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def conv_bn(x: Tensor, W: Dict[str, Tensor], name: str, stride: int, pad: int, eps: float, relu: bool) -> Tensor:
    x = F.conv2d(x, W[name + ".weight"], None, stride=stride, padding=pad)
    x = frozen_bn(x, W, name + ".norm", eps)
    return F.relu(x) if relu else x


def resnet_forward(spec: EngineSpec, W: Dict[str, Tensor], x: Tensor) -> Dict[str, Tensor]:
    """``BasicStem`` + ``BottleneckBlock`` stages (R:100-112)."""
    p = "backbone.bottom_up."
    x = conv_bn(x, W, p + "stem.conv1", 2, 3, spec.bn_eps, True)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs: Dict[str, Tensor] = {"stem": x}
    for si, nblocks in enumerate(spec.res_blocks):
        stage = f"res{si + 2}"
        for bi in range(nblocks):
            name = f"{p}{stage}.{bi}"
            stride = 2 if (bi == 0 and si > 0) else 1
            s1, s3 = (stride, 1) if spec.stride_in_1x1 else (1, stride)
            out = conv_bn(x, W, name + ".conv1", s1, 0, spec.bn_eps, True)
            out = conv_bn(out, W, name + ".conv2", s3, 1, spec.bn_eps, True)
            out = conv_bn(out, W, name + ".conv3", 1, 0, spec.bn_eps, False)
            if (name + ".shortcut.weight") in W:
                sc = conv_bn(x, W, name + ".shortcut", stride, 0, spec.bn_eps, False)
            else:
                sc = x
            x = F.relu(out + sc)
        outs[stage] = x
    return outs


def fpn_forward(spec: EngineSpec, W: Dict[str, Tensor], res: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """``FPN.forward`` + ``LastLevelMaxPool`` (R:61-69).  Returns p2..p6 and the fused
    ``inner`` maps (for stage-wise tests)."""
    names = list(spec.fpn_in_features)            # res2..res5
    lvl = [int(n[3:]) for n in names]
    outs: Dict[str, Tensor] = {}
    prev = None
    for n, l in zip(reversed(names), reversed(lvl)):
        lat = F.conv2d(res[n], W[f"backbone.fpn_lateral{l}.weight"], W[f"backbone.fpn_lateral{l}.bias"])
        if prev is not None:
            top = F.interpolate(prev, scale_factor=2.0, mode="nearest")
            lat = lat + top
        prev = lat
        outs[f"inner{l}"] = lat
        outs[f"p{l}"] = F.conv2d(lat, W[f"backbone.fpn_output{l}.weight"], W[f"backbone.fpn_output{l}.bias"], padding=1)
    outs["p6"] = F.max_pool2d(outs["p5"], kernel_size=1, stride=2, padding=0)
    return outs


# =====================================================================================
# 6-8. RPN  [EXT d2: modeling/proposal_generator/{rpn,proposal_utils}.py, anchor_generator.py,
#            box_regression.py]
# =====================================================================================
def cell_anchors(sizes: Sequence[float], ratios: Sequence[float]) -> Tensor:
    """``DefaultAnchorGenerator.generate_cell_anchors`` (R:45-56): float64 math, stored fp32."""
    out = []
    for size in sizes:
        area = size ** 2.0
        for ar in ratios:
            w = math.sqrt(area / ar)
            h = ar * w
            out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(out, dtype=torch.float32)


def grid_anchors(spec: EngineSpec, level: int, gh: int, gw: int) -> Tensor:
    """``DefaultAnchorGenerator._grid_anchors``: order (y, x, a).  OFFSET R:50."""
    stride = spec.fpn_strides[level]
    base = cell_anchors(spec.anchor_sizes[level], spec.anchor_aspect_ratios)
    sx = torch.arange(spec.anchor_offset * stride, gw * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(spec.anchor_offset * stride, gh * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    xx = xx.reshape(-1)
    yy = yy.reshape(-1)
    shifts = torch.stack((xx, yy, xx, yy), dim=1)
    return (shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4)


def apply_deltas(deltas: Tensor, boxes: Tensor, weights: Sequence[float], scale_clamp: float) -> Tensor:
    """``Box2BoxTransform.apply_deltas`` [EXT d2: modeling/box_regression.py].
    deltas (N, k*4), boxes (N, 4) -> (N, k*4)."""
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = deltas[:, 2::4] / ww
    dh = deltas[:, 3::4] / wh
    dw = torch.clamp(dw, max=scale_clamp)
    dh = torch.clamp(dh, max=scale_clamp)
    pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
    pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
    pred_w = torch.exp(dw) * widths[:, None]
    pred_h = torch.exp(dh) * heights[:, None]
    x1 = pred_ctr_x - 0.5 * pred_w
    y1 = pred_ctr_y - 0.5 * pred_h
    x2 = pred_ctr_x + 0.5 * pred_w
    y2 = pred_ctr_y + 0.5 * pred_h
    return torch.stack((x1, y1, x2, y2), dim=-1).reshape(deltas.shape)


def clip_boxes(boxes: Tensor, size_hw: Tuple[int, int]) -> Tensor:
    """``Boxes.clip`` [EXT d2: structures/boxes.py]."""
    h, w = size_hw
    x1 = boxes[..., 0].clamp(min=0, max=w)
    y1 = boxes[..., 1].clamp(min=0, max=h)
    x2 = boxes[..., 2].clamp(min=0, max=w)
    y2 = boxes[..., 3].clamp(min=0, max=h)
    return torch.stack((x1, y1, x2, y2), dim=-1)


def stable_sort_desc(scores: Tensor) -> Tensor:
    """Index order of a descending sort.  torch/torchvision leave tie order unspecified
    (``sort``/``topk`` are not stable); this framework DEFINES ties as lower index first, in the
    oracle and in the HIP kernels alike."""
    return torch.sort(scores, descending=True, stable=True)[1]


def nms_sorted_np(boxes: np.ndarray, thresh: float) -> np.ndarray:
    """Greedy NMS over boxes already in priority order [EXT tv: csrc/ops/cpu/nms_kernel.cpp and
    cuda/nms_kernel.cu ``devIoU``]: suppress j>i when inter/(areaA+areaB-inter) > thresh, all fp32.
    Returns bool keep mask in the given order."""
    n = boxes.shape[0]
    keep = np.ones(n, dtype=bool)
    if n == 0:
        return keep
    b = boxes.astype(np.float32)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    zero = np.float32(0)
    t = np.float32(thresh)
    for i in range(n):
        if not keep[i]:
            continue
        xx1 = np.maximum(x1[i], x1[i + 1:])
        yy1 = np.maximum(y1[i], y1[i + 1:])
        xx2 = np.minimum(x2[i], x2[i + 1:])
        yy2 = np.minimum(y2[i], y2[i + 1:])
        w = np.maximum(zero, xx2 - xx1)
        h = np.maximum(zero, yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        keep[i + 1:] &= ~(ovr > t)
    return keep


def batched_nms(boxes: Tensor, scores: Tensor, idxs: Tensor, thresh: float, coordinate_trick: Optional[bool] = None) -> Tensor:
    """``torchvision.ops.batched_nms`` [EXT tv: ops/boxes.py].  torchvision 0.11.3 picks the
    per-category loop (``_batched_nms_vanilla``) when ``boxes.numel() > 4000`` and the
    coordinate-offset trick otherwise; ``coordinate_trick=None`` follows that rule,
    ``False`` forces the per-category semantics (what the HIP kernels implement; the two differ
    only by fp32 rounding of the shifted coordinates).  Returns kept indices sorted by score
    (descending, ties: lower index first)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    if coordinate_trick is None:
        coordinate_trick = not (boxes.numel() > 4000)
    if coordinate_trick:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        boxes_for_nms = boxes + offsets[:, None]
        order = stable_sort_desc(scores)
        keep = nms_sorted_np(boxes_for_nms[order].numpy(), thresh)
        return order[torch.from_numpy(keep)]
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for cid in torch.unique(idxs):
        cur = torch.where(idxs == cid)[0]
        order = cur[stable_sort_desc(scores[cur])]
        keep = nms_sorted_np(boxes[order].numpy(), thresh)
        keep_mask[order[torch.from_numpy(keep)]] = True
    keep_idx = torch.where(keep_mask)[0]
    return keep_idx[stable_sort_desc(scores[keep_idx])]


def rpn_head(W: Dict[str, Tensor], feats: Sequence[Tensor]) -> Tuple[List[Tensor], List[Tensor]]:
    """``StandardRPNHead.forward`` (R:230-236)."""
    p = "proposal_generator.rpn_head."
    logits, deltas = [], []
    for x in feats:
        t = F.relu(F.conv2d(x, W[p + "conv.weight"], W[p + "conv.bias"], padding=1))
        logits.append(F.conv2d(t, W[p + "objectness_logits.weight"], W[p + "objectness_logits.bias"]))
        deltas.append(F.conv2d(t, W[p + "anchor_deltas.weight"], W[p + "anchor_deltas.bias"]))
    return logits, deltas


def rpn_proposals(spec: EngineSpec, logits: Sequence[Tensor], deltas: Sequence[Tensor],
                  image_sizes: Sequence[Tuple[int, int]], nms_trick: Optional[bool] = None,
                  ) -> List[Dict[str, Tensor]]:
    """``RPN.predict_proposals`` + ``find_top_rpn_proposals`` (R:245-249, R:90).
    logits[l]: (N, A, H, W); deltas[l]: (N, 4A, H, W).  Returns per image
    {boxes (n,4), logits (n,), level (n,), pre_nms: {...}}."""
    n_img = logits[0].shape[0]
    A = spec.num_anchors
    per_level = []
    for l, (lg, dl) in enumerate(zip(logits, deltas)):
        N, _, H, Wd = lg.shape
        lg_f = lg.permute(0, 2, 3, 1).flatten(1)                                        # (N, HWA)
        dl_f = dl.view(N, A, 4, H, Wd).permute(0, 3, 4, 1, 2).flatten(1, -2)           # (N, HWA, 4)
        anchors = grid_anchors(spec, l, H, Wd)
        k = min(lg_f.shape[1], spec.rpn_pre_nms_topk_test)
        per_level.append((lg_f, dl_f, anchors, k))
    results = []
    for n in range(n_img):
        boxes_l, scores_l, lvl_l, idx_l = [], [], [], []
        for l, (lg_f, dl_f, anchors, k) in enumerate(per_level):
            order = stable_sort_desc(lg_f[n])[:k]
            sc = lg_f[n][order]
            bx = apply_deltas(dl_f[n][order], anchors[order], spec.rpn_bbox_reg_weights, spec.scale_clamp)
            boxes_l.append(bx)
            scores_l.append(sc)
            idx_l.append(order)
            lvl_l.append(torch.full((k,), l, dtype=torch.int64))
        boxes = torch.cat(boxes_l)
        scores = torch.cat(scores_l)
        lvl = torch.cat(lvl_l)
        topk_idx = torch.cat(idx_l)
        pre = {"boxes_decoded": boxes.clone(), "scores": scores.clone(), "level": lvl.clone(), "anchor_idx": topk_idx}
        valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores)
        boxes, scores, lvl = boxes[valid], scores[valid], lvl[valid]
        boxes = clip_boxes(boxes, image_sizes[n])
        w = boxes[:, 2] - boxes[:, 0]
        h = boxes[:, 3] - boxes[:, 1]
        keep = (w > spec.rpn_min_size) & (h > spec.rpn_min_size)
        boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
        pre.update({"boxes_clipped": boxes.clone(), "scores_f": scores.clone(), "level_f": lvl.clone()})
        keep = batched_nms(boxes, scores, lvl, spec.rpn_nms_thresh, coordinate_trick=nms_trick)
        keep = keep[: spec.rpn_post_nms_topk_test]
        results.append({"boxes": boxes[keep], "logits": scores[keep], "level": lvl[keep], "pre_nms": pre})
    return results


# =====================================================================================
# 9. ROIPooler + ROIAlign (aligned=True)  [EXT d2: modeling/poolers.py, layers/roi_align.py;
#    EXT tv: csrc/ops/cpu/roi_align_kernel.cpp]
# =====================================================================================
def assign_levels(boxes: Tensor, min_level: int, max_level: int, canonical_box_size: int = 224, canonical_level: int = 4) -> Tensor:
    """``assign_boxes_to_levels`` [EXT d2: modeling/poolers.py]; returns level - min_level."""
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    box_sizes = torch.sqrt(area)
    lv = torch.floor(canonical_level + torch.log2(box_sizes / canonical_box_size + 1e-8))
    lv = torch.clamp(lv, min=min_level, max=max_level)
    return lv.to(torch.int64) - min_level


def roi_align_one(feat: Tensor, roi: Tensor, out_size: int, spatial_scale: float, sampling_ratio: int = 0) -> Tensor:
    """torchvision ``roi_align`` for ONE roi, ``aligned=True``.  feat (C,H,W) fp32; roi (4,) in
    image coordinates.  Follows the kernel's fp32 operation order (sum over samples in (iy,ix)
    order, then divide by count)."""
    C, H, Wd = feat.shape
    f32 = np.float32
    sc = f32(spatial_scale)
    x1, y1, x2, y2 = [f32(v) for v in roi.tolist()]
    roi_start_w = f32(x1 * sc - f32(0.5))
    roi_start_h = f32(y1 * sc - f32(0.5))
    roi_end_w = f32(x2 * sc - f32(0.5))
    roi_end_h = f32(y2 * sc - f32(0.5))
    roi_w = f32(roi_end_w - roi_start_w)
    roi_h = f32(roi_end_h - roi_start_h)
    bin_h = f32(roi_h / f32(out_size))
    bin_w = f32(roi_w / f32(out_size))
    gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(f32(roi_h / f32(out_size)))))
    gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(f32(roi_w / f32(out_size)))))
    count = f32(max(gh * gw, 1))
    out = torch.zeros((C, out_size, out_size), dtype=torch.float32)
    if gh <= 0 or gw <= 0:
        return out
    ph = np.arange(out_size, dtype=np.float32)
    iy = np.arange(gh, dtype=np.float32)
    ix = np.arange(gw, dtype=np.float32)
    # y = roi_start_h + ph*bin_h + (iy + .5)*bin_h/gh     (fp32, left-to-right like the C++)
    ys = (roi_start_h + ph[:, None] * bin_h) + ((iy[None, :] + f32(0.5)) * bin_h) / f32(gh)   # (P, gh)
    xs = (roi_start_w + ph[:, None] * bin_w) + ((ix[None, :] + f32(0.5)) * bin_w) / f32(gw)   # (P, gw)
    ys = ys.astype(np.float32)
    xs = xs.astype(np.float32)

    def prep(v: np.ndarray, size: int):
        oob = (v < -1.0) | (v > size)
        v = np.where(v <= 0, f32(0), v).astype(np.float32)
        lo = v.astype(np.int32)
        at_edge = lo >= size - 1
        hi = np.where(at_edge, size - 1, lo + 1)
        lo = np.where(at_edge, size - 1, lo)
        v = np.where(at_edge, lo.astype(np.float32), v)
        l = (v - lo.astype(np.float32)).astype(np.float32)
        hh = (f32(1) - l).astype(np.float32)
        return oob, lo, hi, l, hh

    oy, ylo, yhi, ly, hy = prep(ys, H)
    ox, xlo, xhi, lx, hx = prep(xs, Wd)
    acc = torch.zeros((C, out_size, out_size), dtype=torch.float32)
    ft = feat
    for a in range(gh):
        for b in range(gw):
            yl = torch.from_numpy(ylo[:, a]).long()
            yh = torch.from_numpy(yhi[:, a]).long()
            xl = torch.from_numpy(xlo[:, b]).long()
            xh = torch.from_numpy(xhi[:, b]).long()
            w1 = torch.from_numpy(np.outer(hy[:, a], hx[:, b]).astype(np.float32))
            w2 = torch.from_numpy(np.outer(hy[:, a], lx[:, b]).astype(np.float32))
            w3 = torch.from_numpy(np.outer(ly[:, a], hx[:, b]).astype(np.float32))
            w4 = torch.from_numpy(np.outer(ly[:, a], lx[:, b]).astype(np.float32))
            v1 = ft[:, yl][:, :, xl]
            v2 = ft[:, yl][:, :, xh]
            v3 = ft[:, yh][:, :, xl]
            v4 = ft[:, yh][:, :, xh]
            val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
            oob = torch.from_numpy(np.logical_or.outer(oy[:, a], ox[:, b]))
            val = torch.where(oob[None], torch.zeros_like(val), val)
            acc = acc + val
    return acc / float(count)


def roi_align_batched(feat: Tensor, rois: Tensor, out_size: int, spatial_scale: float, chunk: int = 128) -> Tensor:
    """``roi_align_one`` for MANY rois of one feature map, vectorised over the rois that share a sampling grid
    (gh, gw).  Same fp32 operation order per output element (coordinates, bilinear weights, the
    w1*v1 + w2*v2 + w3*v3 + w4*v4 sum, accumulation over (iy, ix), division by the count), so the result is
    bit-identical to the per-roi form (tests/test_oracle_kat.py::test_roi_align_batched_equals_per_roi).
    feat (C,H,W) fp32, rois (R,4) -> (R,C,P,P)."""
    C, H, Wd = feat.shape
    R = int(rois.shape[0])
    P = out_size
    out = torch.zeros((R, C, P, P), dtype=torch.float32)
    if R == 0:
        return out
    f32 = np.float32
    sc = f32(spatial_scale)
    b = rois.numpy().astype(np.float32)
    start_w = (b[:, 0] * sc - f32(0.5)).astype(np.float32)
    start_h = (b[:, 1] * sc - f32(0.5)).astype(np.float32)
    end_w = (b[:, 2] * sc - f32(0.5)).astype(np.float32)
    end_h = (b[:, 3] * sc - f32(0.5)).astype(np.float32)
    roi_w = (end_w - start_w).astype(np.float32)
    roi_h = (end_h - start_h).astype(np.float32)
    bin_h = (roi_h / f32(P)).astype(np.float32)
    bin_w = (roi_w / f32(P)).astype(np.float32)
    gh_all = np.ceil(bin_h.astype(np.float64)).astype(np.int64)
    gw_all = np.ceil(bin_w.astype(np.float64)).astype(np.int64)
    ph = np.arange(P, dtype=np.float32)

    def prep(v: np.ndarray, size: int):
        oob = (v < -1.0) | (v > size)
        v = np.where(v <= 0, f32(0), v).astype(np.float32)
        lo = v.astype(np.int32)
        at_edge = lo >= size - 1
        hi = np.where(at_edge, size - 1, lo + 1)
        lo = np.where(at_edge, size - 1, lo)
        v = np.where(at_edge, lo.astype(np.float32), v)
        l = (v - lo.astype(np.float32)).astype(np.float32)
        hh = (f32(1) - l).astype(np.float32)
        return oob, lo, hi, l, hh

    for gh, gw in sorted(set(zip(gh_all.tolist(), gw_all.tolist()))):
        if gh <= 0 or gw <= 0:
            continue
        sel_all = np.where((gh_all == gh) & (gw_all == gw))[0]
        count = f32(max(gh * gw, 1))
        iy = np.arange(gh, dtype=np.float32)
        ix = np.arange(gw, dtype=np.float32)
        for c0 in range(0, len(sel_all), chunk):
            sel = sel_all[c0:c0 + chunk]
            bh, bw = bin_h[sel][:, None, None], bin_w[sel][:, None, None]
            ys = ((start_h[sel][:, None, None] + ph[None, :, None] * bh) + ((iy[None, None, :] + f32(0.5)) * bh) / f32(gh)).astype(np.float32)
            xs = ((start_w[sel][:, None, None] + ph[None, :, None] * bw) + ((ix[None, None, :] + f32(0.5)) * bw) / f32(gw)).astype(np.float32)
            oy, ylo, yhi, ly, hy = prep(ys, H)         # (r, P, gh)
            ox, xlo, xhi, lx, hx = prep(xs, Wd)        # (r, P, gw)
            acc = torch.zeros((C, len(sel), P, P), dtype=torch.float32)
            T = torch.from_numpy
            for a in range(gh):
                yl, yh = T(ylo[:, :, a]).long()[:, :, None], T(yhi[:, :, a]).long()[:, :, None]      # (r, P, 1)
                for bb in range(gw):
                    xl, xh = T(xlo[:, :, bb]).long()[:, None, :], T(xhi[:, :, bb]).long()[:, None, :]  # (r, 1, P)
                    w1 = T((hy[:, :, a][:, :, None] * hx[:, :, bb][:, None, :]).astype(np.float32))
                    w2 = T((hy[:, :, a][:, :, None] * lx[:, :, bb][:, None, :]).astype(np.float32))
                    w3 = T((ly[:, :, a][:, :, None] * hx[:, :, bb][:, None, :]).astype(np.float32))
                    w4 = T((ly[:, :, a][:, :, None] * lx[:, :, bb][:, None, :]).astype(np.float32))
                    val = w1 * feat[:, yl, xl] + w2 * feat[:, yl, xh] + w3 * feat[:, yh, xl] + w4 * feat[:, yh, xh]
                    oob = T(oy[:, :, a][:, :, None] | ox[:, :, bb][:, None, :])
                    val = torch.where(oob[None], torch.zeros_like(val), val)
                    acc = acc + val
            out[torch.from_numpy(sel)] = (acc / float(count)).permute(1, 0, 2, 3)
    return out


def roi_pooler(feats: Sequence[Tensor], scales: Sequence[float], boxes_per_image: Sequence[Tensor],
               out_size: int, min_level: int = 2, max_level: int = 5) -> Tensor:
    """``ROIPooler.forward`` (R:172-174 box, R:219-221 mask): level assignment then per-level
    RoIAlign.  feats[l]: (N,C,H,W).  Returns (sum n_i, C, out, out) in image-major order."""
    outs = []
    for n, boxes in enumerate(boxes_per_image):
        if boxes.shape[0] == 0:
            continue
        lv = assign_levels(boxes, min_level, max_level)
        res = torch.zeros((boxes.shape[0], feats[0].shape[1], out_size, out_size), dtype=torch.float32)
        for l in range(len(feats)):
            idx = torch.nonzero(lv == l).flatten()
            if idx.numel():
                res[idx] = roi_align_batched(feats[l][n], boxes[idx], out_size, scales[l])
        outs.append(res)
    if not outs:
        return torch.zeros((0, feats[0].shape[1], out_size, out_size), dtype=torch.float32)
    return torch.cat(outs)


# =====================================================================================
# 10-11. Box head  [EXT d2: modeling/roi_heads/{box_head,fast_rcnn}.py]
# =====================================================================================
def box_head(W: Dict[str, Tensor], x: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """``FastRCNNConvFCHead`` (R:167-171) + ``FastRCNNOutputLayers`` linear layers."""
    p = "roi_heads."
    x = x.flatten(1)
    x = F.relu(F.linear(x, W[p + "box_head.fc1.weight"], W[p + "box_head.fc1.bias"]))
    x = F.relu(F.linear(x, W[p + "box_head.fc2.weight"], W[p + "box_head.fc2.bias"]))
    cls = F.linear(x, W[p + "box_predictor.cls_score.weight"], W[p + "box_predictor.cls_score.bias"])
    reg = F.linear(x, W[p + "box_predictor.bbox_pred.weight"], W[p + "box_predictor.bbox_pred.bias"])
    return x, cls, reg


def fast_rcnn_inference_single_image(spec: EngineSpec, boxes: Tensor, scores: Tensor, image_shape: Tuple[int, int],
                                     nms_trick: Optional[bool] = None) -> Dict[str, Tensor]:
    """[EXT d2: modeling/roi_heads/fast_rcnn.py] R:190,194,321.
    boxes (R, 4K) decoded, scores (R, K+1) softmax probabilities."""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not valid.all():
        boxes, scores = boxes[valid], scores[valid]
    scores = scores[:, :-1]
    K = boxes.shape[1] // 4
    boxes = clip_boxes(boxes.reshape(-1, 4), image_shape).view(-1, K, 4)
    filter_mask = scores > spec.score_thresh_test
    filter_inds = filter_mask.nonzero()
    if K == 1:
        b = boxes[filter_inds[:, 0], 0]
    else:
        b = boxes[filter_mask]
    s = scores[filter_mask]
    keep = batched_nms(b, s, filter_inds[:, 1], spec.nms_thresh_test, coordinate_trick=nms_trick)
    if spec.detections_per_image >= 0:
        keep = keep[: spec.detections_per_image]
    return {"boxes": b[keep], "scores": s[keep], "classes": filter_inds[keep, 1], "roi_index": filter_inds[keep, 0]}


# =====================================================================================
# 12-13. Mask head  [EXT d2: modeling/roi_heads/mask_head.py]
# =====================================================================================
def mask_head(spec: EngineSpec, W: Dict[str, Tensor], x: Tensor, classes: Tensor) -> Tuple[Tensor, Tensor]:
    """``MaskRCNNConvUpsampleHead.layers`` (R:215-218) + ``mask_rcnn_inference``.
    Returns (logits (n,K,28,28), probs of the predicted class (n,1,28,28))."""
    p = "roi_heads.mask_head."
    for i in range(spec.mask_num_conv):
        x = F.relu(F.conv2d(x, W[f"{p}mask_fcn{i + 1}.weight"], W[f"{p}mask_fcn{i + 1}.bias"], padding=1))
    x = F.relu(F.conv_transpose2d(x, W[p + "deconv.weight"], W[p + "deconv.bias"], stride=2))
    logits = F.conv2d(x, W[p + "predictor.weight"], W[p + "predictor.bias"])
    n = logits.shape[0]
    if n == 0:
        return logits, logits[:, :1]
    probs = logits[torch.arange(n), classes][:, None].sigmoid()
    return logits, probs


# =====================================================================================
# 14. detector_postprocess + paste_masks_in_image
#     [EXT d2: modeling/postprocessing.py, layers/mask_ops.py]
# =====================================================================================
def paste_masks(masks: Tensor, boxes: Tensor, img_h: int, img_w: int, threshold: float = 0.5) -> Tensor:
    """``paste_masks_in_image`` -> ``_do_paste_mask`` (``skip_empty=False`` form, the one the CUDA
    path runs; the CPU path pastes the same values into a sub-window).  masks (n,1,M,M) probs."""
    n = masks.shape[0]
    if n == 0:
        return torch.zeros((0, img_h, img_w), dtype=torch.bool)
    x0, y0, x1, y1 = torch.split(boxes, 1, dim=1)
    img_y = torch.arange(0, img_h, dtype=torch.float32) + 0.5
    img_x = torch.arange(0, img_w, dtype=torch.float32) + 0.5
    img_y = (img_y - y0) / (y1 - y0) * 2 - 1
    img_x = (img_x - x0) / (x1 - x0) * 2 - 1
    gx = img_x[:, None, :].expand(n, img_y.size(1), img_x.size(1))
    gy = img_y[:, :, None].expand(n, img_y.size(1), img_x.size(1))
    grid = torch.stack([gx, gy], dim=3)
    out = F.grid_sample(masks.float(), grid, align_corners=False)
    return out[:, 0] >= threshold


def detector_postprocess(res: Dict[str, Tensor], image_size: Tuple[int, int], out_h: int, out_w: int,
                         mask_threshold: float = 0.5) -> Dict[str, Tensor]:
    """[EXT d2: modeling/postprocessing.py] scale boxes to the original tile, clip, drop empty,
    paste masks."""
    sx = out_w / image_size[1]
    sy = out_h / image_size[0]
    boxes = res["boxes"].clone()
    boxes[:, 0::2] *= sx
    boxes[:, 1::2] *= sy
    boxes = clip_boxes(boxes, (out_h, out_w))
    keep = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
    out = {"boxes": boxes[keep], "scores": res["scores"][keep], "classes": res["classes"][keep]}
    if "mask_probs" in res:
        out["mask_probs"] = res["mask_probs"][keep]
        out["masks"] = paste_masks(res["mask_probs"][keep], out["boxes"], out_h, out_w, mask_threshold)
    return out


# =====================================================================================
# Whole model
# =====================================================================================
class OracleModel:
    """``GeneralizedRCNN.inference`` restated.  ``W`` maps detectron2 checkpoint key names to fp32
    tensors (SURVEY.md §8c key list)."""

    def __init__(self, spec: EngineSpec, weights: Dict[str, Any], nms_trick: Optional[bool] = False):
        self.spec = spec
        self.W = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) if not isinstance(v, torch.Tensor) else v.float()
                  for k, v in weights.items()}
        # nms_trick: None = torchvision's size rule, False = per-category semantics (engine's)
        self.nms_trick = nms_trick

    @torch.no_grad()
    def backbone(self, x: Tensor) -> Dict[str, Tensor]:
        res = resnet_forward(self.spec, self.W, x)
        fpn = fpn_forward(self.spec, self.W, res)
        res.update(fpn)
        return res

    @torch.no_grad()
    def forward_features(self, feats: Dict[str, Tensor], image_sizes: Sequence[Tuple[int, int]],
                         out_sizes: Sequence[Tuple[int, int]], keep: bool = False) -> List[Dict[str, Any]]:
        spec = self.spec
        rpn_feats = [feats[n] for n in spec.rpn_in_features]
        logits, deltas = rpn_head(self.W, rpn_feats)
        props = rpn_proposals(spec, logits, deltas, image_sizes, nms_trick=self.nms_trick)
        roi_feats = [feats[n] for n in spec.roi_in_features]
        scales = [1.0 / s for s in spec.fpn_strides[: len(roi_feats)]]
        results = []
        for n in range(len(image_sizes)):
            fn = [f[n: n + 1] for f in roi_feats]
            pb = props[n]["boxes"]
            pooled = roi_pooler(fn, scales, [pb], spec.box_pooler_resolution)
            fc, cls, reg = box_head(self.W, pooled)
            probs = F.softmax(cls, dim=-1)
            dec = apply_deltas(reg, pb, spec.box_reg_weights, spec.scale_clamp)
            det = fast_rcnn_inference_single_image(spec, dec, probs, image_sizes[n], nms_trick=self.nms_trick)
            inter: Dict[str, Any] = {}
            if spec.mask_on:
                mp = roi_pooler(fn, scales, [det["boxes"]], spec.mask_pooler_resolution)
                mlog, mprob = mask_head(spec, self.W, mp, det["classes"])
                det["mask_probs"] = mprob
                if keep:
                    inter.update({"mask_pooled": mp, "mask_logits": mlog})
            final = detector_postprocess(det, image_sizes[n], out_sizes[n][0], out_sizes[n][1], spec.mask_threshold)
            if keep:
                inter.update({"rpn_logits": [l[n] for l in logits], "rpn_deltas": [d[n] for d in deltas],
                              "proposals": props[n], "box_pooled": pooled, "fc": fc, "cls_logits": cls, "bbox_deltas": reg,
                              "det_net": det})
                final["inter"] = inter
            results.append(final)
        return results

    @torch.no_grad()
    def __call__(self, images_bgr: Sequence[np.ndarray], keep: bool = False) -> List[Dict[str, Any]]:
        """images: list of HWC uint8 BGR tiles (what ``cv2.imread`` hands ``DefaultPredictor``)."""
        ts, out_sizes = [], []
        for im in images_bgr:
            t, _ = predictor_preprocess(self.spec, im)
            ts.append(t)
            out_sizes.append((im.shape[0], im.shape[1]))
        x, sizes = normalize_and_pad(self.spec, ts)
        feats = self.backbone(x)
        res = self.forward_features(feats, sizes, out_sizes, keep=keep)
        if keep:
            for n, r in enumerate(res):
                r["inter"]["net_input"] = x[n]
                r["inter"]["feats"] = {k: v[n] for k, v in feats.items()}
        return res
