"""Frozen engine specification for the Mask R-CNN R50-FPN hot path.

The reference drives its detector with two YAML files:

* ``R:config/detectron2_config_3bands.yaml`` -- a full detectron2 ``CfgNode`` dump (model spec),
* ``R:config/config_obj_detec.yaml:62-90`` -- the ``train_model.py`` / ``make_detections.py`` sections.

This module reads exactly the keys the inference path consumes into a plain, hashable
``EngineSpec`` (no yacs, no detectron2).  Every default below equals the value in the reference
YAML; the file:line next to each field is where that value lives in the reference
(`R:` = ``/root/reference/config/detectron2_config_3bands.yaml``).
``tests/test_spec.py`` checks the defaults against the real YAML whenever ``/root/reference``
is mounted.
"""
from __future__ import annotations

import ast
import dataclasses
import math
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple

import yaml


def _lit(v: Any) -> Any:
    """yacs decodes string values with ``literal_eval`` (e.g. ``WEIGHT_DECAY_BIAS: None``, R:304)."""
    if isinstance(v, str):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v
    return v


def _get(d: Dict[str, Any], path: str, default: Any) -> Any:
    cur: Any = d
    for k in path.split("."):
        if not isinstance(cur, dict) or k not in cur:
            return default
        cur = cur[k]
    return _lit(cur)


@dataclass(frozen=True)
class EngineSpec:
    # ---- INPUT (R:26-30)
    input_format: str = "RGB"                 # R:26  (images arrive BGR, are flipped to RGB)
    min_size_test: int = 800                  # R:30
    max_size_test: int = 1333                 # R:28
    # ---- MODEL.PIXEL_MEAN / STD (R:81-88).  NOTE the quirk: listed in BGR order but applied to
    # the RGB-ordered tensor, i.e. 103.53 is subtracted from R.
    pixel_mean: Tuple[float, ...] = (103.53, 116.28, 123.675)
    pixel_std: Tuple[float, ...] = (1.0, 1.0, 1.0)
    size_divisibility: int = 32               # FPN backbone property (max stride of p5)
    # ---- RESNETS (R:92-112)
    depth: int = 50                           # R:100
    stem_out_channels: int = 64               # R:110
    res2_out_channels: int = 256              # R:108
    num_groups: int = 1                       # R:102
    width_per_group: int = 64                 # R:112
    stride_in_1x1: bool = True                # R:111
    res5_dilation: int = 1                    # R:109
    norm: str = "FrozenBN"                    # R:101
    bn_eps: float = 1e-5                      # FrozenBatchNorm2d default
    # ---- FPN (R:61-69)
    fpn_in_features: Tuple[str, ...] = ("res2", "res3", "res4", "res5")
    fpn_out_channels: int = 256               # R:69
    fpn_fuse_type: str = "sum"                # R:62
    # ---- ANCHOR_GENERATOR (R:40-56)
    anchor_sizes: Tuple[Tuple[float, ...], ...] = ((32.0,), (64.0,), (128.0,), (256.0,), (512.0,))
    anchor_aspect_ratios: Tuple[float, ...] = (0.5, 1.0, 2.0)
    anchor_offset: float = 0.0                # R:50
    # ---- RPN (R:222-251), PROPOSAL_GENERATOR (R:89-91)
    rpn_in_features: Tuple[str, ...] = ("p2", "p3", "p4", "p5", "p6")
    rpn_bbox_reg_weights: Tuple[float, float, float, float] = (1.0, 1.0, 1.0, 1.0)   # R:224-228
    rpn_pre_nms_topk_test: int = 1000         # R:249
    rpn_post_nms_topk_test: int = 1000        # R:247
    rpn_nms_thresh: float = 0.7               # R:245
    rpn_min_size: float = 0.0                 # R:90
    # ---- ROI_HEADS (R:177-194)
    roi_in_features: Tuple[str, ...] = ("p2", "p3", "p4", "p5")
    num_classes: int = 1                      # R:191 (the object-detector CLI overrides from COCO categories)
    score_thresh_test: float = 0.05           # R:194
    nms_thresh_test: float = 0.5              # R:190
    detections_per_image: int = 100           # R:321
    # ---- ROI_BOX_HEAD (R:159-176)
    box_reg_weights: Tuple[float, float, float, float] = (10.0, 10.0, 5.0, 5.0)      # R:160-164
    box_cls_agnostic: bool = False            # R:165
    box_num_conv: int = 0                     # R:170
    box_num_fc: int = 2                       # R:171
    box_fc_dim: int = 1024                    # R:167
    box_pooler_resolution: int = 7            # R:172
    box_pooler_sampling_ratio: int = 0        # R:173
    box_pooler_type: str = "ROIAlignV2"       # R:174
    # ---- ROI_MASK_HEAD (R:213-221)
    mask_on: bool = True                      # R:72
    mask_cls_agnostic: bool = False           # R:214
    mask_conv_dim: int = 256                  # R:215
    mask_num_conv: int = 4                    # R:218
    mask_pooler_resolution: int = 14          # R:219
    mask_pooler_sampling_ratio: int = 0       # R:220
    mask_pooler_type: str = "ROIAlignV2"      # R:221
    mask_threshold: float = 0.5               # detector_postprocess default
    # ---- engine option (not a detectron2 key): "fp16" = production MFMA path, "fp32" = the reference's arithmetic on the fp32 matrix cores,
    # "split" = reference-equivalent arithmetic on the fp16 matrix cores (hi + lo operand planes, three products; inference engines)
    precision: str = "fp16"
    # ---- derived constants
    scale_clamp: float = field(default=math.log(1000.0 / 16.0))   # Box2BoxTransform default

    # ------------------------------------------------------------------ helpers
    @property
    def num_anchors(self) -> int:
        return len(self.anchor_sizes[0]) * len(self.anchor_aspect_ratios)

    @property
    def in_channels(self) -> int:
        return len(self.pixel_mean)

    @property
    def fpn_strides(self) -> Tuple[int, ...]:
        return (4, 8, 16, 32, 64)

    @property
    def res_blocks(self) -> Tuple[int, int, int, int]:
        return {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}[self.depth]

    def replace(self, **kw: Any) -> "EngineSpec":
        return dataclasses.replace(self, **kw)

    def check_supported(self) -> None:
        """Fail loudly on configurations the engine does not implement (never silently differ)."""
        errs = []
        if self.norm != "FrozenBN":
            errs.append(f"RESNETS.NORM={self.norm!r} (only FrozenBN)")
        if self.num_groups != 1 or self.width_per_group != 64:
            errs.append("grouped / wide ResNet")
        if self.res5_dilation != 1:
            errs.append("RES5_DILATION != 1")
        if self.fpn_fuse_type != "sum":
            errs.append("FPN.FUSE_TYPE != sum")
        if self.box_num_conv != 0 or self.box_num_fc != 2:
            errs.append("box head other than 0 conv + 2 fc")
        if self.box_pooler_type != "ROIAlignV2" or self.mask_pooler_type != "ROIAlignV2":
            errs.append("pooler type other than ROIAlignV2")
        if self.box_pooler_sampling_ratio != 0 or self.mask_pooler_sampling_ratio != 0:
            errs.append("fixed POOLER_SAMPLING_RATIO (only adaptive 0)")
        if self.box_cls_agnostic or self.mask_cls_agnostic:
            errs.append("class-agnostic heads")
        if self.depth != 50:
            errs.append(f"DEPTH={self.depth}")
        if errs:
            raise NotImplementedError("unsupported detectron2 config: " + "; ".join(errs))


def spec_from_d2_dict(cfg: Dict[str, Any], num_classes: Optional[int] = None) -> EngineSpec:
    """Build an ``EngineSpec`` from a parsed detectron2 YAML dict (keys as in the reference dump)."""
    g = lambda p, d: _get(cfg, p, d)  # noqa: E731
    base = EngineSpec()
    sizes = g("MODEL.ANCHOR_GENERATOR.SIZES", [list(s) for s in base.anchor_sizes])
    ars = g("MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS", [list(base.anchor_aspect_ratios)])
    if len(ars) != 1:
        raise NotImplementedError("per-level aspect ratios")
    spec = EngineSpec(
        input_format=g("INPUT.FORMAT", base.input_format),
        min_size_test=int(g("INPUT.MIN_SIZE_TEST", base.min_size_test)),
        max_size_test=int(g("INPUT.MAX_SIZE_TEST", base.max_size_test)),
        pixel_mean=tuple(float(x) for x in g("MODEL.PIXEL_MEAN", base.pixel_mean)),
        pixel_std=tuple(float(x) for x in g("MODEL.PIXEL_STD", base.pixel_std)),
        depth=int(g("MODEL.RESNETS.DEPTH", base.depth)),
        stem_out_channels=int(g("MODEL.RESNETS.STEM_OUT_CHANNELS", base.stem_out_channels)),
        res2_out_channels=int(g("MODEL.RESNETS.RES2_OUT_CHANNELS", base.res2_out_channels)),
        num_groups=int(g("MODEL.RESNETS.NUM_GROUPS", base.num_groups)),
        width_per_group=int(g("MODEL.RESNETS.WIDTH_PER_GROUP", base.width_per_group)),
        stride_in_1x1=bool(g("MODEL.RESNETS.STRIDE_IN_1X1", base.stride_in_1x1)),
        res5_dilation=int(g("MODEL.RESNETS.RES5_DILATION", base.res5_dilation)),
        norm=g("MODEL.RESNETS.NORM", base.norm),
        fpn_in_features=tuple(g("MODEL.FPN.IN_FEATURES", base.fpn_in_features)),
        fpn_out_channels=int(g("MODEL.FPN.OUT_CHANNELS", base.fpn_out_channels)),
        fpn_fuse_type=g("MODEL.FPN.FUSE_TYPE", base.fpn_fuse_type),
        anchor_sizes=tuple(tuple(float(x) for x in s) for s in sizes),
        anchor_aspect_ratios=tuple(float(x) for x in ars[0]),
        anchor_offset=float(g("MODEL.ANCHOR_GENERATOR.OFFSET", base.anchor_offset)),
        rpn_in_features=tuple(g("MODEL.RPN.IN_FEATURES", base.rpn_in_features)),
        rpn_bbox_reg_weights=tuple(float(x) for x in g("MODEL.RPN.BBOX_REG_WEIGHTS", base.rpn_bbox_reg_weights)),
        rpn_pre_nms_topk_test=int(g("MODEL.RPN.PRE_NMS_TOPK_TEST", base.rpn_pre_nms_topk_test)),
        rpn_post_nms_topk_test=int(g("MODEL.RPN.POST_NMS_TOPK_TEST", base.rpn_post_nms_topk_test)),
        rpn_nms_thresh=float(g("MODEL.RPN.NMS_THRESH", base.rpn_nms_thresh)),
        rpn_min_size=float(g("MODEL.PROPOSAL_GENERATOR.MIN_SIZE", base.rpn_min_size)),
        roi_in_features=tuple(g("MODEL.ROI_HEADS.IN_FEATURES", base.roi_in_features)),
        num_classes=int(num_classes if num_classes is not None else g("MODEL.ROI_HEADS.NUM_CLASSES", base.num_classes)),
        score_thresh_test=float(g("MODEL.ROI_HEADS.SCORE_THRESH_TEST", base.score_thresh_test)),
        nms_thresh_test=float(g("MODEL.ROI_HEADS.NMS_THRESH_TEST", base.nms_thresh_test)),
        detections_per_image=int(g("TEST.DETECTIONS_PER_IMAGE", base.detections_per_image)),
        box_reg_weights=tuple(float(x) for x in g("MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS", base.box_reg_weights)),
        box_cls_agnostic=bool(g("MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG", base.box_cls_agnostic)),
        box_num_conv=int(g("MODEL.ROI_BOX_HEAD.NUM_CONV", base.box_num_conv)),
        box_num_fc=int(g("MODEL.ROI_BOX_HEAD.NUM_FC", base.box_num_fc)),
        box_fc_dim=int(g("MODEL.ROI_BOX_HEAD.FC_DIM", base.box_fc_dim)),
        box_pooler_resolution=int(g("MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", base.box_pooler_resolution)),
        box_pooler_sampling_ratio=int(g("MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO", base.box_pooler_sampling_ratio)),
        box_pooler_type=g("MODEL.ROI_BOX_HEAD.POOLER_TYPE", base.box_pooler_type),
        mask_on=bool(g("MODEL.MASK_ON", base.mask_on)),
        mask_cls_agnostic=bool(g("MODEL.ROI_MASK_HEAD.CLS_AGNOSTIC_MASK", base.mask_cls_agnostic)),
        mask_conv_dim=int(g("MODEL.ROI_MASK_HEAD.CONV_DIM", base.mask_conv_dim)),
        mask_num_conv=int(g("MODEL.ROI_MASK_HEAD.NUM_CONV", base.mask_num_conv)),
        mask_pooler_resolution=int(g("MODEL.ROI_MASK_HEAD.POOLER_RESOLUTION", base.mask_pooler_resolution)),
        mask_pooler_sampling_ratio=int(g("MODEL.ROI_MASK_HEAD.POOLER_SAMPLING_RATIO", base.mask_pooler_sampling_ratio)),
        mask_pooler_type=g("MODEL.ROI_MASK_HEAD.POOLER_TYPE", base.mask_pooler_type),
    )
    if g("MODEL.META_ARCHITECTURE", "GeneralizedRCNN") != "GeneralizedRCNN":
        raise NotImplementedError("only GeneralizedRCNN")
    return spec


def load_d2_yaml(path: str, num_classes: Optional[int] = None) -> EngineSpec:
    """Parse a detectron2 config dump such as ``R:config/detectron2_config_3bands.yaml``."""
    with open(path, "r") as f:
        cfg = yaml.safe_load(f)
    return spec_from_d2_dict(cfg, num_classes=num_classes)


def resize_shortest_edge_shape(h: int, w: int, short: int, max_size: int) -> Tuple[int, int]:
    """Output (new_h, new_w) of detectron2 ``ResizeShortestEdge.get_output_shape``
    ([EXT d2: data/transforms/augmentation_impl.py]; sizes R:28,30).  KATs: SURVEY.md §8c."""
    scale = short * 1.0 / min(h, w)
    if h < w:
        newh, neww = float(short), scale * w
    else:
        newh, neww = scale * h, float(short)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)
