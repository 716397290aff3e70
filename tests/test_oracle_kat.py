"""Known-answer tests that pin the oracle (and the host-side spec code) without a GPU.

The reference ships no tests or golden vectors for this path, so the pins are
(i) the reference YAML itself, (ii) closed-form values derived from the published detectron2 0.6 /
torchvision 0.11.3 algorithms (SURVEY.md §8c list), (iii) PIL for the resize.
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import maskrcnn_oracle as O
from proj_roadsurf_amd.spec import EngineSpec, load_d2_yaml, resize_shortest_edge_shape

REF_YAML = "/root/reference/config/detectron2_config_3bands.yaml"


@pytest.mark.skipif(not os.path.exists(REF_YAML), reason="reference not mounted (GPU box)")
def test_spec_defaults_equal_reference_yaml():
    got = load_d2_yaml(REF_YAML)
    assert got == EngineSpec(), "EngineSpec defaults drifted from R:config/detectron2_config_3bands.yaml"
    assert got.num_classes == 1 and got.detections_per_image == 100 and got.rpn_nms_thresh == 0.7
    assert load_d2_yaml(REF_YAML, num_classes=2).num_classes == 2       # CLI override from COCO categories


def test_spec_rejects_unsupported():
    with pytest.raises(NotImplementedError):
        EngineSpec(norm="BN").check_supported()
    with pytest.raises(NotImplementedError):
        EngineSpec(box_pooler_sampling_ratio=2).check_supported()


def test_cell_anchors_kat():
    a = O.cell_anchors([32.0], [0.5, 1.0, 2.0]).numpy()
    want = np.array([[-22.627417, -11.313708, 22.627417, 11.313708], [-16, -16, 16, 16], [-11.313708, -22.627417, 11.313708, 22.627417]], np.float32)
    assert np.allclose(a, want, atol=1e-5)


def test_grid_anchor_order_and_count():
    spec = EngineSpec()
    total = 0
    for l, (h, w) in enumerate([(200, 200), (100, 100), (50, 50), (25, 25), (13, 13)]):
        g = O.grid_anchors(spec, l, h, w)
        assert g.shape == (h * w * 3, 4)
        total += g.shape[0]
        s = spec.fpn_strides[l]
        # order (y, x, a): entry (y=1, x=2, a=1) is the square anchor centred at (2s, 1s)
        idx = (1 * w + 2) * 3 + 1
        half = spec.anchor_sizes[l][0] / 2
        assert np.allclose(g[idx].numpy(), [2 * s - half, s - half, 2 * s + half, s + half])
    assert total == 159882


def test_scale_clamp_and_apply_deltas():
    spec = EngineSpec()
    assert abs(spec.scale_clamp - 4.135166556742356) < 1e-12
    boxes = torch.tensor([[10.0, 20.0, 30.0, 60.0]])
    out = O.apply_deltas(torch.zeros(1, 4), boxes, (1, 1, 1, 1), spec.scale_clamp)
    assert torch.allclose(out, boxes)
    # dw clamp: exp(clamp) * 20 = 1000/16*20 = 1250 wide
    out = O.apply_deltas(torch.tensor([[0.0, 0.0, 100.0, 0.0]]), boxes, (1, 1, 1, 1), spec.scale_clamp)
    assert abs(float(out[0, 2] - out[0, 0]) - 1250.0) < 1e-2
    # weights (10,10,5,5): dx=1 -> shift by 0.1*w
    out = O.apply_deltas(torch.tensor([[1.0, 0.0, 0.0, 0.0]]), boxes, (10, 10, 5, 5), spec.scale_clamp)
    assert torch.allclose(out, boxes + torch.tensor([[2.0, 0.0, 2.0, 0.0]]))


def test_fpn_level_assignment_kat():
    def lv(side):
        return int(O.assign_levels(torch.tensor([[0.0, 0.0, float(side), float(side)]]), 2, 5)[0]) + 2
    assert [lv(s) for s in (448, 224, 112, 111, 56, 900)] == [5, 4, 3, 2, 2, 5]


def test_resize_shape_kat():
    assert resize_shortest_edge_shape(512, 512, 800, 1333) == (800, 800)
    assert resize_shortest_edge_shape(1024, 1024, 800, 1333) == (800, 800)
    assert resize_shortest_edge_shape(256, 256, 800, 1333) == (800, 800)
    assert resize_shortest_edge_shape(600, 900, 800, 1333) == (800, 1200)
    assert resize_shortest_edge_shape(480, 1000, 800, 1333) == (640, 1333)


def test_nms_semantics():
    # IoU exactly 0.5 is NOT suppressed at thresh 0.5 (torchvision: iou > thresh)
    boxes = np.array([[0, 0, 2, 2], [0, 1, 2, 3], [0, 0, 2, 2.0001], [10, 10, 12, 12]], np.float32)
    iou01 = 2.0 / 6.0
    assert iou01 < 0.5
    keep = O.nms_sorted_np(boxes, 0.5)
    assert keep.tolist() == [True, True, False, True]
    b2 = np.array([[0, 0, 2, 2], [0, 0, 2, 1]], np.float32)     # IoU = 0.5 exactly
    assert O.nms_sorted_np(b2, 0.5).tolist() == [True, True]
    assert O.nms_sorted_np(b2, 0.49).tolist() == [True, False]
    # degenerate boxes: 0/0 = NaN is not > thresh
    z = np.zeros((2, 4), np.float32)
    assert O.nms_sorted_np(z, 0.5).tolist() == [True, True]


def test_batched_nms_variants_agree_and_order():
    g = torch.Generator().manual_seed(0)
    n = 600
    xy = torch.rand(n, 2, generator=g) * 100
    wh = torch.rand(n, 2, generator=g) * 40 + 1
    boxes = torch.cat([xy, xy + wh], 1)
    scores = torch.rand(n, generator=g)
    idxs = torch.randint(0, 3, (n,), generator=g)
    a = O.batched_nms(boxes, scores, idxs, 0.5, coordinate_trick=False)
    b = O.batched_nms(boxes, scores, idxs, 0.5, coordinate_trick=True)
    assert torch.equal(a, b)                      # same keep set & order on generic data
    assert torch.all(scores[a][:-1] >= scores[a][1:])
    # torchvision's rule: numel > 4000 -> per-class loop
    assert torch.equal(O.batched_nms(boxes, scores, idxs, 0.5, coordinate_trick=None), b)


def test_stable_tie_break():
    s = torch.tensor([1.0, 3.0, 3.0, 2.0, 3.0])
    assert O.stable_sort_desc(s).tolist() == [1, 2, 4, 3, 0]


def test_roi_align_constant_and_linear_field():
    # a constant map pools to the constant; a linear ramp pools to the ramp at the bin centres
    H = W = 32
    const = torch.full((1, H, W), 3.5)
    out = O.roi_align_one(const, torch.tensor([16.0, 24.0, 80.0, 100.0]), 7, 0.25)
    assert torch.allclose(out, torch.full_like(out, 3.5), atol=1e-6)
    ramp = torch.arange(W, dtype=torch.float32).view(1, 1, W).expand(1, H, W).contiguous()
    roi = torch.tensor([16.0, 16.0, 72.0, 72.0])          # at scale .25: x in [3.5, 17.5] (after the -0.5 shift)
    out = O.roi_align_one(ramp, roi, 7, 0.25)
    centres = 3.5 + (np.arange(7) + 0.5) * 2.0
    assert np.allclose(out[0, 0].numpy(), centres, atol=1e-5)
    # adaptive sampling: 14 px / 7 bins -> ceil(2) = 2 samples per axis
    # out-of-image samples contribute zero
    out = O.roi_align_one(const, torch.tensor([-400.0, -400.0, -200.0, -200.0]), 7, 0.25)
    assert float(out.abs().max()) == 0.0


def test_roi_align_detectron2_published_vector():
    """Known-answer vector of detectron2's own test-suite (tests/layers/test_roi_align.py::test_forward_output, as
    published with v0.6; detectron2 itself is not present here): 5x5 arange map, box (1,1,3,3), 4x4 output,
    sampling_ratio 0, aligned=True."""
    feat = torch.arange(25, dtype=torch.float32).view(1, 5, 5)
    out = O.roi_align_one(feat, torch.tensor([1.0, 1.0, 3.0, 3.0]), 4, 1.0)
    want = np.array([[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0], [12.0, 12.5, 13.0, 13.5]], np.float32)
    assert np.array_equal(out[0].numpy(), want)


def test_anchor_generator_detectron2_published_vector():
    """detectron2 tests/modeling/test_anchor_generator.py::test_default_anchor_generator (v0.6): sizes (32, 64),
    ratios (0.25, 1, 4), stride 4, a 1x2 feature map, offset 0 -- and ::test_default_anchor_generator_centered
    (offset 0.5 shifts everything by +2)."""
    want = np.array([[-32, -8, 32, 8], [-16, -16, 16, 16], [-8, -32, 8, 32], [-64, -16, 64, 16], [-32, -32, 32, 32], [-16, -64, 16, 64],
                     [-28, -8, 36, 8], [-12, -16, 20, 16], [-4, -32, 12, 32], [-60, -16, 68, 16], [-28, -32, 36, 32], [-12, -64, 20, 64]],
                    np.float32)
    spec = EngineSpec(anchor_sizes=((32.0, 64.0),) * 5, anchor_aspect_ratios=(0.25, 1.0, 4.0))
    assert spec.fpn_strides[0] == 4
    assert np.array_equal(O.grid_anchors(spec, 0, 1, 2).numpy(), want)
    centred = EngineSpec(anchor_sizes=((32.0, 64.0),) * 5, anchor_aspect_ratios=(0.25, 1.0, 4.0), anchor_offset=0.5)
    assert np.array_equal(O.grid_anchors(centred, 0, 1, 2).numpy(), want + 2.0)


def test_paste_masks_full_box():
    # a box covering the whole canvas with a constant 0.7 mask pastes to all-True; 0.3 to all-False
    m = torch.full((1, 1, 28, 28), 0.7)
    box = torch.tensor([[0.0, 0.0, 64.0, 64.0]])
    full = O.paste_masks(m, box, 64, 64)[0]
    # zero padding of grid_sample: the corner pixel centre (0.5, 0.5) maps to ix = iy = -0.28, so it
    # blends 0.7 with zeros: 0.7 * 0.72^2 = 0.36 < 0.5 -> only the 4 corner pixels are False
    assert int((~full).sum()) == 4 and not bool(full[0, 0]) and bool(full[0, 1]) and bool(full[1:-1].all())
    assert not bool(O.paste_masks(torch.full((1, 1, 28, 28), 0.3), box, 64, 64).any())
    # pixels outside the box are False (zero padding < 0.5)
    box = torch.tensor([[16.0, 16.0, 48.0, 48.0]])
    out = O.paste_masks(m, box, 64, 64)[0]
    assert bool(out[20:44, 20:44].all()) and not bool(out[:14].any()) and not bool(out[:, 50:].any())


def test_detector_postprocess_scales_and_filters():
    res = {"boxes": torch.tensor([[0.0, 0.0, 800.0, 400.0], [100.0, 100.0, 100.0, 300.0]]), "scores": torch.tensor([0.9, 0.8]),
           "classes": torch.tensor([0, 1])}
    out = O.detector_postprocess(res, (800, 800), 512, 512)
    assert out["boxes"].shape == (1, 4)                       # zero-width box dropped
    assert torch.allclose(out["boxes"][0], torch.tensor([0.0, 0.0, 512.0, 256.0]))


def test_oracle_forward_small_runs_and_is_deterministic():
    from proj_roadsurf_amd.weights import synthetic_weights
    from tests.util import synthetic_tiles

    spec = EngineSpec(num_classes=2, min_size_test=128, max_size_test=256, rpn_pre_nms_topk_test=100, rpn_post_nms_topk_test=100,
                      detections_per_image=20)
    W = synthetic_weights(spec, 0)
    tiles = synthetic_tiles(1, 96, 96, 3, seed=3)
    m = O.OracleModel(spec, W)
    a = m([tiles[0]])[0]
    b = m([tiles[0]])[0]
    assert torch.equal(a["boxes"], b["boxes"]) and torch.equal(a["masks"], b["masks"])
    assert a["boxes"].shape[0] == a["scores"].shape[0] == a["masks"].shape[0] <= 20
    assert a["masks"].shape[1:] == (96, 96)
    assert torch.all(a["scores"][:-1] >= a["scores"][1:])
