"""Known-answer tests that pin the TRAINING oracle (oracle/train_oracle.py) on the CPU: vectors quoted from detectron2
0.6's own published unit tests where they exist (Matcher, pairwise IoU), closed-form values from the reference YAML
(LR schedule, sampler sizes), and consistency with the inference oracle (RoIAlign, box transform)."""
import math

import numpy as np
import pytest
import torch

from oracle import maskrcnn_oracle as O
from oracle import train_oracle as T
from proj_roadsurf_amd.spec import EngineSpec
from proj_roadsurf_amd.weights import synthetic_weights


def test_matcher_detectron2_published_vector():
    """detectron2 tests/modeling/test_matcher.py::test_scriptability (v0.6): thresholds [0.3, 0.5], labels [0, -1, 1],
    allow_low_quality_matches=True."""
    mq = torch.tensor([[0.15, 0.45, 0.2, 0.6], [0.3, 0.65, 0.05, 0.1], [0.05, 0.4, 0.25, 0.4]])
    m, l = T.matcher(mq, [0.3, 0.5], [0, -1, 1], True)
    assert m.tolist() == [1, 1, 2, 0] and l.tolist() == [-1, 1, 0, 1]
    # without low-quality matches prediction 2 (best IoU 0.25 < 0.3) stays background, prediction 0 (0.3) ignored
    m2, l2 = T.matcher(mq, [0.3, 0.5], [0, -1, 1], False)
    assert m2.tolist() == [1, 1, 2, 0] and l2.tolist() == [-1, 1, 0, 1][:2] + [0, 1]
    # no ground truth: everything is background
    m3, l3 = T.matcher(torch.zeros(0, 5), [0.3, 0.7], [0, -1, 1], True)
    assert m3.tolist() == [0] * 5 and l3.tolist() == [0] * 5


def test_pairwise_iou_detectron2_published_vector():
    """detectron2 tests/structures/test_boxes.py::TestBoxIOU::test_pairwise_iou (v0.6)."""
    b1 = torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 1.0, 1.0]])
    b2 = torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 0.5, 1.0], [0.0, 0.0, 1.0, 0.5], [0.0, 0.0, 0.5, 0.5], [0.5, 0.5, 1.0, 1.0],
                       [0.5, 0.5, 1.5, 1.5]])
    want = torch.tensor([[1.0, 0.5, 0.5, 0.25, 0.25, 0.25 / (2 - 0.25)]] * 2)
    assert torch.allclose(T.pairwise_iou(b1, b2), want)


def test_subsample_labels_sizes_and_membership():
    g = torch.Generator().manual_seed(0)
    labels = torch.full((5000,), 0, dtype=torch.int8)
    labels[torch.randperm(5000, generator=g)[:300]] = 1
    labels[torch.randperm(5000, generator=g)[:500]] = -1
    pos, neg = T.subsample_labels(labels, 256, 0.5, 0, T.default_perm(g))
    assert pos.numel() == min(int((labels == 1).sum()), 128) and neg.numel() == 256 - pos.numel()
    assert bool((labels[pos] == 1).all()) and bool((labels[neg] == 0).all())
    assert len(set(pos.tolist())) == pos.numel() and len(set(neg.tolist())) == neg.numel()
    few = torch.tensor([1, 0, 0, -1, 1, 0], dtype=torch.int8)                # fewer candidates than the quota
    p, n = T.subsample_labels(few, 256, 0.5, 0, T.default_perm(g))
    assert sorted(p.tolist()) == [0, 4] and sorted(n.tolist()) == [1, 2, 5]


def test_get_deltas_inverts_apply_deltas():
    g = torch.Generator().manual_seed(1)
    xy = torch.rand(50, 2, generator=g) * 200
    src = torch.cat([xy, xy + torch.rand(50, 2, generator=g) * 90 + 5], 1)
    xy2 = xy + torch.randn(50, 2, generator=g) * 5
    tgt = torch.cat([xy2, xy2 + torch.rand(50, 2, generator=g) * 90 + 5], 1)
    for w in [(1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0)]:
        d = T.get_deltas(src, tgt, w)
        back = O.apply_deltas(d, src, w, EngineSpec().scale_clamp)
        assert torch.allclose(back, tgt, atol=1e-3)
    # known answer: same centre, double width/height -> (0, 0, ww*log 2, wh*log 2)
    d = T.get_deltas(torch.tensor([[10.0, 10.0, 20.0, 30.0]]), torch.tensor([[5.0, 0.0, 25.0, 40.0]]), (10.0, 10.0, 5.0, 5.0))
    assert torch.allclose(d, torch.tensor([[0.0, 0.0, 5 * math.log(2.0), 5 * math.log(2.0)]]), atol=1e-6)


def test_differentiable_roi_align_equals_inference_oracle():
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(6, 40, 44, generator=g)
    for roi, P, sc in [([16.0, 24.0, 80.0, 100.0], 7, 0.25), ([-30.0, -10.0, 60.0, 44.0], 7, 0.25), ([3.0, 5.0, 300.0, 20.0], 14, 0.125),
                       ([100.0, 100.0, 100.5, 100.5], 7, 0.25), ([150.0, 140.0, 400.0, 400.0], 14, 0.25)]:
        r = torch.tensor(roi)
        a = O.roi_align_one(feat, r, P, sc)
        b = T.roi_align_diff(feat, r, P, sc)
        assert torch.allclose(a, b, atol=2e-5), roi
    f = feat.clone().requires_grad_(True)
    T.roi_align_diff(f, torch.tensor([16.0, 24.0, 80.0, 100.0]), 7, 0.25).sum().backward()
    assert float(f.grad.abs().sum()) > 0


def test_lr_schedule_known_answers():
    ts = T.TrainSpec()
    assert T.lr_at(ts, 0) == pytest.approx(1e-5)
    assert T.lr_at(ts, 100) == pytest.approx(0.01 * (0.001 * 0.5 + 0.5))
    assert T.lr_at(ts, 200) == pytest.approx(0.01) and T.lr_at(ts, 2999) == pytest.approx(0.01)
    assert T.lr_at(ts, 3000) == pytest.approx(0.008) and T.lr_at(ts, 4000) == pytest.approx(0.0064)
    assert T.lr_at(ts, 11999) == pytest.approx(0.01 * 0.8 ** 16)


def test_polygon_rasteriser_invariants():
    """rleFrPoly restated (pycocotools is absent: parity unpinned).  Invariants: an axis-aligned integer rectangle fills
    exactly its pixels, areas of large convex polygons match the analytic area to within the perimeter, the mask stays
    inside the polygon's bounding box, union of two polygons = OR."""
    m = T.polygons_to_bitmask([np.array([1.0, 1.0, 4.0, 1.0, 4.0, 3.0, 1.0, 3.0])], 5, 6)
    want = np.zeros((5, 6), bool)
    want[1:3, 1:4] = True
    assert np.array_equal(m, want)
    th = np.linspace(0, 2 * np.pi, 40, endpoint=False)
    poly = np.stack([50 + 30 * np.cos(th), 40 + 20 * np.sin(th)], 1).reshape(-1)
    m = T.polygons_to_bitmask([poly], 80, 100)
    area = 0.5 * abs(np.dot(poly[0::2], np.roll(poly[1::2], -1)) - np.dot(poly[1::2], np.roll(poly[0::2], -1)))
    assert abs(int(m.sum()) - area) < 2 * np.pi * 30
    ys, xs = np.nonzero(m)
    assert xs.min() >= 19 and xs.max() <= 80 and ys.min() >= 19 and ys.max() <= 60
    a = T.polygons_to_bitmask([np.array([2.0, 2, 10, 2, 10, 10, 2, 10])], 20, 20)
    b = T.polygons_to_bitmask([np.array([8.0, 8, 18, 8, 18, 18, 8, 18])], 20, 20)
    ab = T.polygons_to_bitmask([np.array([2.0, 2, 10, 2, 10, 10, 2, 10]), np.array([8.0, 8, 18, 8, 18, 18, 8, 18])], 20, 20)
    assert np.array_equal(ab, a | b)
    # crop_and_resize: a box that is exactly the rectangle -> a full 28x28 mask
    full = T.rasterize_polygons_within_box([np.array([10.0, 20, 50, 20, 50, 60, 10, 60])], np.array([10.0, 20.0, 50.0, 60.0]), 28)
    assert full.shape == (28, 28) and full.all()


def test_train_forward_small_losses_and_gradients():
    """One training batch on a small configuration: five finite losses with the expected magnitudes at random
    initialisation, gradients on every trainable tensor, none on the frozen stem/res2."""
    spec = EngineSpec(num_classes=2, min_size_test=128, max_size_test=213, rpn_pre_nms_topk_test=100, rpn_post_nms_topk_test=100)
    ts = T.TrainSpec(rpn_pre_nms_topk_train=100, rpn_post_nms_topk_train=50, roi_batch_size_per_image=32)
    Wn = synthetic_weights(spec, seed=0)
    W = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in Wn.items()}
    keys = T.trainable_keys(W)
    assert not any(k.startswith("backbone.bottom_up.res2") or ".stem." in k or ".norm." in k for k in keys)
    assert "backbone.bottom_up.res3.0.conv1.weight" in keys and "roi_heads.mask_head.deconv.weight" in keys
    for k in keys:
        W[k].requires_grad_(True)
    g = torch.Generator().manual_seed(3)
    images = torch.randn(2, 3, 128, 128, generator=g)
    gt_boxes = [torch.tensor([[10.0, 20.0, 70.0, 90.0], [60.0, 30.0, 120.0, 100.0]]), torch.tensor([[30.0, 30.0, 100.0, 110.0]])]
    gt_classes = [torch.tensor([0, 1]), torch.tensor([1])]
    polys = [[[np.array([x0, y0, x1, y0, x1, y1, x0, y1], np.float64)] for x0, y0, x1, y1 in b.tolist()] for b in gt_boxes]
    out = T.train_forward(spec, ts, W, images, [(128, 128)] * 2, gt_boxes, gt_classes, polys, T.default_perm(g))
    names = ["loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg", "loss_mask"]
    total = sum(out[n] for n in names)
    assert all(bool(torch.isfinite(out[n])) for n in names) and float(total) > 0
    total.backward()
    for k in keys:
        assert W[k].grad is not None and bool(torch.isfinite(W[k].grad).all()), k
    assert float(W["backbone.bottom_up.res3.0.conv1.weight"].grad.abs().sum()) > 0
    assert W["backbone.bottom_up.res2.0.conv1.weight"].grad is None
    # sampled sets: at most 25 % foreground, the ground-truth boxes themselves are among the candidates (R:193)
    for s in out["_samples"]:
        k = s["classes"]
        assert int((k < 2).sum()) <= 8 and int((k < 2).sum()) >= 1 and k.numel() <= 32


def test_native_polygon_rasteriser_equals_oracle_restatement():
    """rs_rasterize_polygons_within_box (C++, the product's mask-target path) == oracle/train_oracle.py's Python restatement of
    pycocotools' rleFrPoly + detectron2's rasterize_polygons_within_box, bit for bit, on random polygons and boxes."""
    import os
    from proj_roadsurf_amd.engine import LIB_PATH
    if not os.path.exists(LIB_PATH):
        import __graft_entry__ as g
        g.build()
    from proj_roadsurf_amd.train_targets import rasterize_polygons_within_box as native
    rng = np.random.default_rng(7)
    for t in range(200):
        k, n = int(rng.integers(3, 9)), int(rng.integers(1, 4))
        polys = [(rng.random(2 * k) * 120).astype(np.float64) for _ in range(n)]
        x0, y0 = rng.random(2) * 60
        box = np.array([x0, y0, x0 + rng.random() * 70 + 2, y0 + rng.random() * 70 + 2])
        S = 28 if t % 3 else 14
        assert np.array_equal(T.rasterize_polygons_within_box(polys, box, S), native(polys, box, S)), t
    sq = [np.array([10.0, 20, 50, 20, 50, 60, 10, 60])]
    assert native(sq, np.array([10.0, 20.0, 50.0, 60.0]), 28).all()
    assert not native(sq, np.array([100.0, 100.0, 120.0, 130.0]), 28).any()


def test_batched_rasteriser_equals_the_per_roi_call():
    """rs_rasterize_entries (one native call per training step, host threads over the entries) == rs_rasterize_polygons_within_box
    per entry, bit for bit; multi-polygon instances, repeated instances, float32 boxes as read back from the device; also the
    oracle restatement on a subset."""
    import os
    from proj_roadsurf_amd.engine import LIB_PATH
    if not os.path.exists(LIB_PATH):
        import __graft_entry__ as g
        g.build()
    from proj_roadsurf_amd.train_targets import rasterize_entries, rasterize_polygons_within_box as native
    rng = np.random.default_rng(11)
    instances = [[(rng.random(2 * int(rng.integers(3, 12))) * 300).astype(np.float64) for _ in range(int(rng.integers(1, 4)))] for _ in range(17)]
    ne = 300
    ei = rng.integers(0, len(instances), ne)
    xy = rng.random((ne, 2)) * 200
    boxes = np.concatenate([xy, xy + rng.random((ne, 2)) * 150 + 1], 1).astype(np.float32)
    for threads in (1, 3, 0):
        got = rasterize_entries(instances, ei, boxes, 28, threads=threads)
        assert got.shape == (ne, 28, 28) and got.dtype == bool
        for e in range(ne):
            assert np.array_equal(got[e], native(instances[int(ei[e])], boxes[e], 28)), (threads, e)
    for e in range(0, ne, 25):
        assert np.array_equal(got[e], T.rasterize_polygons_within_box(instances[int(ei[e])], boxes[e].astype(np.float64), 28)), e
    assert rasterize_entries(instances, np.zeros(0, np.int32), np.zeros((0, 4), np.float32), 28).shape == (0, 28, 28)
