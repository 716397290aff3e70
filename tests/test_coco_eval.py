"""Closed-form checks of proj_roadsurf_amd/coco_eval.py (pycocotools' COCOeval restated; parity unpinned)."""
import numpy as np
import pytest

from proj_roadsurf_amd.coco_eval import box_iou, evaluate, mask_iou


def _img(boxes, classes, scores=None, masks=None, crowd=None):
    d = {"boxes": np.asarray(boxes, np.float64).reshape(-1, 4), "classes": np.asarray(classes, np.int64)}
    if scores is not None:
        d["scores"] = np.asarray(scores, np.float64)
    if masks is not None:
        d["masks"] = masks
    if crowd is not None:
        d["crowd"] = np.asarray(crowd, bool)
    return d


def test_perfect_detections_score_100():
    g = [_img([[10, 10, 60, 60], [100, 100, 220, 200]], [0, 1]), _img([[5, 5, 25, 25]], [0])]
    d = [_img(x["boxes"], x["classes"], scores=[0.9] * len(x["classes"])) for x in g]
    r = evaluate(g, d, 2, "bbox")
    assert r["AP"] == pytest.approx(100.0) and r["AP50"] == pytest.approx(100.0) and r["AP-class1"] == pytest.approx(100.0)
    assert r["APs"] == pytest.approx(100.0) and r["APm"] == pytest.approx(100.0) and r["APl"] == pytest.approx(100.0)


def test_precision_recall_staircase_known_answer():
    """One class, 2 ground truths; detections by score: TP, FP, TP.  Precision envelope: 1.0 up to recall 0.5, 2/3 up to recall 1
    -> AP@0.5 = (51 * 1.0 + 50 * 2/3) / 101."""
    g = [_img([[0, 0, 10, 10], [20, 20, 30, 30]], [0, 0])]
    d = [_img([[0, 0, 10, 10], [50, 50, 60, 60], [20, 20, 30, 30]], [0, 0, 0], scores=[0.9, 0.8, 0.7])]
    r = evaluate(g, d, 1, "bbox")
    assert r["AP50"] == pytest.approx((51 * 1.0 + 50 * 2 / 3) / 101 * 100)
    assert r["AP"] == pytest.approx(r["AP50"])                     # IoU 1.0 matches pass every threshold
    # the false positive ranked LAST costs nothing
    d2 = [_img([[0, 0, 10, 10], [20, 20, 30, 30], [50, 50, 60, 60]], [0, 0, 0], scores=[0.9, 0.8, 0.1])]
    assert evaluate(g, d2, 1, "bbox")["AP"] == pytest.approx(100.0)


def test_iou_thresholds_and_localisation_quality():
    """A detection with IoU 0.6 counts at thresholds 0.50, 0.55, 0.60 only: AP = 3/10 * 100."""
    g = [_img([[0, 0, 10, 10]], [0])]
    d = [_img([[0, 0, 10, 6]], [0], scores=[0.5])]                  # IoU = 60 / 100
    r = evaluate(g, d, 1, "bbox")
    assert r["AP50"] == pytest.approx(100.0) and r["AP75"] == pytest.approx(0.0) and r["AP"] == pytest.approx(30.0)


def test_duplicate_detections_and_missed_ground_truth():
    g = [_img([[0, 0, 10, 10], [40, 40, 60, 60]], [0, 0])]
    d = [_img([[0, 0, 10, 10], [0, 0, 10, 10]], [0, 0], scores=[0.9, 0.8])]     # second is a duplicate -> FP; one gt missed
    r = evaluate(g, d, 1, "bbox")
    assert r["AP50"] == pytest.approx(51 / 101 * 100)               # precision 1 up to recall 0.5, nothing beyond


def test_area_ranges_and_crowd_regions():
    small, large = [0, 0, 20, 20], [100, 100, 300, 300]             # areas 400 (< 32^2) and 40000 (> 96^2)
    g = [_img([small, large], [0, 0])]
    d = [_img([small], [0], scores=[0.9])]
    r = evaluate(g, d, 1, "bbox")
    assert r["APs"] == pytest.approx(100.0) and r["APl"] == pytest.approx(0.0) and np.isnan(r["APm"])
    # a detection inside a crowd region is ignored (neither TP nor FP); IoU with a crowd uses the detection area as union
    g2 = [_img([[0, 0, 10, 10], [50, 50, 150, 150]], [0, 0], crowd=[False, True])]
    d2 = [_img([[0, 0, 10, 10], [60, 60, 80, 80]], [0, 0], scores=[0.9, 0.8])]
    assert box_iou(d2[0]["boxes"][1:], g2[0]["boxes"][1:], np.array([True]))[0, 0] == pytest.approx(1.0)
    assert evaluate(g2, d2, 1, "bbox")["AP"] == pytest.approx(100.0)


def test_segm_iou_and_evaluation():
    H = W = 40
    gm = np.zeros((1, H, W), bool); gm[0, 10:30, 10:30] = True
    dm = np.zeros((2, H, W), bool); dm[0, 10:30, 10:20] = True; dm[1, 10:30, 10:30] = True
    iou = mask_iou(dm, gm, np.array([False]))
    assert iou[0, 0] == pytest.approx(0.5) and iou[1, 0] == pytest.approx(1.0)
    g = [_img([[10, 10, 30, 30]], [0], masks=gm)]
    d = [_img([[10, 10, 20, 30], [10, 10, 30, 30]], [0, 0], scores=[0.9, 0.8], masks=dm)]
    r = evaluate(g, d, 1, "segm")
    # threshold 0.50: the half mask (score 0.9) matches first, the full mask becomes a FP -> AP 100; above 0.5: FP then TP -> 0.5
    assert r["AP50"] == pytest.approx(100.0) and r["AP75"] == pytest.approx(50.0)
    assert r["AP"] == pytest.approx((100.0 + 9 * 50.0) / 10)


def test_max_dets_and_empty_inputs():
    g = [_img([[0, 0, 10, 10]], [0])]
    boxes = [[100 + i, 100, 110 + i, 110] for i in range(5)] + [[0, 0, 10, 10]]
    d = [_img(boxes, [0] * 6, scores=[0.9, 0.8, 0.7, 0.6, 0.5, 0.4])]
    assert evaluate(g, d, 1, "bbox", max_dets=5)["AP"] == pytest.approx(0.0)          # the true positive is cut off
    assert evaluate(g, d, 1, "bbox", max_dets=100)["AP50"] == pytest.approx(1 / 6 * 100)
    r = evaluate([_img(np.zeros((0, 4)), [])], [_img(np.zeros((0, 4)), [], scores=[])], 1, "bbox")
    assert np.isnan(r["AP"])
    # segm with an image that has ground truth but no detection of the class (empty mask stacks)
    gm = np.zeros((1, 8, 8), bool); gm[0, 2:6, 2:6] = True
    r = evaluate([_img([[2, 2, 6, 6]], [0], masks=gm)], [_img(np.zeros((0, 4)), [], scores=[], masks=np.zeros((0, 8, 8), bool))], 1, "segm")
    assert r["AP"] == pytest.approx(0.0)


def test_cocoeval_rules_crowd_area_range_maxdets_ties():
    """The rules of pycocotools' COCOeval.evaluateImg / accumulate that decide ignore flags and ordering, each on a case whose answer
    follows from the published rule (pycocotools itself is absent: parity unpinned):
    * a crowd region absorbs ANY number of detections (its gtm flag never blocks) and they become "ignored", not false positives;
    * a detection that reaches a regular ground truth keeps it even when a crowd region overlaps it more (regular gts are scanned
      first and the scan stops at the first ignored gt once a regular match exists);
    * per area range, ground truths outside the range are "ignore", detections matched to them are ignored, UNMATCHED detections
      outside the range are ignored too;
    * maxDets caps the detections per (image, category), highest scores first;
    * equal scores keep their input order (mergesort)."""
    # crowd absorbs two detections
    g = [_img([[0, 0, 10, 10], [50, 50, 150, 150]], [0, 0], crowd=[False, True])]
    d = [_img([[0, 0, 10, 10], [60, 60, 80, 80], [100, 100, 120, 120]], [0, 0, 0], scores=[0.9, 0.8, 0.7])]
    assert evaluate(g, d, 1, "bbox")["AP"] == pytest.approx(100.0)
    # duplicate of a detected object inside a crowd region: ignored with the crowd flag, a false positive without it
    gb = [[0, 0, 10, 10], [0, 0, 10, 12], [40, 40, 60, 60]]
    dd = [_img([[0, 0, 10, 10], [0, 0, 10, 10], [40, 40, 60, 60]], [0, 0, 0], scores=[0.9, 0.8, 0.7])]
    assert evaluate([_img(gb, [0, 0, 0], crowd=[False, True, False])], dd, 1, "bbox")["AP50"] == pytest.approx(100.0)
    stair = (51 * 1.0 + 50 * 2 / 3) / 101 * 100
    assert evaluate([_img([gb[0], gb[2]], [0, 0])], dd, 1, "bbox")["AP50"] == pytest.approx(stair)
    # the first detection keeps the REGULAR ground truth although the crowd's IoU (intersection over detection area) is as high
    one = [_img([[0, 0, 10, 10]], [0], scores=[0.9])]
    assert evaluate([_img([[0, 0, 10, 10], [0, 0, 10, 12]], [0, 0], crowd=[False, True])], one, 1, "bbox")["AP"] == pytest.approx(100.0)
    # area ranges
    g = [_img([[0, 0, 20, 20], [100, 100, 300, 300]], [0, 0])]                      # small (undetected), large (detected)
    d = [_img([[100, 100, 300, 300], [400, 400, 600, 600]], [0, 0], scores=[0.9, 0.5])]   # + an unmatched large detection
    r = evaluate(g, d, 1, "bbox")
    assert r["APl"] == pytest.approx(100.0)                  # small gt ignored; TP, then the FP ranked last
    assert r["APs"] == pytest.approx(0.0)                    # the small gt is never reached; both detections are ignored in this range
    assert r["AP50"] == pytest.approx(51 / 101 * 100)        # all areas: two gts, TP then FP
    # maxDets per (image, category)
    g = [_img([[0, 0, 10, 10], [0, 0, 10, 10]], [0, 1])]
    d = [_img([[50, 50, 60, 60], [70, 70, 80, 80], [0, 0, 10, 10], [0, 0, 10, 10]], [0, 0, 0, 1], scores=[0.9, 0.8, 0.7, 0.6])]
    r = evaluate(g, d, 2, "bbox", max_dets=2)
    assert r["AP-class0"] == pytest.approx(0.0) and r["AP-class1"] == pytest.approx(100.0) and r["AP"] == pytest.approx(50.0)
    # ties: input order decides
    g = [_img([[0, 0, 10, 10]], [0])]
    fp_first = [_img([[50, 50, 60, 60], [0, 0, 10, 10]], [0, 0], scores=[0.5, 0.5])]
    tp_first = [_img([[0, 0, 10, 10], [50, 50, 60, 60]], [0, 0], scores=[0.5, 0.5])]
    assert evaluate(g, fp_first, 1, "bbox")["AP50"] == pytest.approx(50.0) and evaluate(g, tp_first, 1, "bbox")["AP50"] == pytest.approx(100.0)


# ---------------------------------------------------------------------------------------------------------------------------------
# Independent checker (oracle/coco_ap_oracle.py): a brute-force statement of the published definitions -- declarative matching, interpolated
# precision by its definition, explicit ranked lists -- against the COCOeval-shaped implementation, on random scenes.  Both remain PARITY
# UNPINNED against pycocotools (absent offline); what this pins is that two differently built statements of the rules agree.
# ---------------------------------------------------------------------------------------------------------------------------------
def _random_scene(rng, n_img, n_cls, with_masks, side=64):
    gts, dets = [], []
    for _ in range(n_img):
        ng = int(rng.integers(0, 7)) if rng.random() > 0.15 else 0           # some images without ground truth
        gb, gc, gm = [], [], []
        for _ in range(ng):
            s = float(rng.choice([6.0, 14.0, 30.0, 45.0]))                     # areas on both sides of 32^2 (and of 96^2 through "area" below)
            x, y = float(rng.uniform(0, side - s)), float(rng.uniform(0, side - s))
            gb.append([x, y, x + s * float(rng.uniform(0.6, 1.0)), y + s * float(rng.uniform(0.6, 1.0))])
            gc.append(int(rng.integers(0, n_cls)))
        g = {"boxes": np.asarray(gb, np.float64).reshape(-1, 4), "classes": np.asarray(gc, np.int64), "crowd": rng.random(ng) < 0.15}
        db, dc, ds = [], [], []
        for k in range(ng):                                                    # jittered copies (some twice: duplicates), some with the wrong class
            for _ in range(int(rng.integers(0, 3))):
                j = rng.normal(0, 1.5, 4)
                db.append((g["boxes"][k] + j).tolist())
                dc.append(gc[k] if rng.random() > 0.1 else int(rng.integers(0, n_cls)))
                ds.append(round(float(rng.uniform(0.05, 1.0)), 1 if rng.random() < 0.5 else 3))      # coarse scores: many ties
        for _ in range(int(rng.integers(0, 5))):                               # false positives anywhere
            s = float(rng.uniform(4, 40))
            x, y = float(rng.uniform(0, side - 4)), float(rng.uniform(0, side - 4))
            db.append([x, y, x + s, y + s]); dc.append(int(rng.integers(0, n_cls))); ds.append(round(float(rng.uniform(0.05, 1.0)), 2))
        d = {"boxes": np.asarray(db, np.float64).reshape(-1, 4), "classes": np.asarray(dc, np.int64), "scores": np.asarray(ds, np.float64)}
        if with_masks:
            def raster(boxes):
                m = np.zeros((len(boxes), side, side), bool)
                for i, b in enumerate(boxes):
                    x0, y0, x1, y1 = (int(round(v)) for v in np.clip(b, 0, side))
                    m[i, y0:max(y1, y0 + 1), x0:max(x1, x0 + 1)] = True
                    if rng.random() < 0.5:                                     # not a rectangle: cut a corner
                        m[i, y0:(y0 + y1) // 2, x0:(x0 + x1) // 2] = False
                return m
            g["masks"], d["masks"] = raster(g["boxes"]), raster(d["boxes"])
        elif rng.random() < 0.3 and ng:
            g["area"] = rng.choice([100.0, 2000.0, 20000.0], ng)               # annotated areas (COCO's `area` field) reach the "large" range
        gts.append(g); dets.append(d)
    return gts, dets


@pytest.mark.parametrize("iou_type,n_sets", [("bbox", 160), ("segm", 60)])
def test_ap_equals_the_independent_brute_force_checker(iou_type, n_sets):
    from oracle.coco_ap_oracle import average_precision
    rng = np.random.default_rng(20260 + (iou_type == "segm"))
    seen = {"crowd": 0, "empty_img": 0, "capped": 0, "nan": 0}
    for s in range(n_sets):
        n_cls = int(rng.integers(1, 4))
        gts, dets = _random_scene(rng, int(rng.integers(1, 9)), n_cls, iou_type == "segm")
        max_dets = int(rng.choice([1, 3, 100]))
        seen["crowd"] += any(g["crowd"].any() for g in gts)
        seen["empty_img"] += any(len(g["classes"]) == 0 for g in gts)
        seen["capped"] += any(len(d["classes"]) > max_dets for d in dets)
        got = evaluate(gts, dets, n_cls, iou_type, max_dets)
        want = average_precision(gts, dets, n_cls, iou_type, max_dets)
        assert set(got) == set(want)
        for k in want:
            if np.isnan(want[k]):
                seen["nan"] += 1
                assert np.isnan(got[k]), (s, k, got[k])
            else:
                assert got[k] == pytest.approx(want[k], abs=1e-9), (s, k, got[k], want[k])
    assert min(seen.values()) > 0, seen          # the random sets did exercise crowd regions, empty images, the max_dets cap and undefined ranges
