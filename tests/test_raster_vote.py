"""Raster voting (proj_roadsurf_amd/raster_vote.py, csrc/raster_vote.hip): host restatement against closed-form
cases of R:scripts/road_segmentation/determine_class.py:97-190, device kernel against its numpy statement."""
import numpy as np
import pytest

from oracle.host_tail_oracle import overlap_counts
from proj_roadsurf_amd import raster_vote as RV


def _pack(m):
    return np.packbits(m, axis=-1, bitorder="little")


def test_weighted_scores_and_vote_closed_form():
    h = w = 32
    road_a = np.zeros((h, w), bool); road_a[4:8, 0:32] = True            # 128 px
    road_b = np.zeros((h, w), bool); road_b[20:24, 0:16] = True           # 64 px
    road_c = np.zeros((h, w), bool); road_c[28:30, 0:8] = True            # 16 px, no detection on it
    det0 = np.zeros((h, w), bool); det0[0:12, 0:16] = True                # covers half of road a            -> 0.5, artificial
    det1 = np.zeros((h, w), bool); det1[0:12, 16:24] = True               # a quarter of road a              -> 0.25, natural
    det2 = np.zeros((h, w), bool); det2[18:26, 0:16] = True               # all of road b                    -> 1.0, natural
    det3 = np.zeros((h, w), bool); det3[4:8, 30:32] = True                # 8 px of road a = 0.0625 -> 0.06  -> kept (> 0.05), artificial
    det4 = np.zeros((h, w), bool); det4[4:5, 24:29] = True                # 5 px = 0.039 -> 0.04             -> dropped
    dets = _pack(np.stack([det0, det1, det2, det3, det4]))
    labs = _pack(np.stack([road_a, road_b, road_c]))
    inter, area = overlap_counts(dets, labs)
    assert area.tolist() == [128, 64, 16]
    assert inter[0].tolist() == [64, 32, 0, 8, 5] and inter[1].tolist() == [0, 0, 64, 0, 0] and not inter[2].any()
    scores = np.array([0.9, 0.8, 0.7, 0.5, 0.99], np.float32)
    classes = np.array([0, 1, 1, 0, 0])
    rows = RV.weighted_scores(inter, area, scores, classes, ["a", "b", "c"])
    assert [(r["OBJECTID"], r["det"], r["area_pred_in_label"]) for r in rows] == [("a", 0, 0.5), ("a", 1, 0.25), ("a", 3, 0.06), ("b", 2, 1.0)]
    assert abs(rows[0]["weighted_score"] - 0.45) < 1e-7
    votes = {v["road_id"]: v for v in RV.determine_detected_class(rows, ["a", "b", "c"])}
    # road a: artificial index = (0.5*0.9 + 0.06*0.5) / 0.56, natural index = 0.8
    art = (0.45 + 0.03) / 0.56
    assert votes["a"]["cover_type"] == "artificial" and votes["a"]["art_score"] == round(art, 3) and abs(votes["a"]["nat_score"] - 0.8) < 1e-6
    assert abs(votes["a"]["diff_score"] - (art - 0.8)) < 1e-6
    assert votes["b"]["cover_type"] == "natural" and abs(votes["b"]["nat_score"] - 0.7) < 1e-6 and votes["b"]["art_score"] == 0
    assert votes["c"] == {"road_id": "c", "cover_type": "undetected", "nat_score": 0, "art_score": 0, "diff_score": 0}
    # the score threshold of determine_detected_class removes det1: road a then has no natural vote
    v2 = {v["road_id"]: v for v in RV.determine_detected_class(rows, ["a"], threshold=0.85)}
    assert v2["a"]["cover_type"] == "artificial" and v2["a"]["nat_score"] == 0
    # a tie is "undetermined"
    tie = [dict(rows[0], det_class_name="natural", OBJECTID="t"), dict(rows[0], OBJECTID="t")]
    assert RV.determine_detected_class(tie, ["t"])[0]["cover_type"] == "undetermined"


@pytest.mark.gpu
def test_mask_overlap_kernel_equals_numpy(gpu_required):
    import torch

    from proj_roadsurf_amd.engine import load_library
    lib = load_library()
    rng = np.random.default_rng(5)
    h, w = 96, 200
    dets = rng.random((7, h, w)) > 0.6
    labs = rng.random((5, h, w)) > 0.8
    labs[3] = False
    dp, lp = _pack(dets), _pack(labs)
    want_i, want_a = overlap_counts(dp, lp)
    d_dev = torch.from_numpy(dp).cuda()
    got_i, got_a = RV.overlap_counts_device(lib, d_dev.data_ptr(), 7, lp, h, w)
    assert np.array_equal(got_i, want_i) and np.array_equal(got_a, want_a)


@pytest.mark.gpu
def test_engine_label_overlap_on_the_last_forward(gpu_required):
    """The vote straight from the engine's device-resident masks: counts for tile 1 of a batch == numpy on the masks fetched to the host."""
    import ctypes as C

    import torch

    from proj_roadsurf_amd.engine import Engine
    from proj_roadsurf_amd.spec import EngineSpec
    from proj_roadsurf_amd.synthetic import synthetic_tiles
    from proj_roadsurf_amd.weights import synthetic_weights
    spec = EngineSpec(num_classes=2, min_size_test=192, max_size_test=320, rpn_pre_nms_topk_test=200, rpn_post_nms_topk_test=200)
    eng = Engine(spec, synthetic_weights(spec, 0), (128, 128, 3), max_batch=2)
    try:
        dets = eng.infer(synthetic_tiles(2, 128, 128, 3, seed=5))
        roads = [[np.array([10.0, 20.0, 120.0, 20.0, 120.0, 30.0, 10.0, 30.0])], [np.array([60.0, 0.0, 70.0, 0.0, 70.0, 128.0, 60.0, 128.0])]]
        lp = RV.label_rasters(roads, 128, 128)
        lab = torch.from_numpy(lp).cuda()
        D = spec.detections_per_image
        inter = torch.zeros((2, D), dtype=torch.int32, device="cuda")
        area = torch.zeros((2,), dtype=torch.int32, device="cuda")
        eng.lib.rs_engine_label_overlap.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        assert eng.lib.rs_engine_label_overlap(eng._h, 1, C.c_void_p(lab.data_ptr()), 2, C.c_void_p(inter.data_ptr()), C.c_void_p(area.data_ptr())) == 0
        eng.sync()
        n = len(dets[1])
        want_i, want_a = overlap_counts(dets[1]._packed, lp)
        assert n > 0 and np.array_equal(inter.cpu().numpy()[:, :n], want_i) and np.array_equal(area.cpu().numpy(), want_a)
        assert want_a.tolist() == [110 * 10, 10 * 128]
        rows = RV.weighted_scores(want_i, want_a, dets[1].scores, dets[1].pred_classes, ["r1", "r2"])
        votes = RV.determine_detected_class(rows, ["r1", "r2"])
        assert {v["road_id"] for v in votes} == {"r1", "r2"} and all(v["cover_type"] in ("artificial", "natural", "undetermined", "undetected") for v in votes)
    finally:
        eng.close()
