"""CPU checks of the training host code (proj_roadsurf_amd/train_model.py, weights.train_tensors / master_to_d2): no GPU."""
import json
import os

import numpy as np
import pytest

from proj_roadsurf_amd import train_model as TM
from proj_roadsurf_amd import weights as Wt
from proj_roadsurf_amd.spec import EngineSpec

REF_YAML = "/root/reference/config/detectron2_config_3bands.yaml"


@pytest.mark.skipif(not os.path.exists(REF_YAML), reason="reference not mounted (GPU box)")
def test_solver_values_come_from_the_reference_yaml():
    from oracle import train_oracle as T
    sv = TM.load_solver(REF_YAML)
    ts = T.TrainSpec()
    assert (sv["base_lr"], sv["momentum"], sv["weight_decay"], sv["gamma"]) == (ts.base_lr, ts.momentum, ts.weight_decay, ts.gamma)
    assert sv["steps"] == ts.steps and sv["max_iter"] == ts.max_iter and sv["warmup_iters"] == ts.warmup_iters
    assert sv["ims_per_batch"] == 8 and sv["checkpoint_period"] == 500
    assert (sv["rpn_batch"], sv["rpn_pos"], sv["roi_batch"], sv["roi_pos"]) == (256, 0.5, 1024, 0.25)
    for it in (0, 1, 100, 199, 200, 2999, 3000, 5499, 5500, 11999):
        assert TM.lr_at(sv, it) == pytest.approx(T.lr_at(ts, it))


def test_master_layouts_round_trip_to_detectron2_keys():
    """train_tensors (detectron2 tensors -> engine master layouts) followed by master_to_d2 returns every trainable tensor
    unchanged -- conv OHWI, fused RPN heads, fc1's (h,w,c) K order, fused predictor, transposed deconv, padded mask predictor --
    and leaves the frozen stem / res2 / FrozenBN tensors alone."""
    spec = EngineSpec(num_classes=2)
    W = Wt.synthetic_weights(spec, 0)
    T = Wt.train_tensors(spec, W)
    raw = Wt.engine_tensors(spec, W, w_dtype=np.float32, fold_bn=False)

    def fetch(name):
        layer, kind = name[2:].rsplit(".", 1)
        if kind == "w":
            return T[layer + (".m32T" if layer.endswith("mask_head.deconv") else ".m32")]
        if layer.endswith("mask_head.deconv"):
            return T[layer + ".b256"]
        if layer.endswith("predictor16"):
            return T[layer + ".b"]
        return raw[layer + ".b"]
    base = {k: (np.zeros_like(v) if ("res3" in k or "fpn" in k or "roi_heads" in k or "rpn_head" in k) and ".norm." not in k else v) for k, v in W.items()}
    back = Wt.master_to_d2(spec, base, fetch)
    assert set(back) == set(W)
    for k in W:
        assert back[k].shape == W[k].shape and np.array_equal(back[k], W[k].astype(np.float32)), k
    assert sum(T[k].size for k in T if k.endswith(".m32")) == pytest.approx(43.7e6, rel=0.01)       # 43.7 M trainable


def test_coco_mapper_flip_scale_and_sampler(tmp_path):
    coco = {"images": [{"id": 1, "file_name": "a.tif", "width": 100, "height": 50}, {"id": 2, "file_name": "b.tif", "width": 100, "height": 50},
                       {"id": 3, "file_name": "empty.tif", "width": 100, "height": 50}],
            "categories": [{"id": 5, "name": "x"}, {"id": 9, "name": "y"}],
            "annotations": [{"id": 1, "image_id": 1, "category_id": 9, "bbox": [10, 5, 30, 20], "segmentation": [[10, 5, 40, 5, 40, 25, 10, 25]], "iscrowd": 0},
                            {"id": 2, "image_id": 1, "category_id": 5, "bbox": [60, 10, 10, 10], "segmentation": [[60, 10, 70, 10, 70, 20]], "iscrowd": 1},
                            {"id": 3, "image_id": 2, "category_id": 5, "bbox": [0, 0, 100, 50], "segmentation": [[0, 0, 100, 0, 100, 50, 0, 50]], "iscrowd": 0}]}
    p = tmp_path / "c.json"
    json.dump(coco, open(p, "w"))
    recs, cats = TM.load_coco_training_set(str(p))
    assert cats == [5, 9] and [r["file_name"] for r in recs] == ["a.tif", "b.tif"]            # empty image filtered, crowd dropped
    assert recs[0]["classes"].tolist() == [1] and recs[0]["boxes"].tolist() == [[10, 5, 40, 25]]
    tile = np.arange(50 * 100 * 3, dtype=np.uint8).reshape(50, 100, 3)
    t, b, c, polys = TM.map_record(recs[0], tile, (100, 200), flip=True)                        # x2 scale, horizontal flip
    assert np.array_equal(t, tile[:, ::-1]) and c.tolist() == [1]
    assert np.allclose(b, [[(100 - 40) * 2, 10, (100 - 10) * 2, 50]])
    assert np.allclose(polys[0][0][0::2], (100 - np.array([10, 40, 40, 10])) * 2) and np.allclose(polys[0][0][1::2], np.array([5, 5, 25, 25]) * 2)
    a, b2 = TM.training_sampler(10, 3, 0, 2), TM.training_sampler(10, 3, 1, 2)
    s0, s1 = [next(a) for _ in range(10)], [next(b2) for _ in range(10)]
    assert sorted(s0[:5] + s1[:5]) == list(range(10)) and sorted(s0[5:] + s1[5:]) == list(range(10))


def test_solver_validation_accepts_the_reference_yaml_and_rejects_unsupported_values(tmp_path):
    """R:config/detectron2_config_3bands.yaml's solver / sampler values pass; values the training engine does not implement are
    rejected up front (never clamped or ignored), as are images with more ground-truth boxes than the engine holds."""
    import os

    import yaml

    from proj_roadsurf_amd import train_model as TM
    ref = "/root/reference/config/detectron2_config_3bands.yaml"
    if os.path.exists(ref):
        sv = TM.load_solver(ref)
        TM.validate_solver(sv)
        assert sv["pre_nms_topk_train"] == 2000 and sv["post_nms_topk_train"] == 1000 and sv["roi_batch"] == 1024 and sv["max_size_train"] == 1333
    base = {"INPUT": {"MAX_SIZE_TEST": 320}, "SOLVER": {"BASE_LR": 0.01}, "MODEL": {"ROI_HEADS": {"BATCH_SIZE_PER_IMAGE": 64}}}
    p = tmp_path / "ok.yaml"
    yaml.safe_dump(base, open(p, "w"))
    TM.validate_solver(TM.load_solver(str(p)))            # silent on MAX_SIZE_TRAIN: trains at the test maximum
    bad = [({"SOLVER": {"WARMUP_METHOD": "constant"}}, "WARMUP_METHOD"), ({"SOLVER": {"BIAS_LR_FACTOR": 2.0}}, "BIAS_LR_FACTOR"),
           ({"SOLVER": {"CLIP_GRADIENTS": {"ENABLED": True}}}, "CLIP_GRADIENTS"), ({"INPUT": {"MAX_SIZE_TRAIN": 1000, "MAX_SIZE_TEST": 1333}}, "MAX_SIZE_TRAIN"),
           ({"MODEL": {"ROI_HEADS": {"BATCH_SIZE_PER_IMAGE": 2048}}}, "BATCH_SIZE_PER_IMAGE"),
           ({"MODEL": {"ROI_HEADS": {"BATCH_SIZE_PER_IMAGE": 1024, "POSITIVE_FRACTION": 0.5}}}, "mask-head entries"),
           ({"MODEL": {"RPN": {"PRE_NMS_TOPK_TRAIN": 6000}}}, "PRE_NMS_TOPK_TRAIN"), ({"SOLVER": {"NESTEROV": True}}, "NESTEROV")]
    for cfg, word in bad:
        q = tmp_path / "bad.yaml"
        yaml.safe_dump(cfg, open(q, "w"))
        with pytest.raises(SystemExit) as e:
            TM.validate_solver(TM.load_solver(str(q)))
        assert word in str(e.value), (word, str(e.value))
    recs = [{"file_name": "a.tif", "classes": np.zeros(5)}, {"file_name": "b.tif", "classes": np.zeros(129)}]
    with pytest.raises(SystemExit) as e:
        TM.check_gt_capacity(recs, "training set")
    assert "b.tif" in str(e.value) and "129" in str(e.value)
    TM.check_gt_capacity(recs[:1], "training set")
