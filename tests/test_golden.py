"""Golden vectors (tests/golden/oracle_small.npz, written by tests/golden/make_golden.py from the CPU
oracle -- the reference holds no fixtures for this path):

* CPU: the oracle still reproduces them (regression pin of the restatement);
* GPU: the HIP engine, through the C ABI, reproduces them within the fp16 tolerance, without running the
  oracle at test time.
"""
import os

import numpy as np
import pytest

from proj_roadsurf_amd.spec import EngineSpec
from proj_roadsurf_amd.weights import synthetic_weights
from tests.golden.make_golden import SPEC_KW
from tests.util import match_detections

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_small.npz")


def _unpack(m, w):
    return np.unpackbits(m, axis=-1, bitorder="little")[..., :w].astype(bool)


def test_oracle_reproduces_golden():
    from oracle.maskrcnn_oracle import OracleModel
    g = np.load(GOLD)
    spec = EngineSpec(**SPEC_KW)
    res = OracleModel(spec, synthetic_weights(spec, 0))([g["tiles"][0], g["tiles"][1]], keep=True)
    for i, r in enumerate(res):
        assert r["boxes"].shape[0] == g[f"boxes{i}"].shape[0]
        assert np.array_equal(r["classes"].numpy(), g[f"classes{i}"])
        # other BLAS/oneDNN builds may differ in the last bits of the convolutions
        assert np.abs(r["boxes"].numpy() - g[f"boxes{i}"]).max() <= 5e-2
        assert np.abs(r["scores"].numpy() - g[f"scores{i}"]).max() <= 1e-4
        assert np.array_equal(r["inter"]["net_input"].numpy().astype(np.float16), g[f"net_input{i}"])
        m = _unpack(g[f"masks{i}"], 128)
        assert np.logical_xor(r["masks"].numpy(), m).sum() <= 1e-3 * m.size


@pytest.mark.gpu
def test_engine_reproduces_golden(gpu_required):
    from proj_roadsurf_amd.engine import Engine
    g = np.load(GOLD)
    spec = EngineSpec(**SPEC_KW)
    eng = Engine(spec, synthetic_weights(spec, 0), (128, 128, 3), max_batch=2)
    try:
        dets = eng.infer(g["tiles"], want_probs=True)
        x = eng.tensor("net_input", n=2)
        for i in range(2):
            assert np.array_equal(x[i, :, :, :3].transpose(2, 0, 1), g[f"net_input{i}"]), "pre-processing must be bit-exact"
            for k in ("p2", "p5", "res4"):
                f = eng.tensor(k, n=2)[i].astype(np.float32).transpose(2, 0, 1)
                s = f[::16, ::3, ::3]
                rel = np.linalg.norm(s - g[f"{k}_sample{i}"]) / np.linalg.norm(g[f"{k}_sample{i}"])
                assert rel <= 1.5e-2, (k, rel)
            ref = {"boxes": g[f"boxes{i}"], "scores": g[f"scores{i}"], "classes": g[f"classes{i}"], "masks": _unpack(g[f"masks{i}"], 128)}
            got = {"boxes": dets[i].pred_boxes, "scores": dets[i].scores, "classes": dets[i].pred_classes, "masks": dets[i].pred_masks}
            fw, bw = match_detections(ref, got), match_detections(got, ref)
            print("golden", i, fw, bw)
            # fp16 mode on the random-weight workload: see the note at the top of tests/test_gpu_engine.py
            assert fw["frac_matched"] >= 0.85 and bw["frac_matched"] >= 0.85, (fw, bw)
            assert fw["max_dscore"] <= 0.02 and fw["agg_mask_iou"] >= 0.95, fw
    finally:
        eng.close()
    # reference-precision mode: the fixture's detections to the stated tolerance (>= 98 % both ways)
    for precision in ("fp32", "split"):      # fp32 matrix cores; hi + lo fp16 operand planes on the fp16 matrix cores
        spec32 = EngineSpec(**dict(SPEC_KW, precision=precision))
        eng = Engine(spec32, synthetic_weights(spec32, 0), (128, 128, 3), max_batch=2)
        try:
            dets = eng.infer(g["tiles"])
            for i in range(2):
                ref = {"boxes": g[f"boxes{i}"], "scores": g[f"scores{i}"], "classes": g[f"classes{i}"], "masks": _unpack(g[f"masks{i}"], 128)}
                got = {"boxes": dets[i].pred_boxes, "scores": dets[i].scores, "classes": dets[i].pred_classes, "masks": dets[i].pred_masks}
                fw, bw = match_detections(ref, got), match_detections(got, ref)
                print("golden", precision, i, fw, bw)
                assert fw["frac_matched"] >= 0.98 and bw["frac_matched"] >= 0.98, (fw, bw)
                assert fw["max_dscore"] <= 1e-4 and fw["agg_mask_iou"] >= 0.995, fw
        finally:
            eng.close()
