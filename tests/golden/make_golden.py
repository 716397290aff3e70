#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_small.npz.

No reference code can be imported in this environment (detectron2/torchvision absent, SURVEY.md
§8c), so these vectors are produced by THIS repo's CPU oracle (oracle/maskrcnn_oracle.py) with
seeded synthetic weights; they pin oracle regressions and the HIP engine against the oracle, not
the oracle against detectron2 ("parity unpinned").

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.maskrcnn_oracle import OracleModel  # noqa: E402
from proj_roadsurf_amd.spec import EngineSpec  # noqa: E402
from proj_roadsurf_amd.weights import synthetic_weights  # noqa: E402
from tests.util import synthetic_tiles  # noqa: E402

SPEC_KW = dict(num_classes=2, min_size_test=160, max_size_test=266, rpn_pre_nms_topk_test=150, rpn_post_nms_topk_test=150,
               detections_per_image=30)


def main():
    spec = EngineSpec(**SPEC_KW)
    W = synthetic_weights(spec, seed=0)
    tiles = synthetic_tiles(2, 128, 128, 3, seed=2024)
    res = OracleModel(spec, W)([tiles[0], tiles[1]], keep=True)
    out = {"tiles": tiles}
    for i, r in enumerate(res):
        it = r["inter"]
        out[f"boxes{i}"] = r["boxes"].numpy()
        out[f"scores{i}"] = r["scores"].numpy()
        out[f"classes{i}"] = r["classes"].numpy().astype(np.int32)
        out[f"masks{i}"] = np.packbits(r["masks"].numpy(), axis=-1, bitorder="little")
        out[f"mask_probs{i}"] = r["mask_probs"].numpy().astype(np.float16)
        out[f"proposals{i}"] = it["proposals"]["boxes"].numpy()
        out[f"proposal_logits{i}"] = it["proposals"]["logits"].numpy()
        out[f"net_input{i}"] = it["net_input"].numpy().astype(np.float16)
        for k in ("p2", "p5", "res4"):
            f = it["feats"][k].numpy()
            out[f"{k}_sample{i}"] = f[::16, ::3, ::3].astype(np.float32)      # strided sample keeps the file small
            out[f"{k}_norm{i}"] = np.float32(np.linalg.norm(f))
    p = os.path.join(ROOT, "tests", "golden", "oracle_small.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
