#!/usr/bin/env python3
"""How much feature noise does the matched-detections criterion (>= 98 % matched at IoU >= 0.95) tolerate?

CPU only.  Runs the oracle once in fp32, then re-runs everything downstream of the FPN maps (RPN head ... paste) on the
same maps perturbed by relative Gaussian noise of a given size, and matches the two detection sets both ways.  fp16
storage of every activation gives ~3e-3 relative noise on p2..p6 (tests/test_gpu_engine.py::test_backbone_features).

    python tools/parity/noise_sensitivity.py [--weights random|<npz>] [--tiles 2] [--noise 3e-3 1e-3 3e-4 1e-4]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def as_np(r):
    return {"boxes": r["boxes"].numpy(), "scores": r["scores"].numpy(), "classes": r["classes"].numpy(), "masks": r["masks"].numpy()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default="random")
    ap.add_argument("--tiles", type=int, default=2)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--noise", type=float, nargs="+", default=[3e-3, 1e-3, 3e-4, 1e-4])
    args = ap.parse_args()
    from oracle.maskrcnn_oracle import OracleModel, normalize_and_pad, predictor_preprocess
    from proj_roadsurf_amd.spec import EngineSpec
    from proj_roadsurf_amd.weights import synthetic_weights
    from tests.util import match_detections, synthetic_tiles

    spec = EngineSpec(num_classes=2)
    if args.weights == "random":
        W = synthetic_weights(spec, seed=0)
    else:
        W = dict(np.load(args.weights))
    m = OracleModel(spec, W)
    tiles = synthetic_tiles(args.tiles, args.tile, args.tile, 3, seed=args.seed)
    out = []
    for i in range(args.tiles):
        t0 = time.time()
        t, _ = predictor_preprocess(spec, tiles[i])
        x, sizes = normalize_and_pad(spec, [t])
        feats = m.backbone(x)
        ref = as_np(m.forward_features(feats, sizes, [(args.tile, args.tile)])[0])
        print(f"tile {i}: {len(ref['scores'])} detections, scores {ref['scores'][:3]} .. {ref['scores'][-3:]}  ({time.time() - t0:.1f} s)", flush=True)
        for s in args.noise:
            g = torch.Generator().manual_seed(1000 + i)
            noisy = {k: v + s * v.std() * torch.randn(v.shape, generator=g) if k.startswith("p") else v for k, v in feats.items()}
            got = as_np(m.forward_features(noisy, sizes, [(args.tile, args.tile)])[0])
            fw, bw = match_detections(ref, got), match_detections(got, ref)
            rec = {"tile": i, "noise": s, "fw": fw["frac_matched"], "bw": bw["frac_matched"], "n_ref": fw["n_ref"],
                   "max_dscore": fw["max_dscore"], "agg_mask_iou": fw["agg_mask_iou"]}
            print(json.dumps(rec), flush=True)
            out.append(rec)
    print(json.dumps({"weights": args.weights, "results": out}))


if __name__ == "__main__":
    main()
