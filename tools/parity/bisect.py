#!/usr/bin/env python3
"""Where do the fp16 engine's detections start to differ from the fp32 oracle's?  (GPU box.)

For random (``weights.synthetic_weights``) and trained-like (``synthetic.train_trained_like``) weights and a few tiles:
  A  oracle, fp32 end to end
  B  oracle downstream of the ENGINE's fp16 FPN maps p2..p6 (everything after the backbone in fp32)
  E  engine, fp16 production mode
and the oracle re-run on its own maps with relative Gaussian noise (how much noise the >= 98 % criterion tolerates).
Prints one JSON line per comparison and a summary; writes gpurun_out/parity/bisect.json.

    python tools/parity/bisect.py [--steps 300] [--tiles 3] [--which random trained]
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
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--tiles", type=int, default=3)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--which", nargs="+", default=["random", "trained"])
    ap.add_argument("--noise", type=float, nargs="*", default=[3e-3, 1e-3])
    args = ap.parse_args()
    from oracle.maskrcnn_oracle import OracleModel, normalize_and_pad, predictor_preprocess
    from proj_roadsurf_amd.engine import Engine
    from proj_roadsurf_amd.spec import EngineSpec
    from proj_roadsurf_amd.synthetic import synthetic_scenes, synthetic_tiles, train_trained_like
    from proj_roadsurf_amd.weights import synthetic_weights
    from tests.util import match_detections

    spec = EngineSpec(num_classes=2)
    T = args.tile
    results = []
    log = lambda s: print(s, file=sys.stderr, flush=True)
    for which in args.which:
        if which == "random":
            W = synthetic_weights(spec, seed=0)
            tiles = synthetic_tiles(args.tiles, T, T, 3, seed=1234)
        else:
            t0 = time.time()
            W, curve = train_trained_like(spec, T, steps=args.steps, batch=args.batch, log=log)
            log(f"trained {args.steps} steps in {time.time() - t0:.1f} s, loss {curve[0]:.3f} -> {np.mean(curve[-10:]):.3f}")
            tiles, gtb, gtc, _ = synthetic_scenes(args.tiles, T, T, 3, seed=987654)
        eng = Engine(spec, W, (T, T, 3), max_batch=args.tiles)
        dets = eng.infer(tiles)
        P = {f"p{l}": torch.from_numpy(eng.tensor(f"p{l}", n=args.tiles).astype(np.float32)).permute(0, 3, 1, 2) for l in (2, 3, 4, 5, 6)}
        eng.close()
        m = OracleModel(spec, W)
        for i in range(args.tiles):
            t, _ = predictor_preprocess(spec, tiles[i])
            x, sizes = normalize_and_pad(spec, [t])
            feats = m.backbone(x)
            A = as_np(m.forward_features(feats, sizes, [(T, T)])[0])
            fB = dict(feats)
            rel = {}
            for k in P:
                rel[k] = float((P[k][i:i + 1] - feats[k]).norm() / feats[k].norm())
                fB[k] = P[k][i:i + 1]
            B = as_np(m.forward_features(fB, sizes, [(T, T)])[0])
            E = {"boxes": dets[i].pred_boxes, "scores": dets[i].scores, "classes": dets[i].pred_classes, "masks": dets[i].pred_masks}
            n10 = int((A["scores"] >= 0.1).sum())
            rec = {"weights": which, "tile": i, "n_det": len(A["scores"]), "n_score_ge_0.1": n10,
                   "scores_head": [round(float(s), 4) for s in A["scores"][:5]], "scores_tail": [round(float(s), 4) for s in A["scores"][-3:]],
                   "fpn_rel_err": {k: round(v, 5) for k, v in rel.items()}}
            if which != "random":
                rec["gt"] = int(len(gtb[i]))
            for name, X, Y in (("A_vs_B", A, B), ("A_vs_E", A, E), ("B_vs_E", B, E)):
                fw, bw = match_detections(X, Y), match_detections(Y, X)
                rec[name] = {"fw": round(fw["frac_matched"], 4), "bw": round(bw["frac_matched"], 4), "n_fw": fw["n_ref"], "n_bw": bw["n_ref"],
                             "max_dscore": round(fw["max_dscore"], 5), "agg_mask_iou": round(float(fw["agg_mask_iou"]), 4),
                             "min_mask_iou": round(float(fw["min_mask_iou"]), 4), "max_dbox": round(fw["max_dbox"], 3)}
            for s in args.noise:
                g = torch.Generator().manual_seed(1000 + i)
                noisy = {k: v + s * v.std() * torch.randn(v.shape, generator=g) if k.startswith("p") else v for k, v in feats.items()}
                N = as_np(m.forward_features(noisy, sizes, [(T, T)])[0])
                fw, bw = match_detections(A, N), match_detections(N, A)
                rec[f"noise_{s:g}"] = {"fw": round(fw["frac_matched"], 4), "bw": round(bw["frac_matched"], 4)}
            print(json.dumps(rec), flush=True)
            results.append(rec)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "parity"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity", "bisect.json"), "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
