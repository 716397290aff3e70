#!/usr/bin/env python3
"""Training-step rate and a short convergence run of the MI355X training engine on synthetic data.

    python tools/train_bench.py [--batch 8] [--tile 512] [--iters 30] [--overfit 0]

--overfit N: train N iterations on one fixed batch (reference YAML hyper-parameters, lr fixed at BASE_LR after a 10-iteration
warm-up) and print the loss curve -- a sanity check that the assembled forward/backward/SGD actually learns."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--overfit", type=int, default=0)
    ap.add_argument("--loss-scale", type=float, default=1024.0)
    ap.add_argument("--gt-per-image", type=int, default=0, help="0: 3-8 boxes per tile; N: exactly N small boxes (every gt box is "
                    "also a foreground RoI, so this sets the mask-branch load: host rasterisation + mask-head GEMMs)")
    args = ap.parse_args()
    from proj_roadsurf_amd.engine import Trainer
    from proj_roadsurf_amd.spec import EngineSpec
    from proj_roadsurf_amd.weights import synthetic_weights
    from proj_roadsurf_amd.synthetic import synthetic_tiles

    spec = EngineSpec(num_classes=2)
    W = synthetic_weights(spec, seed=0)
    B, T = args.batch, args.tile
    tiles = synthetic_tiles(B, T, T, 3, seed=4321)
    rng = np.random.default_rng(1)
    s = 800.0 / T
    boxes, classes, polys = [], [], []
    for i in range(B):
        k = args.gt_per_image or int(rng.integers(3, 9))
        xy = rng.uniform(20, T - 160, (k, 2))
        wh = rng.uniform(30, 150, (k, 2)) if not args.gt_per_image else rng.uniform(15, 60, (k, 2))
        b = np.concatenate([xy, xy + wh], 1) * s
        boxes.append(b.astype(np.float32))
        classes.append(rng.integers(0, 2, k))
        polys.append([[np.array([x0, y0, x1, y0, x1, y1, x0, y1])] for x0, y0, x1, y1 in b.tolist()])
    tr = Trainer(spec, W, (T, T, 3), batch=B, loss_scale=args.loss_scale)
    out = {"batch": B, "tile": T, "trainable_values_M": tr.param_count / 1e6, "gt_per_image": args.gt_per_image or "3-8"}
    if args.overfit:
        curve = []
        for it in range(args.overfit):
            l = tr.train_step(tiles, boxes, classes, polys, seed=it)
            lr = 0.01 * min(1.0, (it + 1) / 10.0)
            tr.apply_sgd(lr, 0.9, 1e-4)
            curve.append(round(sum(l.values()), 4))
            if it % 10 == 0 or it == args.overfit - 1:
                print(f"[overfit] iter {it:4d} total {sum(l.values()):.4f} " + " ".join(f"{k[5:]} {v:.4f}" for k, v in l.items()), file=sys.stderr, flush=True)
        out["overfit_total_loss_curve"] = curve
    else:
        for it in range(3):
            tr.train_step(tiles, boxes, classes, polys, seed=it)
            tr.apply_sgd(1e-5, 0.9, 1e-4)
        tr.sync()
        t0 = time.perf_counter()
        for it in range(args.iters):
            tr.train_step(tiles, boxes, classes, polys, seed=100 + it)
            tr.apply_sgd(1e-5, 0.9, 1e-4)
        tr.sync()
        dt = time.perf_counter() - t0
        out.update({"iters": args.iters, "s_per_iter": dt / args.iters, "images_per_s": B * args.iters / dt,
                    "mask_entries_last_step": int(tr.tensor("mask_total")[0]) if spec.mask_on else 0})
    print(json.dumps(out))
    tr.close()


if __name__ == "__main__":
    main()
