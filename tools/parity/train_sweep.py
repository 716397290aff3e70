#!/usr/bin/env python3
"""Find solver settings under which `synthetic.train_trained_like` converges (GPU box): a few (lr, steps) arms, each reporting the
loss curve and how the resulting detector behaves on training-pool and fresh scenes (detections with score >= 0.5 per tile,
ground-truth boxes recalled at IoU >= 0.5/0.75, spread of the top scores)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arms", nargs="+", default=["0.005:300", "0.01:300", "0.02:300"])
    ap.add_argument("--batch", type=int, default=4)
    args = ap.parse_args()
    from proj_roadsurf_amd.engine import Engine
    from proj_roadsurf_amd.spec import EngineSpec
    from proj_roadsurf_amd.synthetic import synthetic_scenes, train_trained_like
    from tests.util import box_iou

    spec = EngineSpec(num_classes=2)
    log = lambda s: print(s, file=sys.stderr, flush=True)
    out = []
    for arm in args.arms:
        lr, steps = arm.split(":")
        lr, steps = float(lr), int(steps)
        t0 = time.time()
        try:
            W, curve = train_trained_like(spec, 512, steps=steps, batch=args.batch, lr=lr, log=log)
        except RuntimeError as e:
            rec = {"lr": lr, "steps": steps, "diverged": str(e)[:200]}
            print(json.dumps(rec), flush=True)
            out.append(rec)
            continue
        rec = {"lr": lr, "steps": steps, "train_s": round(time.time() - t0, 1), "loss_first": round(curve[0], 3),
               "loss_mean_last20": round(float(np.mean(curve[-20:])), 3), "curve_every_25": [round(c, 3) for c in curve[::25]]}
        eng = Engine(spec, W, (512, 512, 3), max_batch=8)
        for name, seed in (("pool", 1), ("fresh", 987654)):
            tiles, gtb, gtc, _ = synthetic_scenes(8, 512, 512, 3, seed=seed)
            dets = eng.infer(tiles)
            n50, rec50, rec75, cls_ok, tops = [], [], [], [], []
            for d, b, c in zip(dets, gtb, gtc):
                hi = d.scores >= 0.5
                n50.append(int(hi.sum()))
                iou = box_iou(b, d.pred_boxes[hi]) if hi.any() else np.zeros((len(b), 0))
                best = iou.max(1) if iou.shape[1] else np.zeros(len(b))
                rec50.append(float((best >= 0.5).mean())); rec75.append(float((best >= 0.75).mean()))
                if iou.shape[1]:
                    j = iou.argmax(1)
                    cls_ok.append(float((d.pred_classes[hi][j][best >= 0.5] == c[best >= 0.5]).mean()) if (best >= 0.5).any() else 0.0)
                tops.append([round(float(s), 3) for s in d.scores[:6]])
            rec[name] = {"dets_ge_0.5_per_tile": n50, "gt_per_tile": [len(b) for b in gtb], "recall@0.5": round(float(np.mean(rec50)), 3),
                         "recall@0.75": round(float(np.mean(rec75)), 3), "class_acc": round(float(np.mean(cls_ok)), 3) if cls_ok else None,
                         "top_scores_tile0": tops[0], "n_dets_total": [len(d) for d in dets]}
        eng.close()
        print(json.dumps(rec), flush=True)
        out.append(rec)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "parity"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity", "train_sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
