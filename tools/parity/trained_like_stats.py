#!/usr/bin/env python3
"""How robustly does the fp16 engine meet the SURVEY 8d criterion (>= 98 % of detections with score >= 0.1 matched both ways at
IoU >= 0.95, |dscore| <= 0.02, mask IoU >= 0.95) on the trained-like workload?  Several training seeds / lengths, N fresh
scenes each, aggregated over the scenes.  (GPU box; the oracle runs on the host cores.)"""
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
    ap.add_argument("--arms", nargs="+", default=["0:300", "1:300", "0:600", "1:600"], help="seed:steps")
    ap.add_argument("--tiles", type=int, default=12)
    args = ap.parse_args()
    from oracle.maskrcnn_oracle import OracleModel
    from proj_roadsurf_amd.engine import Engine
    from proj_roadsurf_amd.spec import EngineSpec
    from proj_roadsurf_amd.synthetic import synthetic_scenes, train_trained_like
    from tests.util import match_detections

    spec = EngineSpec(num_classes=2)
    out = []
    for arm in args.arms:
        seed, steps = (int(x) for x in arm.split(":"))
        t0 = time.time()
        W, curve = train_trained_like(spec, 512, steps=steps, seed=seed)
        tt = time.time() - t0
        tiles, gtb, gtc, _ = synthetic_scenes(args.tiles, 512, 512, 3, seed=987654 + seed)
        eng = Engine(spec, W, (512, 512, 3), max_batch=args.tiles)
        dets = eng.infer(tiles)
        eng.close()
        m = OracleModel(spec, W)
        ref = m([tiles[i] for i in range(args.tiles)])
        tot = {"fw_n": 0, "fw_m": 0, "bw_n": 0, "bw_m": 0}
        worst_ds, worst_miou, worst_agg = 0.0, 1.0, 1.0
        per_tile = []
        for i in range(args.tiles):
            r = {"boxes": ref[i]["boxes"].numpy(), "scores": ref[i]["scores"].numpy(), "classes": ref[i]["classes"].numpy(), "masks": ref[i]["masks"].numpy()}
            g = {"boxes": dets[i].pred_boxes, "scores": dets[i].scores, "classes": dets[i].pred_classes, "masks": dets[i].pred_masks}
            fw, bw = match_detections(r, g), match_detections(g, r)
            tot["fw_n"] += fw["n_ref"]; tot["fw_m"] += round(fw["frac_matched"] * fw["n_ref"])
            tot["bw_n"] += bw["n_ref"]; tot["bw_m"] += round(bw["frac_matched"] * bw["n_ref"])
            worst_ds = max(worst_ds, fw["max_dscore"]); worst_miou = min(worst_miou, float(fw["min_mask_iou"])); worst_agg = min(worst_agg, float(fw["agg_mask_iou"]))
            per_tile.append((fw["n_ref"], round(fw["frac_matched"], 3), round(bw["frac_matched"], 3)))
        rec = {"seed": seed, "steps": steps, "train_s": round(tt, 1), "loss_last20": round(float(np.mean(curve[-20:])), 3), **tot,
               "fw": round(tot["fw_m"] / max(tot["fw_n"], 1), 4), "bw": round(tot["bw_m"] / max(tot["bw_n"], 1), 4),
               "max_dscore": round(worst_ds, 5), "min_mask_iou": round(worst_miou, 4), "min_agg_mask_iou": round(worst_agg, 4), "per_tile": per_tile}
        print(json.dumps(rec), flush=True)
        out.append(rec)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "parity"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity", "trained_like_stats.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
