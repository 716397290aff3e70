#!/usr/bin/env python3
"""Which stage AFTER the FPN turns "oracle == oracle-on-the-engine's-maps" into missed detections?  (GPU box.)

Trained-like detectors (``synthetic.train_trained_like``, several seeds), N fresh scenes each.  A (oracle) and E (engine) are
compared on EVERY scene (SURVEY 8d matching: same class, IoU >= 0.95, reference score >= 0.1; pooled counts + Wilson lower
bounds); on the scenes where they differ the chain of cuts below is evaluated to see at which stage the difference appears:

  A  oracle, fp32 end to end
  B  oracle downstream of the ENGINE's FPN maps p2..p6
  C  oracle downstream of the ENGINE's RPN head outputs (objectness + deltas), pooling from the engine's maps
  D  oracle downstream of the ENGINE's proposals and pooled 7x7 box features (fc1, fc2, predictors in fp32)
  F  oracle downstream of the ENGINE's class logits / box deltas (softmax, decode, NMS, mask head in fp32)
  E  engine

The step X -> Y of the chain A B C D F E that loses detections is the stage whose arithmetic sits between the two cuts:
A->B backbone + FPN, B->C RPN 3x3 + heads, C->D top-k / NMS / RoIAlign, D->F fc1 / fc2 / predictors, F->E box glue + mask head.
Writes gpurun_out/parity/bisect_stages.json; prints one JSON line per seed and the pooled table.

    python tools/parity/bisect_stages.py [--seeds 0 1 2 3] [--tiles 48] [--steps 600]
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def as_np(r):
    return {"boxes": r["boxes"].numpy(), "scores": r["scores"].numpy(), "classes": r["classes"].numpy(), "masks": r["masks"].numpy()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2, 3])
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--tiles", type=int, default=48)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--objects", type=int, nargs=2, default=[3, 8], help="objects per evaluation scene (min max)")
    ap.add_argument("--out", default="bisect_stages.json")
    args = ap.parse_args()
    from oracle import maskrcnn_oracle as O
    from proj_roadsurf_amd.engine import Engine
    from proj_roadsurf_amd.spec import EngineSpec
    from proj_roadsurf_amd.synthetic import synthetic_scenes, train_trained_like
    from proj_roadsurf_amd.matching import match_detections, wilson_lower

    # a one-GPU box's CPU share is 16 cores whatever the host shows: torch's default (one thread per visible core) oversubscribes
    # them and the oracle runs ~5x slower
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 16)))
    spec = EngineSpec(num_classes=2)
    K = spec.num_classes
    A_ = spec.num_anchors
    T = 512
    log = lambda s: print(s, file=sys.stderr, flush=True)
    chain = ["A", "B", "C", "D", "F", "E"]
    pairs = [(chain[i], chain[i + 1]) for i in range(len(chain) - 1)] + [("A", x) for x in chain[2:]]
    pooled = {f"{x}_vs_{y}": {"fw_n": 0, "fw_m": 0, "bw_n": 0, "bw_m": 0} for x, y in pairs}
    per_seed, misses = [], []

    def tail(m, fn, pb, size, pooled_feat=None, clsreg=None):
        """forward_features' per-image body (oracle/maskrcnn_oracle.py OracleModel.forward_features) with overrides"""
        scales = [1.0 / s for s in spec.fpn_strides[: len(fn)]]
        if clsreg is None:
            pf = pooled_feat if pooled_feat is not None else O.roi_pooler(fn, scales, [pb], spec.box_pooler_resolution)
            _, cls, reg = O.box_head(m.W, pf)
        else:
            cls, reg = clsreg
        probs = F.softmax(cls, dim=-1)
        dec = O.apply_deltas(reg, pb, spec.box_reg_weights, spec.scale_clamp)
        det = O.fast_rcnn_inference_single_image(spec, dec, probs, size, nms_trick=False)
        mp = O.roi_pooler(fn, scales, [det["boxes"]], spec.mask_pooler_resolution)
        _, mprob = O.mask_head(spec, m.W, mp, det["classes"])
        det["mask_probs"] = mprob
        return as_np(O.detector_postprocess(det, size, T, T, spec.mask_threshold))

    def unmatched(X, Y):
        """detections of X (score >= 0.1) without a same-class partner at IoU >= 0.95 in Y: (score, class, best same-class IoU in Y,
        that partner's score, box side lengths)"""
        from proj_roadsurf_amd.matching import box_iou
        iou = box_iou(X["boxes"], Y["boxes"])
        out = []
        for a in np.where(X["scores"] >= 0.1)[0]:
            same = np.where(Y["classes"] == X["classes"][a])[0]
            best = same[np.argmax(iou[a, same])] if len(same) else -1
            bi = float(iou[a, best]) if best >= 0 else 0.0
            if bi < 0.95:
                b = X["boxes"][a]
                out.append({"score": round(float(X["scores"][a]), 4), "cls": int(X["classes"][a]), "best_iou": round(bi, 4),
                            "partner_score": round(float(Y["scores"][best]), 4) if best >= 0 else None,
                            "wh": [round(float(b[2] - b[0]), 1), round(float(b[3] - b[1]), 1)]})
        return out

    os.makedirs(os.path.join(ROOT, "gpurun_out", "parity"), exist_ok=True)
    stream = open(os.path.join(ROOT, "gpurun_out", "parity", args.out + "l"), "w")       # one JSON line per event, flushed: survives a killed run

    def emit(rec):
        stream.write(json.dumps(rec) + "\n"); stream.flush()
        print(json.dumps(rec), flush=True)

    def add(tot, fw, bw):
        tot["fw_n"] += fw["n_ref"]; tot["fw_m"] += round(fw["frac_matched"] * fw["n_ref"])
        tot["bw_n"] += bw["n_ref"]; tot["bw_m"] += round(bw["frac_matched"] * bw["n_ref"])

    worst = {"max_dscore": 0.0, "min_mask_iou": 1.0, "min_agg_mask_iou": 1.0}
    for seed in args.seeds:
        t0 = time.time()
        try:
            W, curve = train_trained_like(spec, T, steps=args.steps, seed=seed, lr=args.lr, warmup=args.warmup)
        except RuntimeError as ex:                 # a diverged run is reported, not hidden: it counts as a failed campaign
            emit({"seed": seed, "diverged": str(ex)})
            per_seed.append({"seed": seed, "diverged": True})
            continue
        log(f"seed {seed}: trained {args.steps} steps in {time.time() - t0:.1f} s, loss {curve[0]:.3f} -> {np.mean(curve[-20:]):.3f}")
        tiles, gtb, gtc, _ = synthetic_scenes(args.tiles, T, T, 3, seed=987654 + seed, objects=tuple(args.objects))
        eng = Engine(spec, W, (T, T, 3), max_batch=args.batch)
        m = O.OracleModel(spec, W)
        seed_tot = {k: {"fw_n": 0, "fw_m": 0, "bw_n": 0, "bw_m": 0} for k in pooled}
        n_bisected = 0
        for b0 in range(0, args.tiles, args.batch):
            nb = min(args.batch, args.tiles - b0)
            dets = eng.infer(tiles[b0:b0 + nb])
            cut = None                                    # the engine's intermediate tensors of this batch, fetched on the first miss
            for i in range(nb):
                t, _ = O.predictor_preprocess(spec, tiles[b0 + i])
                x, sizes = O.normalize_and_pad(spec, [t])
                feats = m.backbone(x)
                R = {"A": as_np(m.forward_features(feats, sizes, [(T, T)])[0])}
                d = dets[i]
                R["E"] = {"boxes": d.pred_boxes, "scores": d.scores, "classes": d.pred_classes, "masks": d.pred_masks}
                fw, bw = match_detections(R["A"], R["E"]), match_detections(R["E"], R["A"])
                add(pooled["A_vs_E"], fw, bw); add(seed_tot["A_vs_E"], fw, bw)
                worst["max_dscore"] = max(worst["max_dscore"], fw["max_dscore"])
                worst["min_mask_iou"] = min(worst["min_mask_iou"], float(fw["min_mask_iou"]))
                worst["min_agg_mask_iou"] = min(worst["min_agg_mask_iou"], float(fw["agg_mask_iou"]))
                if fw["frac_matched"] == 1 and bw["frac_matched"] == 1:
                    continue
                # ---- a tile with a miss: where along the chain does it appear?
                if cut is None:
                    cut = {"P": {f"p{l}": torch.from_numpy(eng.tensor(f"p{l}", n=nb).astype(np.float32)).permute(0, 3, 1, 2) for l in (2, 3, 4, 5, 6)},
                           "heads": [torch.from_numpy(eng.tensor(f"rpn_head{l}", n=nb)) for l in (2, 3, 4, 5, 6)],
                           "pb": eng.tensor("proposal_boxes", n=nb), "pc": eng.tensor("proposal_count", n=nb),
                           "pooled": eng.tensor("box_pooled", strip_halo=False), "pred": eng.tensor("box_pred", n=nb)}
                n_bisected += 1
                fB = dict(feats)
                for k in cut["P"]:
                    fB[k] = cut["P"][k][i:i + 1]
                R["B"] = as_np(m.forward_features(fB, sizes, [(T, T)])[0])
                fn = [fB[n] for n in spec.roi_in_features]
                lg = [h[i:i + 1, ..., :A_].permute(0, 3, 1, 2).contiguous() for h in cut["heads"]]
                dl = [h[i:i + 1, ..., A_:5 * A_].permute(0, 3, 1, 2).contiguous() for h in cut["heads"]]
                props = O.rpn_proposals(spec, lg, dl, sizes, nms_trick=False)
                R["C"] = tail(m, fn, props[0]["boxes"], sizes[0])
                n = int(cut["pc"][i])
                pbe = torch.from_numpy(cut["pb"][i, :n].copy())
                pf = torch.from_numpy(cut["pooled"][i * 1024:i * 1024 + n].astype(np.float32)).permute(0, 3, 1, 2).contiguous()
                R["D"] = tail(m, fn, pbe, sizes[0], pooled_feat=pf)
                pr = torch.from_numpy(cut["pred"][i, :n].copy())
                R["F"] = tail(m, fn, pbe, sizes[0], clsreg=(pr[:, :K + 1], pr[:, K + 1:5 * K + 1]))
                rec = {"seed": seed, "tile": b0 + i, "miss": {}}
                for xk, yk in pairs:
                    key = f"{xk}_vs_{yk}"
                    f2, b2 = match_detections(R[xk], R[yk]), match_detections(R[yk], R[xk])
                    if key != "A_vs_E":
                        add(pooled[key], f2, b2); add(seed_tot[key], f2, b2)
                    rec["miss"][key] = {"fw": round(f2["frac_matched"], 3), "bw": round(b2["frac_matched"], 3),
                                        "x_unmatched": unmatched(R[xk], R[yk]), "y_unmatched": unmatched(R[yk], R[xk])}
                misses.append(rec)
                emit(rec)
            log(f"seed {seed}: {b0 + nb} tiles, {time.time() - t0:.0f} s")
        eng.close()
        rec = {"seed": seed, "steps": args.steps, "tiles": args.tiles, "loss_last20": round(float(np.mean(curve[-20:])), 3),
               "A_vs_E": seed_tot["A_vs_E"], "tiles_bisected": n_bisected}
        emit(rec)
        per_seed.append(rec)
    table = {}
    for k, v in pooled.items():
        table[k] = {**v, "fw": round(v["fw_m"] / max(v["fw_n"], 1), 4), "bw": round(v["bw_m"] / max(v["bw_n"], 1), 4),
                    "fw_wilson_lo": round(wilson_lower(v["fw_m"], v["fw_n"]), 4), "bw_wilson_lo": round(wilson_lower(v["bw_m"], v["bw_n"]), 4)}
        print(k, json.dumps(table[k]), flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity", args.out), "w") as f:
        json.dump({"tool": "python tools/parity/bisect_stages.py " + " ".join(sys.argv[1:]),
                   "note": "A_vs_E is pooled over ALL tiles; the other pairs only over the tiles where A and E differ (the bisected ones)",
                   "seeds": per_seed, "pooled": table, "worst": worst, "misses": misses}, f, indent=1)


if __name__ == "__main__":
    main()
