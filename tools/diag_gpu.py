#!/usr/bin/env python3
"""Diagnostics on the GPU box: RoI sampling-grid statistics of the bench workload and fp16-vs-oracle error stats."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from proj_roadsurf_amd.engine import Engine
from proj_roadsurf_amd.spec import EngineSpec
from proj_roadsurf_amd.weights import synthetic_weights
from proj_roadsurf_amd.synthetic import synthetic_tiles

spec = EngineSpec(num_classes=2)
W = synthetic_weights(spec, 0)
B = 4
tiles = synthetic_tiles(B, 512, 512, 3, seed=1234)
eng = Engine(spec, W, (512, 512, 3), max_batch=B)
dets = eng.infer(tiles, want_probs=True)
pb = eng.tensor("proposal_boxes", n=B)[:, :1000]
lv = eng.tensor("box_roi_level", n=B)[:, :1000]
print("proposal level hist", np.bincount(lv.ravel(), minlength=4))
w = pb[..., 2] - pb[..., 0]; h = pb[..., 3] - pb[..., 1]
sc = np.array([1 / 4, 1 / 8, 1 / 16, 1 / 32])[lv]
gw = np.ceil(w * sc / 7); gh = np.ceil(h * sc / 7)
print("box roi: mean w,h px", w.mean(), h.mean(), "mean gw*gh", (gw * gh).mean(), "max", (gw * gh).max(), "p50/p90/p99", np.percentile(gw * gh, [50, 90, 99]))
dn = eng.tensor("det_boxes_net", n=B)
w = dn[..., 2] - dn[..., 0]; h = dn[..., 3] - dn[..., 1]
area = np.sqrt(np.maximum(w * h, 0)) / 224 + 1e-8
l = np.clip(np.floor(4 + np.log2(area)), 2, 5).astype(int) - 2
sc = np.array([1 / 4, 1 / 8, 1 / 16, 1 / 32])[l]
g = np.ceil(w * sc / 14) * np.ceil(h * sc / 14)
print("mask roi: level hist", np.bincount(l.ravel(), minlength=4), "mean g", g.mean(), "max", g.max())
print("dets per tile", [len(d) for d in dets], "score range", [(float(d.scores.min()), float(d.scores.max())) for d in dets])
print("mask prob hist", np.histogram(np.concatenate([d.mask_probs.ravel() for d in dets]), bins=10, range=(0, 1))[0])
eng.close()
