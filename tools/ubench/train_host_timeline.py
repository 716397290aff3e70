#!/usr/bin/env python3
"""Host-side timeline of one training step (batch 8): wall time spent INSIDE each host call of Trainer.train_step (enqueue cost, or a wait
when the call synchronises), averaged over 10 steps, and the step time with / without the loss read-back at the end."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from proj_roadsurf_amd.engine import Trainer
from proj_roadsurf_amd.spec import EngineSpec
from proj_roadsurf_amd.synthetic import synthetic_scenes
from proj_roadsurf_amd.weights import synthetic_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = 512
spec = EngineSpec(num_classes=2)
W = synthetic_weights(spec, seed=0)
tiles, boxes, classes, polys = synthetic_scenes(B, T, T, 3, seed=4321)
s = 800.0 / T
nb = [b * np.float32(s) for b in boxes]
npoly = [[[p * s for p in inst] for inst in img] for img in polys]
tr = Trainer(spec, W, (T, T, 3), batch=B, device=0, loss_scale=1024.0)
acc = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


for nm in ("set_targets", "upload_tiles", "forward_trunk", "rpn_forward", "roi_step", "mask_forward", "mask_entries", "mask_backward", "rpn_step",
           "backward_trunk", "tensor", "apply_sgd"):
    setattr(tr, nm, timed(nm, getattr(tr, nm)))
for it in range(3):
    tr.train_step(tiles, nb, classes, npoly, seed=it)
    tr.apply_sgd(1e-5, 0.9, 1e-4)
tr.sync(); torch.cuda.synchronize()
acc.clear()
N = 10
t0 = time.perf_counter()
for it in range(N):
    tr.train_step(tiles, nb, classes, npoly, seed=100 + it)
    tr.apply_sgd(1e-5, 0.9, 1e-4)
tr.sync(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"batch {B}: {dt * 1e3:.2f} ms/step")
for k, v in acc.items():
    print(f"  {k:16s} {v / N * 1e3:7.3f} ms")
print(f"  sum              {sum(acc.values()) / N * 1e3:7.3f} ms")
tr.close()
