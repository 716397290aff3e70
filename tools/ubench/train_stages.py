#!/usr/bin/env python3
"""Per-stage table of one training step (BASELINE configs[4] on one GPU, fp16 trainer unless PRECISION=fp32): HIP events per stage on the
stream the stage runs on (rs_trainer_set_profiling), averaged over 4 steps, sorted as executed; the 25 longest stages again at the end.
Usage: train_stages.py [batch]"""
import os
import sys

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
if os.environ.get("PRECISION") == "fp32":
    spec = spec.replace(precision="fp32")
W = synthetic_weights(spec, seed=0)
tiles, boxes, classes, polys = synthetic_scenes(B, T, T, 3, seed=4321)
s = 800.0 / T
nb = [b * np.float32(s) for b in boxes]
npoly = [[[p * s for p in inst] for inst in img] for img in polys]
tr = Trainer(spec, W, (T, T, 3), batch=B, device=0, loss_scale=1024.0)


def run(n):
    for it in range(n):
        tr.train_step(tiles, nb, classes, npoly, seed=100 + it, allreduce=False)
        tr.apply_sgd(1e-5, 0.9, 1e-4)
    tr.sync()
    torch.cuda.synchronize()


run(3)
import time
t0 = time.perf_counter(); run(10); dt = (time.perf_counter() - t0) / 10
print(f"batch {B}: {dt * 1e3:.2f} ms/step unprofiled")
tr.set_profiling(True)
run(4)
st = [x for x in tr.stage_times() if x["calls"]]
tr.set_profiling(False)
tot = {False: 0.0, True: 0.0}
rows = []
for x in st:
    ms = x["ms_total"] / x["calls"]
    tot[x["side"]] += ms
    rows.append((x["name"], ms, x["flops"] / ms / 1e9 if x["flops"] > 0 else 0.0, x["side"]))
for nm, ms, tf, side in rows:
    print(f"{nm:44s} {ms:8.4f} ms {tf:8.1f} TF {'side' if side else ''}")
print(f"chain {tot[False]:.2f} ms, side stream {tot[True]:.2f} ms")
print("---- longest")
for nm, ms, tf, side in sorted(rows, key=lambda r: -r[1])[:30]:
    print(f"{nm:44s} {ms:8.4f} ms {tf:8.1f} TF {'side' if side else ''}")
tr.close()
