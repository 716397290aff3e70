#!/usr/bin/env python3
"""Race check for the trainer's side stream (weight / bias gradients next to the input-gradient chain): the flat gradient of
the same step (same tiles, targets, sampling seed) with RS_TRAIN_SIDE=0 and =1, several repeats each.  The float atomics of
ROIAlign-backward make two runs differ in the last bits; a race would show up as a difference far above that floor.

    python tools/ubench/side_stream_check.py [--batch 4] [--tile 256] [--repeats 4]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--tile", type=int, default=256)
    ap.add_argument("--repeats", type=int, default=4)
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
        k = int(rng.integers(3, 9))
        xy = rng.uniform(10, T - 90, (k, 2))
        wh = rng.uniform(20, 80, (k, 2))
        b = np.concatenate([xy, xy + wh], 1) * s
        boxes.append(b.astype(np.float32))
        classes.append(rng.integers(0, 2, k))
        polys.append([[np.array([x0, y0, x1, y0, x1, y1, x0, y1])] for x0, y0, x1, y1 in b.tolist()])

    def grads(side):
        os.environ["RS_TRAIN_SIDE"] = str(side)
        tr = Trainer(spec, W, (T, T, 3), batch=B, loss_scale=1024.0)
        out = []
        try:
            for r in range(args.repeats):
                tr.train_step(tiles, boxes, classes, polys, seed=7)
                tr.sync()
                n = tr.param_count
                host = np.empty(n, np.float32)
                ptr = int(tr.lib.rs_trainer_grad_buffer(tr._h))
                rc = tr.lib.rs_memcpy_d2h(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), host.nbytes)
                assert rc == 0
                out.append(host)
        finally:
            tr.close()
        return out

    g0 = grads(0)
    g1 = grads(1)
    ref = g0[0]
    scale = float(np.abs(ref).max())

    def d(a, b):
        return float(np.abs(a - b).max()) / scale, float(np.linalg.norm(a - b) / np.linalg.norm(b))
    res = {"scale": scale,
           "floor_side0_vs_side0": [d(g, ref) for g in g0[1:]],
           "side1_vs_side0": [d(g, ref) for g in g1]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
