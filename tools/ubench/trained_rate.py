#!/usr/bin/env python3
"""Rates of the lane pipeline on the TRAINED-LIKE workload (a handful of clean objects per tile, what make_detections sees on real
tiles) instead of bench.py's saturated random-weight workload: tiles resident in HBM, the streaming host interface, and the stage
table -- to see what the CLI's forward thread can reach at best."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from proj_roadsurf_amd.engine import LanePipeline
from proj_roadsurf_amd.spec import EngineSpec
from proj_roadsurf_amd.synthetic import synthetic_scenes, train_trained_like

B = 16
spec = EngineSpec(num_classes=2)
W, _ = train_trained_like(spec, 512, steps=300)
tiles = synthetic_scenes(B, 512, 512, 3, seed=555)[0]
for lanes in (2, 1):
    pipe = LanePipeline(spec, W, (512, 512, 3), max_batch=B, lanes=lanes)
    ptrs = [e.upload_tiles(tiles) for e in pipe.engines]
    for k in range(6):
        pipe.submit(ptrs[k % lanes], B)
    pipe.sync()
    t0 = time.perf_counter()
    for k in range(40):
        pipe.submit(ptrs[pipe.k % lanes], B)
    pipe.sync()
    dt = time.perf_counter() - t0
    print(f"lanes {lanes}: resident {40 * B / dt:8.1f} tiles/s", flush=True)
    for _ in pipe.run(tiles for _ in range(4)):
        pass
    t1 = time.perf_counter()
    nd = 0
    for res in pipe.run(tiles for _ in range(40)):
        nd += sum(len(r) for r in res)
    dt = time.perf_counter() - t1
    print(f"lanes {lanes}: host interface {40 * B / dt:8.1f} tiles/s, {nd / 40 / B:.1f} detections per tile", flush=True)
    if lanes == 1:
        e = pipe.engines[0]
        e.set_profiling(2)
        for k in range(8):
            pipe.submit(ptrs[0], B)
        pipe.sync()
        st = [s for s in e.stage_times() if s["calls"]]
        tot = sum(s["ms_total"] / s["calls"] for s in st)
        print(f"stage sum {tot:.3f} ms/batch; " + ", ".join(f"{s['name']} {s['ms_total'] / s['calls']:.3f}" for s in st if s["name"].startswith(("mask", "box", "rpn.s", "rpn.n", "rpn.m"))))
    pipe.close()
