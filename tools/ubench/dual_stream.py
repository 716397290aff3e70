#!/usr/bin/env python3
"""Experiment: does splitting a batch of 16 into sub-batches on independent engines/streams (so one sub-batch's
latency-bound detection glue overlaps the other's convolutions) beat a single engine at batch 16?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from proj_roadsurf_amd.engine import Engine
from proj_roadsurf_amd.spec import EngineSpec
from proj_roadsurf_amd.weights import synthetic_weights
from proj_roadsurf_amd.synthetic import synthetic_tiles

spec = EngineSpec(num_classes=2)
W = synthetic_weights(spec, seed=0)
B, T, steps = 16, 512, 20
tiles = synthetic_tiles(B, T, T, 3, seed=1234)
for nsub in [1, 2, 4]:
    b = B // nsub
    engs = [Engine(spec, W, (T, T, 3), max_batch=b, device=0) for _ in range(nsub)]
    ptrs = [e.upload_tiles(tiles[i * b:(i + 1) * b]) for i, e in enumerate(engs)]
    for _ in range(3):
        for e, p in zip(engs, ptrs):
            e.infer_device(p, b)
    for e in engs:
        e.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for e, p in zip(engs, ptrs):
            e.infer_device(p, b)
    for e in engs:
        e.sync()
    dt = time.perf_counter() - t0
    print(f"nsub={nsub} batch/engine={b}: {B * steps / dt:.1f} tiles/s ({dt / steps * 1e3:.3f} ms/step)", flush=True)
    del engs, ptrs

# pipelined: full batches of 16, consecutive batches alternate between independent engines (streams)
for neng in [2, 3]:
    engs = [Engine(spec, W, (T, T, 3), max_batch=B, device=0) for _ in range(neng)]
    ptrs = [e.upload_tiles(tiles) for e in engs]
    for k in range(2 * neng):
        engs[k % neng].infer_device(ptrs[k % neng], B)
    for e in engs:
        e.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        engs[k % neng].infer_device(ptrs[k % neng], B)
    for e in engs:
        e.sync()
    dt = time.perf_counter() - t0
    print(f"pipelined engines={neng} batch=16: {B * steps / dt:.1f} tiles/s ({dt / steps * 1e3:.3f} ms/step)", flush=True)
    del engs, ptrs
