#!/usr/bin/env python3
"""Stage table of the reference-precision (fp32 MFMA) engine at batch 16 of 512x512x3 tiles: ms per call, TFLOP/s against the 157.3 TFLOP/s
fp32 matrix peak, GB/s of algorithmic bytes; the 20 longest stages again at the end."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from proj_roadsurf_amd.engine import Engine
from proj_roadsurf_amd.spec import EngineSpec
from proj_roadsurf_amd.synthetic import synthetic_tiles
from proj_roadsurf_amd.weights import synthetic_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
PREC = sys.argv[2] if len(sys.argv) > 2 else "fp32"      # "fp32" (fp32 matrix cores) or "split" (hi + lo planes on the fp16 matrix cores)
spec = EngineSpec(num_classes=2, precision=PREC)
W = synthetic_weights(spec, seed=0)
tiles = synthetic_tiles(B, 512, 512, 3, seed=1234)
e = Engine(spec, W, (512, 512, 3), max_batch=B)
p = e.upload_tiles(tiles)
for _ in range(2):
    e.infer_device(p, B)
e.sync()
e.set_profiling(2)
for _ in range(4):
    e.infer_device(p, B)
e.sync()
torch.cuda.synchronize()
st = [s for s in e.stage_times() if s["calls"]]
rows = [(s["name"], s["ms_total"] / s["calls"], s["flops"], s["bytes"], s["kernel"]) for s in st]
tot = sum(r[1] for r in rows)
for nm, ms, fl, by, kn in rows:
    print(f"{nm:32s} {ms:8.4f} ms {fl / ms / 1e9 if fl else 0:7.1f} TF {by / ms / 1e6 if by else 0:7.0f} GB/s  {kn[:40]}")
print(f"sum {tot:.2f} ms; matrix stages {sum(r[2] for r in rows) / sum(r[1] for r in rows if r[2]) / 1e9:.1f} TFLOP/s over {sum(r[1] for r in rows if r[2]):.2f} ms")
print("---- longest")
for nm, ms, fl, by, kn in sorted(rows, key=lambda r: -r[1])[:20]:
    print(f"{nm:32s} {ms:8.4f} ms {fl / ms / 1e9 if fl else 0:7.1f} TF {by / ms / 1e6 if by else 0:7.0f} GB/s")
e.close()
