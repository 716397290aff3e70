#!/usr/bin/env python3
"""Time rs_op_conv2d tile variants on the small-map layers of the batch-16 forward (res4 / res5 / laterals / fc2), where the pixel
count fills the 256 CUs badly in 256-pixel tiles.  Kernel time by HIP events on the null stream, 200 back-to-back launches after
a 1 s warm-up.  Usage: conv_shapes.py [variant ...]   (default 7 0 12 15 16 17)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from proj_roadsurf_amd import engine as E

lib = E.load_library(os.environ.get("RS_LIB") or None)
dev = torch.device("cuda:0")
variants = [int(v) for v in sys.argv[1:]] or [7, 0, 12, 15, 16, 17]
B = int(os.environ.get("BATCH", "16"))
shapes = [("res4.x.conv1", B, 50, 50, 1024, 1, 256), ("res4.0.conv1", B, 50, 50, 512, 1, 256), ("res4.x.conv2", B, 50, 50, 256, 3, 256),
          ("res4.x.conv3", B, 50, 50, 256, 1, 1024), ("fpn_lateral4", B, 50, 50, 1024, 1, 256), ("fpn_lateral3", B, 100, 100, 512, 1, 256),
          ("res5.x.conv1", B, 25, 25, 2048, 1, 512), ("res5.x.conv2", B, 25, 25, 512, 3, 512), ("res5.x.conv3", B, 25, 25, 512, 1, 2048),
          ("fpn_lateral5", B, 25, 25, 2048, 1, 256), ("fpn_lateral2", B, 200, 200, 256, 1, 256), ("box.fc2", B * 10, 10, 10, 1024, 1, 1024), ("box.fc1", B * 10, 10, 10, 12544, 1, 1024), ("mask.fcn", B * 100, 14, 14, 256, 3, 256)]
only = os.environ.get("SHAPES")
for name, N, H, W, Cin, k, Cout in shapes:
    if only and not any(name.startswith(x) for x in only.split(",")):
        continue
    pad = k // 2
    x = torch.randn(N, H + 2 * pad, W + 2 * pad, Cin, device=dev).half()
    w = (torch.randn(Cout, k * k * Cin, device=dev) * 0.02).half()
    b = torch.zeros(Cout, device=dev)
    o = torch.zeros(N, H + 2, W + 2, Cout, device=dev, dtype=torch.float16)
    flop = 2.0 * N * H * W * k * k * Cin * Cout
    line = f"{name:13s} M {N * H * W:6d} K {k * k * Cin:5d} N {Cout:4d}:"
    for v in variants:
        def launch():
            return lib.rs_op_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(o.data_ptr()), None, None,
                                    N, H, W, Cin, pad, k, k, 1, pad, Cout, k * k * Cin, 1, 1, 0, 0, v, 1, None)
        if launch() != 0:
            line += f"  v{v}: n/a"
            continue
        t0 = time.time()
        while time.time() - t0 < 0.7:
            for _ in range(50):
                launch()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
        line += f"  v{v}: {ms * 1e3:6.1f} us {flop / ms / 1e9:6.0f} TF"
    print(line, flush=True)
