#!/usr/bin/env python3
"""Tile shapes of the reference-precision (fp32 MFMA) convolution on the layers where its 128 x 128 tile fills the chip badly: kernel time by
HIP events, 100 back-to-back launches after a warm-up.  Variants: 30 = 128 px x 128 ch (4 waves), 31 = 128 x 64 (4 waves), 32 = 64 x 128,
33 = 256 x 128 (8 waves), 34 = 128 x 256 (8 waves), 35 = 128 x 64 (2 waves).  BATCH=16"""
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
variants = [int(v) for v in sys.argv[1:]] or [30, 31, 32, 33, 34, 35]
B = int(os.environ.get("BATCH", "16"))
shapes = [("res4.x.conv2", B, 50, 50, 256, 3, 256), ("res4.x.conv1", B, 50, 50, 1024, 1, 256), ("res4.x.conv3", B, 50, 50, 256, 1, 1024),
          ("res5.x.conv2", B, 25, 25, 512, 3, 512), ("res5.x.conv1", B, 25, 25, 2048, 1, 512), ("res5.x.conv3", B, 25, 25, 512, 1, 2048),
          ("res3.x.conv2", B, 100, 100, 128, 3, 128), ("res3.x.conv3", B, 100, 100, 128, 1, 512), ("res2.x.conv3", B, 200, 200, 64, 1, 256),
          ("res2.x.conv2", B, 200, 200, 64, 3, 64), ("fpn_output4", B, 50, 50, 256, 3, 256), ("fpn_lateral2", B, 200, 200, 256, 1, 256),
          ("fpn_output3", B, 100, 100, 256, 3, 256)]
for name, N, H, W, Cin, k, Cout in shapes:
    pad = k // 2
    x = torch.randn(N, H + 2 * pad, W + 2 * pad, Cin, device=dev)
    w = torch.randn(Cout, k * k * Cin, device=dev) * 0.02
    b = torch.zeros(Cout, device=dev)
    o = torch.zeros(N, H + 2, W + 2, Cout, device=dev)
    flop = 2.0 * N * H * W * k * k * Cin * Cout
    line = f"{name:13s} M {N * H * W:6d} K {k * k * Cin:5d} N {Cout:4d}:"
    for v in variants:
        def launch():
            return lib.rs_op_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(o.data_ptr()), None, None,
                                    N, H, W, Cin, pad, k, k, 1, pad, Cout, k * k * Cin, 1, 1, 1, 0, v, -1, None)
        if launch() != 0:
            line += f"  v{v}:   n/a      "
            continue
        t0 = time.time()
        while time.time() - t0 < 0.5:
            for _ in range(20):
                launch()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        line += f"  v{v}: {ms * 1e3:6.0f} us {flop / ms / 1e9:5.0f}"
    print(line, flush=True)
