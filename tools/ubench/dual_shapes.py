#!/usr/bin/env python3
"""Time rs_op_conv2d_dual tile variants on the projection-block outputs (conv3 + strided shortcut as one GEMM over two K sources)
of the batch-16 forward.  Usage: dual_shapes.py [variant ...]   (default 0 4 7 10 14)"""
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
variants = [int(v) for v in sys.argv[1:]] or [0, 4, 7, 10, 14]
B = int(os.environ.get("BATCH", "16"))
shapes = [("res3.0.conv3", B, 100, 100, 128, 256, 512), ("res4.0.conv3", B, 50, 50, 256, 512, 1024), ("res5.0.conv3", B, 25, 25, 512, 1024, 2048)]
for name, N, H, W, cin, cin2, cout in shapes:
    a = torch.randn(N, H + 2, W + 2, cin, device=dev).half()
    b = torch.randn(N, 2 * H + 2, 2 * W + 2, cin2, device=dev).half()
    w = (torch.randn(cout, cin + cin2, device=dev) * 0.02).half()
    bias = torch.zeros(cout, device=dev)
    o = torch.zeros(N, H + 2, W + 2, cout, device=dev, dtype=torch.float16)
    flop = 2.0 * N * H * W * (cin + cin2) * cout
    line = f"{name:13s} M {N * H * W:6d} K {cin + cin2:5d} N {cout:4d}:"
    for v in variants:
        def launch():
            return lib.rs_op_conv2d_dual(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()),
                                         C.c_void_p(o.data_ptr()), N, H, W, cin, 1, 1, 1, 1, 0, 2 * H, 2 * W, cin2, 1, 2, cout, cin + cin2, 1, 1, v, None)
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
